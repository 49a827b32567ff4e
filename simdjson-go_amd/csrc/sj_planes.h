// sj_planes.h -- the scan element of 64 consecutive tokens from the bit planes of their kinds, host+device.
//
// token_element (sj_stage2.h) states what ONE token contributes to the device-wide scan -- tape words, bracket and
// record counts, and the allowed-context function of its gap -- as a function of its kind and the kinds of its
// neighbours; the kernels evaluate it per token through a 512-entry table (token_pelement), ~28 instructions and an LDS
// look-up each, and the token kernels are bound by instruction issue.  This header is the same
// function for 64 tokens at once: with the kinds of a group transposed into four 64-bit planes (bit j = token j), every
// class of tokens is an AND of plane literals, "the token in front is an X" is a shift by one with a carry from the
// group in front, and
//     tape words      = popc(w1) + popc(w2)          brackets = popc(br)      opens = popc(open)
//     record newlines = popc(nlr)
//     allowed contexts: three masks a_root / a_obj / a_arr (token j is legal in that context), gap_start (the token in
//                       front is a bracket), from which the composed context function of the group and the set of any
//                       bracket's gap follow with a handful of mask operations (group_function, group_gap_set)
// -- a few instructions per 64 tokens instead of per token.  Today it is the THIRD statement of the element: the host
// replay (host_selftest.cpp) folds token_element over every group of every document of the CPU suite and fails on any
// difference from the plane form; the kernels still use the table.  It is what a token pass on bit planes (the next
// step DESIGN.md names for k_measure / k_s2_emit) computes per lane.
// Grammar: grammar_violation_v restates unifiedMachine (stage2_build_tape_amd64.go:160-446); the masks below are that
// rule solved for "the set of contexts in which token j is legal", class by class.
#pragma once
#include "sj_stage2.h"

namespace sj {

struct KindPlanes {
    u64 b0, b1, b2, b3;  // bit j of b_i = bit i of the kind of token j
};

// reference transposition (the device form is a byte -> bit-plane butterfly like stage 1's)
SJ_HD KindPlanes kind_planes(const u8 *kinds, u32 count) {
    KindPlanes p{0, 0, 0, 0};
    for (u32 j = 0; j < count && j < 64; j++) {
        const u64 k = kinds[j];
        p.b0 |= (k & 1u) << j;
        p.b1 |= ((k >> 1) & 1u) << j;
        p.b2 |= ((k >> 2) & 1u) << j;
        p.b3 |= ((k >> 3) & 1u) << j;
    }
    return p;
}

// The same planes from the 64 kind bytes as sixteen little-endian words, without a loop over the tokens: bit b of the
// four bytes of a word is isolated ((w >> b) & 0x01010101) and gathered into a nibble by one multiplication -- the
// partial products 2^(8i) * 2^(28 - 7j) land on distinct bits, the ones with i == j on bits 28 + i (the device form;
// count < 64: the bytes behind the end must be zero).
SJ_HD KindPlanes kind_planes_words(const u32 *w16) {
    u64 pl[4] = {0, 0, 0, 0};
    for (int q = 0; q < 16; q++) {
        const u32 w = w16[q];
        for (int b = 0; b < 4; b++) {
            const u32 nib = (((w >> b) & 0x01010101u) * 0x10204080u) >> 28;
            pl[b] |= (u64)nib << (4 * q);
        }
    }
    return KindPlanes{pl[0], pl[1], pl[2], pl[3]};
}

struct KindClasses {
    u64 open_obj, open_arr, close_obj, close_arr, colon, comma, string, num, atom, nl;
    u64 open, close, bracket;
};
SJ_HD KindClasses kind_classes(const KindPlanes &p, u64 valid) {
    const u64 n0 = ~p.b0, n1 = ~p.b1, n2 = ~p.b2, n3 = ~p.b3;
    KindClasses c;
    c.open_obj = n3 & n2 & n1 & p.b0 & valid;    // 1
    c.open_arr = n3 & n2 & p.b1 & n0 & valid;    // 2
    c.close_obj = n3 & n2 & p.b1 & p.b0 & valid; // 3
    c.close_arr = n3 & p.b2 & n1 & n0 & valid;   // 4
    c.colon = n3 & p.b2 & n1 & p.b0 & valid;     // 5
    c.comma = n3 & p.b2 & p.b1 & n0 & valid;     // 6
    c.string = n3 & p.b2 & p.b1 & p.b0 & valid;  // 7
    c.num = p.b3 & n2 & n1 & n0 & valid;         // 8
    c.atom = p.b3 & n2 & (p.b1 | p.b0) & valid;  // 9, 10, 11
    c.nl = p.b3 & p.b2 & n1 & n0 & valid;        // 12
    c.open = c.open_obj | c.open_arr;
    c.close = c.close_obj | c.close_arr;
    c.bracket = c.open | c.close;
    return c;
}

// what the group needs from its neighbours: the kinds of the two tokens in front of it (K_NONE where there is none:
// the document starts inside this group) and of the token behind it (K_NL behind the last token, like the kernels'
// sentinel: a newline run at the very end writes no root pair)
struct GroupCarry {
    u8 ppk, pk, nk;
};

struct GroupMasks {
    u64 w1, w2;        // tokens that write at least one / two tape words
    u64 br, open;      // brackets, opening brackets
    u64 nlr;           // newlines that separate two records (the last one of a run, a token behind it)
    u64 a_root, a_obj, a_arr;  // token j is legal in the context
    u64 gap_start;     // the token in front of j is a bracket: j is the first token of a gap
    u64 valid;
};

SJ_HD u64 kind_bit(u8 k, u8 want) { return k == want ? 1ull : 0ull; }

// first_of_document: token 0 of this group is token 0 of the message (legal iff it opens a container, in any context)
SJ_HD GroupMasks group_masks(const KindPlanes &p, u64 valid, const GroupCarry &cy, bool first_of_document) {
    const KindClasses c = kind_classes(p, valid);
    // "the token in front is an X": shift by one, the carry is the last token of the group in front
    auto prev = [&](u64 x, u64 carry) { return (x << 1) | carry; };
    const u64 p_open_obj = prev(c.open_obj, kind_bit(cy.pk, K_OPEN_OBJ)), p_open_arr = prev(c.open_arr, kind_bit(cy.pk, K_OPEN_ARR));
    const u64 p_close = prev(c.close, (cy.pk == K_CLOSE_OBJ || cy.pk == K_CLOSE_ARR) ? 1ull : 0ull);
    const u64 p_colon = prev(c.colon, kind_bit(cy.pk, K_COLON)), p_comma = prev(c.comma, kind_bit(cy.pk, K_COMMA));
    const u64 p_string = prev(c.string, kind_bit(cy.pk, K_STRING)), p_nl = prev(c.nl, kind_bit(cy.pk, K_NL));
    const u64 p_scalar = prev(c.num | c.atom, (cy.pk == K_NUM || cy.pk == K_TRUE || cy.pk == K_FALSE || cy.pk == K_NULL) ? 1ull : 0ull);
    const u64 p_bracket = p_open_obj | p_open_arr | p_close;
    // "the token two in front is '{' or ','": a string in front of j is then a key (string_is_key_v)
    const u64 kp_src = c.open_obj | c.comma;
    const u64 key_prev = (kp_src << 2) | ((cy.pk == K_OPEN_OBJ || cy.pk == K_COMMA) ? 2ull : 0ull) |
                         ((cy.ppk == K_OPEN_OBJ || cy.ppk == K_COMMA) ? 1ull : 0ull);
    // "the token behind is a newline"
    const u64 n_nl = (c.nl >> 1) | (kind_bit(cy.nk, K_NL) << 63);
    // a value ends in front of j (ends_value_v): a close, a scalar, or a string that is not a key -- in an array a
    // string never is one
    const u64 p_end = p_close | p_scalar;
    const u64 p_end_obj = p_end | (p_string & ~key_prev), p_end_arr = p_end | p_string;
    const u64 value = c.open | c.num | c.atom;  // containers and scalars: the same rule below the root
    GroupMasks m;
    m.valid = valid;
    m.br = c.bracket;
    m.open = c.open;
    m.nlr = c.nl & ~n_nl;
    m.w2 = c.string | c.num | m.nlr;
    m.w1 = c.bracket | c.atom | m.w2;
    m.a_root = (c.open & p_nl) | (c.nl & (p_close | p_nl));
    m.a_obj = (value & p_colon) | (c.string & (p_open_obj | p_comma | p_colon)) | (c.colon & p_string & key_prev) | (c.comma & p_end_obj) |
              (c.close_obj & (p_open_obj | p_end_obj));
    m.a_arr = ((value | c.string) & (p_open_arr | p_comma)) | (c.comma & p_end_arr) | (c.close_arr & (p_open_arr | p_end_arr));
    m.gap_start = p_bracket & valid;
    if (first_of_document) {  // `if (i == 0) return !is_open(k)` for every context; no token in front
        const u64 ok = c.open & 1ull;
        m.a_root = (m.a_root & ~1ull) | ok;
        m.a_obj = (m.a_obj & ~1ull) | ok;
        m.a_arr = (m.a_arr & ~1ull) | ok;
        m.gap_start &= ~1ull;
    }
    // tokens behind the end of the message: the identity element
    m.a_root |= ~valid;
    m.a_obj |= ~valid;
    m.a_arr |= ~valid;
    return m;
}

// The composed context function of the group, in the form of Agg::am (p | q << 3, see am_combine): without a gap start
// the group narrows the set in front of it (p = the contexts every token allows, q = none); with one, what is in front
// no longer matters (p = none) and q = the contexts every token from the LAST gap start on allows.
SJ_HD u32 group_function(const GroupMasks &m) {
    const u64 bad[3] = {~m.a_root, ~m.a_obj, ~m.a_arr};
    u32 am = 0;
    if (m.gap_start == 0) {
        for (int c = 0; c < 3; c++)
            if (bad[c] == 0) am |= 1u << c;
        return am;
    }
    const u64 from_last = ~0ull << (63 - clz64(m.gap_start));  // bits from the last gap start on
    for (int c = 0; c < 3; c++)
        if ((bad[c] & from_last) == 0) am |= 8u << c;
    return am;
}

// the scan element of the whole group (Strings.B bytes are not part of it: the emit masks place the strings)
SJ_HD Agg group_aggregate(const GroupMasks &m) {
    Agg a;
    const u32 opens = (u32)popc64(m.open), brackets = (u32)popc64(m.br);
    a.d = (i32)(2u * opens) - (i32)brackets;
    a.w = (u32)popc64(m.w1) + (u32)popc64(m.w2);
    a.s = 0;
    a.nb = (u32)popc64(m.nlr);
    a.bc = brackets;
    a.am = group_function(m);
    return a;
}

// allowed contexts of the gap that ends with token j of the group (a bracket: gap_mask(x, e) of the per-token form);
// am_in = Agg::am of everything in front of the group
SJ_HD u32 group_gap_set(const GroupMasks &m, u32 j, u32 am_in) {
    const u64 upto = j == 63 ? ~0ull : ((1ull << (j + 1)) - 1ull);  // tokens 0 .. j
    const u64 starts = m.gap_start & upto;
    const u64 bad[3] = {~m.a_root, ~m.a_obj, ~m.a_arr};
    u32 set = 0;
    if (starts == 0) {  // the gap began in front of the group
        for (int c = 0; c < 3; c++)
            if ((bad[c] & upto) == 0) set |= 1u << c;
        return am_value(am_combine(am_in, set));
    }
    const u64 range = upto & (~0ull << (63 - clz64(starts)));
    for (int c = 0; c < 3; c++)
        if ((bad[c] & range) == 0) set |= 1u << c;
    return set;
}

// tape offset (relative to the group) of token j, and the number of brackets in front of it inside the group: what a
// lane needs per token once the group's masks are known
SJ_HD u32 group_words_before(const GroupMasks &m, u32 j) {
    const u64 below = j == 0 ? 0ull : (~0ull >> (64 - j));
    return (u32)popc64(m.w1 & below) + (u32)popc64(m.w2 & below);
}
SJ_HD u32 group_brackets_before(const GroupMasks &m, u32 j) {
    const u64 below = j == 0 ? 0ull : (~0ull >> (64 - j));
    return (u32)popc64(m.br & below);
}

}  // namespace sj
