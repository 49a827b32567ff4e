// sj_tok16.h -- the token pass of stage 2 on bit planes, sixteen tokens per lane (host+device).
//
// unifiedMachine (stage2_build_tape_amd64.go:160-446) visits one structural index per iteration; the first three rounds
// of this engine gave every token a lane and recovered the machine's state from a scan over per-token elements
// (sj_stage2.h token_element: a 512-entry table look-up and ~28 instructions per token, then ~40 more to turn the prefix
// into offsets and queue entries) -- the token kernels were bound by instruction issue.  Here a lane owns SIXTEEN
// consecutive tokens: their kind bytes (one 16-byte load) are transposed into four 16-bit planes with an 8x8 bit-matrix
// transposition (three butterfly stages on two dwords), widened into 19-bit WINDOWS that also hold the two tokens in front
// and the one behind (bit i of a window = token i - 2), and everything token_element states becomes a boolean function of
// the windows, evaluated for all sixteen tokens at once:
//     tape words = popc(w1) + popc(w2), brackets = popc(br), opens = popc(open), records = popc(nlr),
//     the three allowed-context masks a_root / a_obj / a_arr and gap_start  (sj_planes.h states the same algebra for 64
//     tokens per word; this is its 16-token form with the neighbours inside the word, so no carries),
// and per-token work remains only for tokens that WRITE something (strings, scalars, brackets, record newlines): their
// tape offset is a popcount below their bit.  csrc/host_selftest.cpp folds token_element over every lane of every
// document of the CPU suite and fails on any difference (aggregate, per-token offsets, bracket depth / gap set).
#pragma once
#include "sj_stage2.h"

namespace sj {

// ---- kinds -> planes ----------------------------------------------------------------------------------------------
// w[q] holds kinds 4q .. 4q+3 (one per byte, values 0..15).  Returns P01 = plane0 | plane1 << 16, P23 = plane2 | plane3 << 16
// (bit j of a plane = bit p of the kind of token j).
struct Planes16 {
    u32 p01, p23;
};
SJ_HD u32 tok_perm(u32 hi, u32 lo, u32 sel) {  // v_perm_b32: selector bytes 0-3 pick from lo, 4-7 from hi
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const u64 v = ((u64)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; i++) r |= (u32)((v >> (8 * ((sel >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
    return r;
#endif
}
SJ_HD Planes16 planes16(u32 w0, u32 w1, u32 w2, u32 w3) {
    // rows of an 8x8 bit matrix: byte r of (lo, hi) = kind[r] | kind[8 + r] << 4
    u32 lo = w0 | (w2 << 4), hi = w1 | (w3 << 4);
    u32 t;
    // transpose (bit 8r + c <-> bit 8c + r): 1x1 blocks inside 2x2, 2x2 inside 4x4, 4x4 inside 8x8
    t = (lo ^ (lo >> 7)) & 0x00aa00aau; lo = lo ^ t ^ (t << 7);
    t = (hi ^ (hi >> 7)) & 0x00aa00aau; hi = hi ^ t ^ (t << 7);
    t = (lo ^ (lo >> 14)) & 0x0000ccccu; lo = lo ^ t ^ (t << 14);
    t = (hi ^ (hi >> 14)) & 0x0000ccccu; hi = hi ^ t ^ (t << 14);
    t = (lo ^ (hi << 4)) & 0xf0f0f0f0u; lo ^= t; hi ^= t >> 4;
    // byte c of lo = plane c of tokens 0..7, byte c of hi = plane c of tokens 8..15
    return Planes16{tok_perm(hi, lo, 0x05010400u), tok_perm(hi, lo, 0x07030602u)};
}

// ---- the masks of a lane -----------------------------------------------------------------------------------------
struct Lane16 {  // bit j = token j of the lane (16 bits each)
    u32 w1, w2;                 // tokens that write at least one / two tape words
    u32 br, open;               // brackets, opening brackets
    u32 nlr;                    // newlines that separate two records
    u32 a_root, a_obj, a_arr;   // token j is legal in the context (tokens behind the end: legal everywhere)
    u32 gap_start;              // the token in front of j is a bracket
    u32 str, keystr;            // strings; strings followed by ':' (object keys, for the key flags)
    u32 num, atom;              // number tokens; true / false / null
    u32 b0, b1;                 // planes 0 and 1 of the own tokens (the kind of a bracket or an atom from two bits)
    u32 valid;
};
// prev2 = kind[-2] | kind[-1] << 8 (K_NONE in front of the message), next1 = kind[N] (K_NL behind the message);
// valid = the lane's tokens that exist (a prefix of the N bits); first = token 0 of the lane is token 0 of the message.
// N = tokens the lane owns (16, or 8: the kinds of tokens 8 .. 15 are then zero and `valid` has at most eight bits)
template <int N = 16>
SJ_HD Lane16 lane16_masks(const Planes16 &pl, u32 prev2, u32 next1, u32 valid, bool first) {
    static_assert(N == 8 || N == 16, "a lane owns eight or sixteen tokens");
    // windows: bit i = token i - 2
    u32 W[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const u32 own = p & 1 ? ((p & 2 ? pl.p23 : pl.p01) >> 16) : ((p & 2 ? pl.p23 : pl.p01) & 0xffffu);
        const u32 nb = ((prev2 >> p) & 1u) | (((prev2 >> (8 + p)) & 1u) << 1) | (((next1 >> p) & 1u) << (N + 2));
        W[p] = (own << 2) | nb;
    }
    const u32 b0 = W[0], b1 = W[1], b2 = W[2], b3 = W[3];
    const u32 n0 = ~b0, n1 = ~b1, n2 = ~b2, n3 = ~b3;
    const u32 open_obj = n3 & n2 & n1 & b0, open_arr = n3 & n2 & b1 & n0, close_obj = n3 & n2 & b1 & b0, close_arr = n3 & b2 & n1 & n0;
    const u32 colon = n3 & b2 & n1 & b0, comma = n3 & b2 & b1 & n0, string = n3 & b2 & b1 & b0;
    const u32 num = b3 & n2 & n1 & n0, atom = b3 & n2 & (b1 | b0), nl = b3 & b2 & n1 & n0;
    const u32 open = open_obj | open_arr, close = close_obj | close_arr, bracket = open | close;
    // "the token in front is an X": one bit up (the neighbours are inside the window: no carries)
    const u32 p_open_obj = open_obj << 1, p_open_arr = open_arr << 1, p_close = close << 1, p_colon = colon << 1, p_comma = comma << 1;
    const u32 p_string = string << 1, p_nl = nl << 1, p_scalar = (num | atom) << 1;
    const u32 key_prev = (open_obj | comma) << 2;  // the token two in front is '{' or ',': a string in front of j is a key
    const u32 n_nl = nl >> 1;                       // the token behind is a newline
    const u32 p_end = p_close | p_scalar;           // a value ends in front of j (ends_value_v)
    const u32 p_end_obj = p_end | (p_string & ~key_prev), p_end_arr = p_end | p_string;
    const u32 value = open | num | atom;
    const u32 own = (valid & 0xffffu) << 2;
    const u32 nlr = nl & ~n_nl;
    u32 a_root = (open & p_nl) | (nl & (p_close | p_nl));
    u32 a_obj = (value & p_colon) | (string & (p_open_obj | p_comma | p_colon)) | (colon & p_string & key_prev) | (comma & p_end_obj) |
                (close_obj & (p_open_obj | p_end_obj));
    u32 a_arr = ((value | string) & (p_open_arr | p_comma)) | (comma & p_end_arr) | (close_arr & (p_open_arr | p_end_arr));
    if (first) {  // `if (i == 0) return !is_open(k)` in every context (its neighbours in front are K_NONE: nothing above is set)
        const u32 ok = open & 4u;
        a_root |= ok;
        a_obj |= ok;
        a_arr |= ok;
    }
    Lane16 m;
    m.valid = valid & 0xffffu;
    m.br = (bracket & own) >> 2;
    m.open = (open & own) >> 2;
    m.nlr = (nlr & own) >> 2;
    m.str = (string & own) >> 2;
    m.keystr = (string & (colon >> 1) & own) >> 2;
    m.num = (num & own) >> 2;
    m.atom = (atom & own) >> 2;
    m.w2 = m.str | m.num | m.nlr;
    m.w1 = m.br | m.atom | m.w2;
    const u32 inv = ~m.valid & 0xffffu;
    m.a_root = ((a_root & own) >> 2) | inv;
    m.a_obj = ((a_obj & own) >> 2) | inv;
    m.a_arr = ((a_arr & own) >> 2) | inv;
    m.gap_start = ((p_open_obj | p_open_arr | p_close) & own) >> 2;
    m.b0 = (b0 >> 2) & 0xffffu;
    m.b1 = (b1 >> 2) & 0xffffu;
    return m;
}
SJ_HD u32 popc32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__popc(x);
#else
    return (u32)__builtin_popcount(x);
#endif
}
// the composed context function of the lane, in the form of Agg::am (sj_planes.h group_function)
SJ_HD u32 lane16_function(const Lane16 &m) {
    const u32 bad_root = ~m.a_root & 0xffffu, bad_obj = ~m.a_obj & 0xffffu, bad_arr = ~m.a_arr & 0xffffu;
    if (m.gap_start == 0) return (bad_root == 0 ? 1u : 0u) | (bad_obj == 0 ? 2u : 0u) | (bad_arr == 0 ? 4u : 0u);
    const u32 from_last = ~0u << (31 - __builtin_clz(m.gap_start));  // bits from the last gap start on
    return ((bad_root & from_last) == 0 ? 8u : 0u) | ((bad_obj & from_last) == 0 ? 16u : 0u) | ((bad_arr & from_last) == 0 ? 32u : 0u);
}
// the lane's scan element in the packed in-tile form (sj_stage2.h PAgg; s = Strings.B bytes is the caller's)
SJ_HD PAgg lane16_pagg(const Lane16 &m) {
    return PAgg{popc32(m.w1) + popc32(m.w2) + (popc32(m.br) << 14), popc32(m.open) | (popc32(m.nlr) << 13), lane16_function(m), 0u};
}
// strings | scalars << 13 of the lane: the queue slots of a tile come from the same scan
SJ_HD u32 lane16_counts(const Lane16 &m) { return popc32(m.str) | (popc32(m.num | m.atom) << 13); }
// is some token of the lane legal in no context at all?
SJ_HD bool lane16_illegal(const Lane16 &m) { return (~(m.a_root | m.a_obj | m.a_arr) & m.valid) != 0; }
// tape words the lane's tokens in front of token j write
SJ_HD u32 lane16_words_before(const Lane16 &m, u32 j) {
    const u32 below = (1u << j) - 1u;
    return popc32(m.w1 & below) + popc32(m.w2 & below);
}
// allowed contexts of the gap that ends with token j (a bracket); am_in = Agg::am of everything in front of the lane
SJ_HD u32 lane16_gap_set(const Lane16 &m, u32 j, u32 am_in) {
    const u32 upto = (2u << j) - 1u;  // tokens 0 .. j
    const u32 starts = m.gap_start & upto;
    const u32 bad_root = ~m.a_root & 0xffffu, bad_obj = ~m.a_obj & 0xffffu, bad_arr = ~m.a_arr & 0xffffu;
    if (starts == 0) {  // the gap began in front of the lane
        const u32 set = ((bad_root & upto) == 0 ? 1u : 0u) | ((bad_obj & upto) == 0 ? 2u : 0u) | ((bad_arr & upto) == 0 ? 4u : 0u);
        return am_value(am_combine(am_in, set));
    }
    const u32 range = upto & (~0u << (31 - __builtin_clz(starts)));
    return ((bad_root & range) == 0 ? 1u : 0u) | ((bad_obj & range) == 0 ? 2u : 0u) | ((bad_arr & range) == 0 ? 4u : 0u);
}
// kind of the bracket at token j from planes 0 and 1: '{' 01  '[' 10  '}' 11  ']' 00
SJ_HD u8 lane16_bracket_kind(const Lane16 &m, u32 j) {
    const u32 v = ((m.b0 >> j) & 1u) | (((m.b1 >> j) & 1u) << 1);
    return v ? (u8)v : (u8)K_CLOSE_ARR;
}
// kind of the atom at token j: true 1001, false 1010, null 1011
SJ_HD u8 lane16_atom_kind(const Lane16 &m, u32 j) { return (u8)(8u | ((m.b0 >> j) & 1u) | (((m.b1 >> j) & 1u) << 1)); }

// ---- the brackets of a tile, matched inside the tile ---------------------------------------------------------------
// One 32-bit entry per bracket of a 4096-token tile, in document order:
//   bits 0-12 tape offset inside the tile | 13-26 depth BEHIND the bracket relative to the tile's start + 4096 |
//   27-28 kind - 1 | 29-31 allowed contexts of the gap that ends with it
SJ_HD u32 tbr_pack(u32 off, i32 drel, u8 kind, u32 gap) { return off | ((u32)(drel + 4096) << 13) | ((u32)(kind - 1u) << 27) | (gap << 29); }
SJ_HD u32 tbr_off(u32 e) { return e & 0x1fffu; }
SJ_HD i32 tbr_depth(u32 e) { return (i32)((e >> 13) & 0x3fffu) - 4096; }
SJ_HD u8 tbr_kind(u32 e) { return (u8)(((e >> 27) & 3u) + 1u); }
SJ_HD u32 tbr_gap(u32 e) { return e >> 29; }
// br_info of the compact bracket view: kind | gap << 4 | BR_DONE; a bracket that is DONE needs nothing from the device-wide
// matcher (its container lies in its own tile: pair words written, gap checked there)
static constexpr u8 BR_DONE = 0x80u;

}  // namespace sj
