// sj_stage2.h -- stage 2 (tape build) as data-parallel per-token functions, host+device.
//
// The reference builds the tape with a sequential goto state machine over the structural
// indexes (unifiedMachine, stage2_build_tape_amd64.go:160-446).  Here every structural index
// ("token") is handled by an independent lane; the state the machine carries is recovered from
// one prefix scan over the tokens and previous-smaller-value queries over the brackets:
//
//   depth, tape offset, Strings.B offset, record ordinal, bracket ordinal   sums of per-token elements
//   allowed contexts of the gap a token lies in                             segmented AND (segments start behind
//                                                                           brackets), carried as a bit function
//   partner of a close bracket / parent of an open bracket                  "last bracket in front with depth <= q"
//                                                                           over the compact bracket view (min tree)
//
// The grammar is a per-token rule over (kind, previous kinds, context of the innermost open container);
// it is evaluated for all three contexts at once and checked once per bracket against the context the
// bracket pass derives.  Any violation sets one global error flag; like the
// reference, a failed parse returns no tape, so only the first-violation-free prefix needs exact bookkeeping.
// csrc/host_selftest.cpp runs these functions as plain loops (the CPU test-suite checks them against the
// oracle); csrc/stage2.hip holds the kernels.
#pragma once
#include <stdint.h>

#include "sj_bounds.h"
#include "sj_chunk.h"

namespace sj {

typedef int32_t i32;
typedef int64_t i64;

enum Kind : u8 {
    K_BAD = 0,
    K_OPEN_OBJ = 1,
    K_OPEN_ARR = 2,
    K_CLOSE_OBJ = 3,
    K_CLOSE_ARR = 4,
    K_COLON = 5,
    K_COMMA = 6,
    K_STRING = 7,
    K_NUM = 8,
    K_TRUE = 9,
    K_FALSE = 10,
    K_NULL = 11,
    K_NL = 12,
    K_NONE = 15,  // no token: stands in front of the first token wherever a predecessor's kind is asked for
};
enum Ctx : u8 { CTX_ROOT = 0, CTX_OBJ = 1, CTX_ARR = 2 };

static constexpr u64 STRINGBUFBIT = 0x0080000000000000ull;  // parsed_json.go:29

SJ_HDC u8 token_kind(u8 c, bool ndjson) {
    switch (c) {
    case '{': return K_OPEN_OBJ;
    case '[': return K_OPEN_ARR;
    case '}': return K_CLOSE_OBJ;
    case ']': return K_CLOSE_ARR;
    case ':': return K_COLON;
    case ',': return K_COMMA;
    case '"': return K_STRING;
    case 't': return K_TRUE;
    case 'f': return K_FALSE;
    case 'n': return K_NULL;
    case '-': return K_NUM;
    case '\n': return ndjson ? K_NL : K_BAD;
    default: return (c >= '0' && c <= '9') ? K_NUM : K_BAD;
    }
}
SJ_HDC bool is_open(u8 k) { return k == K_OPEN_OBJ || k == K_OPEN_ARR; }
SJ_HDC bool is_close(u8 k) { return k == K_CLOSE_OBJ || k == K_CLOSE_ARR; }
SJ_HDC bool is_bracket(u8 k) { return k >= K_OPEN_OBJ && k <= K_CLOSE_ARR; }
SJ_HDC i32 depth_delta(u8 k) { return is_open(k) ? 1 : (is_close(k) ? -1 : 0); }

// number of tape words a token writes (stage2_build_tape_amd64.go: write_tape call sites).
// A newline token writes the "close root / open root" pair (:213-218) iff it is the last one of
// its run and another token follows.
SJ_HDC u32 tape_words(u8 k, u8 next_kind, bool is_last) {
    switch (k) {
    case K_OPEN_OBJ:
    case K_OPEN_ARR:
    case K_CLOSE_OBJ:
    case K_CLOSE_ARR:
    case K_TRUE:
    case K_FALSE:
    case K_NULL: return 1;
    case K_STRING:
    case K_NUM: return 2;
    case K_NL: return (!is_last && next_kind != K_NL) ? 2 : 0;
    default: return 0;
    }
}

// ---- strings: parse_string_amd64.s restated byte-serially (one string per lane) ----------------
// digittoval with the DATA-section hole (bytes < 0x30 -> 0), see oracle/sjo_parse_string.c (Q3).
SJ_HDC i32 hex_digit(u8 b) {  // selects, no control flow: the string kernels call it under divergence
    const u32 l = ((u32)b | 0x20u) - 'a', d = (u32)b - '0';
    i32 v = l < 6u ? (i32)l + 10 : -1;
    v = d < 10u ? (i32)d : v;
    return b < 0x30 ? 0 : v;
}
SJ_HDC u8 escape_value_of(u8 b) {  // escape_map, parse_string_amd64.s:38-69
    switch (b) {
    case '"': return 0x22;
    case '/': return 0x2f;
    case '\\': return 0x5c;
    case 'b': return 0x08;
    case 'f': return 0x0c;
    case 'n': return 0x0a;
    case 'r': return 0x0d;
    case 't': return 0x09;
    default: return 0;
    }
}
struct EscapeLut {
    u8 v[256];
};
constexpr EscapeLut make_escape_lut() {
    EscapeLut t{};
    for (u32 c = 0; c < 256; c++) t.v[c] = escape_value_of((u8)c);
    return t;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const EscapeLut ESCAPE_LUT = make_escape_lut();
#else
static constexpr EscapeLut ESCAPE_LUT = make_escape_lut();
#endif
SJ_HD u8 escape_value(u8 b) { return ESCAPE_LUT.v[b]; }  // one load instead of a compare chain

struct MsgView {
    Arr<const u8> p;  // (sj_bounds.h: a plain pointer in the product build)
    u64 len;
    SJ_HD u8 at(u64 i) const { return i < len ? p[i] : (u8)0; }  // the Go caller zero-pads (stage2…:75-86)
};

SJ_HD u32 hex4(const MsgView &m, u64 p) {
    const u32 d0 = (u32)hex_digit(m.at(p)), d1 = (u32)hex_digit(m.at(p + 1)), d2 = (u32)hex_digit(m.at(p + 2)),
              d3 = (u32)hex_digit(m.at(p + 3));
    return (d0 << 12) | (d1 << 8) | (d2 << 4) | d3;  // sign-extended -1 poisons the high bits
}

// ---- unaligned 8-byte access and the exact zero-byte detector used by the plain-byte fast path ------
SJ_HD u64 load_u64(const u8 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const u64 *>(p);
#else
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
#endif
}
SJ_HD void store_u64(u8 *p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<u64 *>(p) = v;
#else
    __builtin_memcpy(p, &v, 8);
#endif
}
// the low n (< 8) bytes of v
SJ_HD void store_bytes(u8 *p, u64 v, u32 n) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (n & 4u) {
        *reinterpret_cast<u32 *>(p) = (u32)v;
        p += 4;
        v >>= 32;
    }
    if (n & 2u) {
        *reinterpret_cast<uint16_t *>(p) = (uint16_t)v;
        p += 2;
        v >>= 16;
    }
    if (n & 1u) *p = (u8)v;
#else
    for (u32 i = 0; i < n; i++) p[i] = (u8)(v >> (8 * i));
#endif
}
// 0x80 in every byte of x that is zero, 0 elsewhere (exact, no borrow artefacts)
SJ_HD u64 zero_bytes(u64 x) {
    const u64 k = 0x7f7f7f7f7f7f7f7full;
    return ~(((x & k) + k) | x | k);
}

// Walks the string whose opening quote is at `q`.  If dst != nullptr the unescaped bytes are
// written there.  Returns false if the reference's _parse_string_validate_only fails.
SJ_HD bool string_walk(const MsgView &m, u64 q, u8 *dst, u32 *src_len, u32 *dst_len) {
    u64 pos = q + 1;
    u32 out = 0;
    for (;;) {
        // plain bytes, eight at a time (unaligned 8-byte loads and stores are fine on gfx950 global memory)
        while (pos + 8 <= m.len) {
            const u64 x = load_u64(arr_at(m.p, pos, 8));
            const u64 z = zero_bytes(x ^ 0x2222222222222222ull) | zero_bytes(x ^ 0x5c5c5c5c5c5c5c5cull);
            if (z == 0) {
                if (dst) store_u64(dst + out, x);
                pos += 8;
                out += 8;
                continue;
            }
            const u32 nplain = (u32)ctz64(z) >> 3;  // bytes in front of the first quote / backslash
            if (dst) store_bytes(dst + out, x, nplain);
            pos += nplain;
            out += nplain;
            break;
        }
        if (pos >= m.len) return false;  // unterminated: unreachable once stage 1 has accepted the document
        const u8 c = m.p[pos];
        if (c == '"') {
            *src_len = (u32)(pos - (q + 1));
            *dst_len = out;
            return true;
        }
        if (c != '\\') {
            if (dst) dst[out] = c;
            out++;
            pos++;
            continue;
        }
        const u8 e = m.at(pos + 1);
        if (e != 'u') {
            const u8 v = escape_value(e);
            if (v == 0) return false;
            if (dst) dst[out] = v;
            out++;
            pos += 2;
            continue;
        }
        // \uXXXX: the next raw quote must be at least 6 (12 for a surrogate pair) bytes away
        u32 d = 12;
        for (u32 j = 1; j < 12; j++)
            if (m.at(pos + j) == '"') {
                d = j;
                break;
            }
        if (d < 6) return false;
        u32 cp = hex4(m, pos + 2);
        u64 next = pos + 6;
        if ((cp & 0xfffffc00u) == 0xd800u) {
            if (d < 12) return false;
            if (m.at(pos + 6) != '\\' || m.at(pos + 7) != 'u') return false;
            const u32 cp2 = hex4(m, pos + 8);
            if ((cp | cp2) > 0xffffu) return false;
            cp = (((cp << 10) + 0xfca00000u) | (cp2 + 0xffff2400u)) + 0x10000u;  // 32-bit wrap-around, low half unchecked
            next = pos + 12;
        }
        if (cp < 0x80u) {
            if (dst) dst[out] = (u8)cp;
            out += 1;
        } else if (cp < 0x800u) {
            if (dst) {
                dst[out] = (u8)((cp >> 6) + 192);
                dst[out + 1] = (u8)((cp & 63) | 128);
            }
            out += 2;
        } else if (cp < 0x10000u) {
            if (dst) {
                dst[out] = (u8)((cp >> 12) + 224);
                dst[out + 1] = (u8)(((cp >> 6) & 63) | 128);
                dst[out + 2] = (u8)((cp & 63) | 128);
            }
            out += 3;
        } else if (cp <= 0x10ffffu) {
            if (dst) {
                dst[out] = (u8)((cp >> 18) + 240);
                dst[out + 1] = (u8)(((cp >> 12) & 63) | 128);
                dst[out + 2] = (u8)(((cp >> 6) & 63) | 128);
                dst[out + 3] = (u8)((cp & 63) | 128);
            }
            out += 4;
        } else {
            return false;
        }
        pos = next;
    }
}

// ---- atoms (stage2_build_tape_amd64.go:124-158, table :455-476) ----------------------------------
SJ_HD bool atom_terminator(u8 c) {  // 0 \t \n \r space , : [ ] { }   (a bit set: no control flow under divergence)
    const u64 LO = (1ull << 0) | (1ull << 9) | (1ull << 10) | (1ull << 13) | (1ull << 32) | (1ull << 44) | (1ull << 58);
    const u64 HI = (1ull << ('[' - 64)) | (1ull << (']' - 64)) | (1ull << ('{' - 64)) | (1ull << ('}' - 64));
    const u64 m = (c & 64u) ? HI : LO;
    return (c < 128u) & (bool)((m >> (c & 63u)) & 1u);
}
// w8 = the 8 message bytes at the token (little endian, zero beyond the end), rem = bytes from the token to the
// end of the message
SJ_HD bool atom_valid_word(u64 w8, u64 rem, u8 kind) {
    const u32 want = kind == K_TRUE ? 0x65757274u : (kind == K_NULL ? 0x6c6c756eu : 0x736c6166u);  // true null fals
    const bool head = (u32)w8 == want;
    const u8 b4 = (u8)(w8 >> 32), b5 = (u8)(w8 >> 40);
    const bool ok5 = (rem >= 5) & head & atom_terminator(b4);
    const bool ok6 = (rem >= 6) & head & (b4 == 'e') & atom_terminator(b5);
    return kind == K_FALSE ? ok6 : ok5;
}
SJ_HD u64 load8_guarded(const MsgView &m, u64 p) {
    if (p + 8 <= m.len) return load_u64(arr_at(m.p, p, 8));
    u64 v = 0;
    for (u32 k = 0; k < 8 && p + k < m.len; k++) v |= (u64)m.p[p + k] << (8 * k);
    return v;
}
SJ_HD bool atom_valid(const MsgView &m, u64 p, u8 kind) { return atom_valid_word(load8_guarded(m, p), m.len - p, kind); }

// ---- previous-smaller-value over depth[] with a 64-ary min tree ------------------------------------
struct MinTree {
    static constexpr int MAXLEV = 7;
    Arr<const i32> lev[MAXLEV];  // lev[0] = depth[], lev[k][g] = min of lev[k-1][64g .. 64g+63]
    u64 size[MAXLEV];
    int nlev;
};

// last k < i with depth[k] < tau, or -1
SJ_HD i64 psv(const MinTree &t, i64 i, i32 tau) {
    if (i <= 0) return -1;
    i64 k = i - 1;
    int L = 0;
    for (;;) {
        const i64 gstart = (k >> 6) << 6;
        for (; k >= gstart; k--)
            if (t.lev[L][k] < tau) goto found;
        if (gstart == 0 || L + 1 >= t.nlev) return -1;
        k = (gstart >> 6) - 1;
        L++;
    }
found:
    while (L > 0) {
        L--;
        const i64 base = k << 6;
        i64 hi = base + 63;
        if ((u64)hi >= t.size[L]) hi = (i64)t.size[L] - 1;
        for (k = hi; k > base; k--)
            if (t.lev[L][k] < tau) break;
    }
    return k;
}

// ---- grammar ------------------------------------------------------------------------------------------
// a string token is an object key iff it sits in an object right after '{' or ','
SJ_HDC bool string_is_key_v(u8 ctx, bool has_prev, u8 prev_kind) {
    return ctx == CTX_OBJ && has_prev && (prev_kind == K_OPEN_OBJ || prev_kind == K_COMMA);
}
// does a token of kind k (whose predecessor has kind prev_kind, if any) end a value of the container with context ctx?
SJ_HDC bool ends_value_v(u8 k, u8 ctx, bool has_prev, u8 prev_kind) {
    if (k == K_CLOSE_OBJ || k == K_CLOSE_ARR || k == K_NUM || k == K_TRUE || k == K_FALSE || k == K_NULL) return true;
    if (k == K_STRING) return !string_is_key_v(ctx, has_prev, prev_kind);
    return false;
}
// Grammar check of token i (true = violation): k = its kind, pk / ppk = kinds of tokens i-1 / i-2 (if they
// exist), G = context of the gap in front of it (the innermost open container).  Mirrors the transitions of
// unifiedMachine: continueRoot/startContinue (:176-221), object_begin/object_key_state/objectContinue (:225-325),
// arrayBegin/mainArraySwitch/arrayContinue (:346-426).
SJ_HDC bool grammar_violation_v(u32 i, u8 k, u8 pk, u8 ppk, u8 G) {
    if (k == K_BAD) return true;
    if (i == 0) return !is_open(k);
    const bool has_pp = i > 1;
    switch (k) {
    case K_OPEN_OBJ:
    case K_OPEN_ARR:
        if (G == CTX_ROOT) return pk != K_NL;
        // fallthrough: a container is a value
    case K_NUM:
    case K_TRUE:
    case K_FALSE:
    case K_NULL:
        if (G == CTX_OBJ) return pk != K_COLON;
        if (G == CTX_ARR) return !(pk == K_OPEN_ARR || pk == K_COMMA);
        return true;
    case K_STRING:
        if (G == CTX_OBJ) return !(pk == K_OPEN_OBJ || pk == K_COMMA || pk == K_COLON);
        if (G == CTX_ARR) return !(pk == K_OPEN_ARR || pk == K_COMMA);
        return true;
    case K_COLON:
        return !(G == CTX_OBJ && pk == K_STRING && string_is_key_v(G, has_pp, ppk));
    case K_COMMA:
        return !(G != CTX_ROOT && ends_value_v(pk, G, has_pp, ppk));
    case K_CLOSE_OBJ:
        return !(G == CTX_OBJ && (pk == K_OPEN_OBJ || ends_value_v(pk, G, has_pp, ppk)));
    case K_CLOSE_ARR:
        return !(G == CTX_ARR && (pk == K_OPEN_ARR || ends_value_v(pk, G, has_pp, ppk)));
    case K_NL:
        return !(G == CTX_ROOT && (is_close(pk) || pk == K_NL));
    default:
        return true;
    }
}

// The context G is only known once the brackets are matched, so the per-token pass evaluates the rule for all
// three contexts at once: allowed_contexts = { G : token i is legal in context G } as a bit set (bit CTX_*).
// The rule reads i only as "first token" / "has two predecessors" and ppk only as "the token before the
// previous one makes a following string a key".  With K_NONE standing for the missing predecessors of the
// first two tokens that makes it a 512-entry table:
//   index = k | pk << 4 | key_prev << 8         (pk == K_NONE: first token)
// All tokens between two brackets share one context, so the AND of their sets (a segmented scan, below) is
// checked once per bracket against the context the bracket pass derives.
static constexpr u32 LUT_SIZE = 512;
SJ_HDC u32 grammar_lut_index(u8 k, u8 pk, u8 ppk) {
    return (u32)k | ((u32)pk << 4) | ((ppk == K_OPEN_OBJ || ppk == K_COMMA) ? 256u : 0u);
}
SJ_HDC u8 allowed_contexts_at(u32 index) {
    const u8 k = (u8)(index & 15u), pk = (u8)((index >> 4) & 15u);
    const bool key_prev = (index >> 8) & 1u, first = pk == K_NONE;
    const u32 i = first ? 0u : 2u;
    const u8 ppk = key_prev ? (u8)K_COMMA : (u8)K_BAD;
    u8 m = 0;
    for (u8 G = 0; G < 3; G++)
        if (!grammar_violation_v(i, k, pk, ppk, G)) m = (u8)(m | (1u << G));
    return m;
}
struct KindLut {
    u8 v[256];  // token_kind(byte, ndjson = true); '\n' is demoted to K_BAD by the caller for plain JSON
};
constexpr KindLut make_kind_lut() {
    KindLut t{};
    for (u32 c = 0; c < 256; c++) t.v[c] = token_kind((u8)c, true);
    return t;
}
// The same index also selects everything else a token contributes to the scan except what depends on the
// NEXT token (a newline writes its root pair only as the last one of its run): one 32-bit entry per index,
//   bits 0-1 tape words (newline: 0) | bit 14 bracket | bit 24 open bracket | bits 26-31 the context function
// i.e. the packed in-tile form of the scan element (PAgg below) in one word.
struct ElementLut {
    u32 v[LUT_SIZE];
};
constexpr ElementLut make_element_lut() {
    ElementLut t{};
    for (u32 x = 0; x < LUT_SIZE; x++) {
        const u8 k = (u8)(x & 15u), pk = (u8)((x >> 4) & 15u);
        const bool first = pk == K_NONE;
        const u32 w = k == K_NL ? 0u : tape_words(k, (u8)K_BAD, false);
        const u32 A = allowed_contexts_at(x);
        const u32 z = (!first && is_bracket(pk)) ? A << 3 : A;  // a gap starts behind a bracket
        t.v[x] = w | (is_bracket(k) ? 1u << 14 : 0u) | (is_open(k) ? 1u << 24 : 0u) | (z << 26);
    }
    return t;
}

// ---- the device-wide scan ---------------------------------------------------------------------------------
// One element per token; the inclusive/exclusive prefixes give every token its depth, tape offset, Strings.B
// offset (only when strings are copied selectively), record ordinal, compact bracket index, and the allowed
// contexts of the gap it lies in.  The last one is a segmented AND (segments start behind brackets); it is
// carried as the function it applies to the set v in front of it, v -> (v & p) | q, stored as am = p | q << 3:
// a token inside a gap is (A, 0), the first token of a gap is (0, A), and functions compose with two bit
// operations (am_combine).  Applied to "all contexts" it yields the set itself: am_value.
struct Agg {
    i32 d;
    u32 w, s, nb, bc, am;
};
static constexpr u32 AM_ALL = 7u;  // also the identity function (p = all, q = none)
SJ_HD Agg agg_identity() { return Agg{0, 0u, 0u, 0u, 0u, AM_ALL}; }
SJ_HD u32 am_combine(u32 a, u32 b) {  // a in front of b
    const u32 bp = b & 7u;
    return (a & (bp | (bp << 3))) | (b & 0x38u);
}
SJ_HD u32 am_value(u32 am) { return (am | (am >> 3)) & AM_ALL; }
SJ_HD Agg agg_combine(const Agg &a, const Agg &b) {  // a in front of b
    return Agg{a.d + b.d, a.w + b.w, a.s + b.s, a.nb + b.nb, a.bc + b.bc, am_combine(a.am, b.am)};
}
// element of token i of n.  k / pk / ppk / nk: kinds of tokens i, i-1, i-2, i+1 (K_BAD where there is none);
// copied: bytes the token appends to Strings.B through the scan (0 when the emit masks place the strings)
SJ_HD Agg token_element(u32 i, u32 n, u8 k, u8 pk, u8 ppk, u8 nk, u32 copied) {
    const bool last = i + 1 == n;
    Agg a;
    a.d = depth_delta(k);
    a.w = tape_words(k, nk, last);
    a.s = copied;
    a.nb = (k == K_NL && !last && nk != K_NL) ? 1u : 0u;
    a.bc = is_bracket(k) ? 1u : 0u;
    u32 allowed = 0;
    for (u8 G = 0; G < 3; G++)
        if (!grammar_violation_v(i, k, pk, ppk, G)) allowed |= 1u << G;
    a.am = (i > 0 && is_bracket(pk)) ? allowed << 3 : allowed;  // a gap starts behind a bracket
    return a;
}
// allowed contexts of the gap that ends with token i (x = exclusive prefix, e = its element)
SJ_HD u32 gap_mask(const Agg &x, const Agg &e) { return am_value(am_combine(x.am, e.am)); }

// ---- packed form of the scan inside one 4096-token tile ----------------------------------------------------------
// Inside a tile every quantity fits a few bits:
//   x = w (14 bits) | bc << 14 (13 bits)      y = opens (13 bits) | nb << 13      z = am      s = Strings.B bytes
// (depth = 2 * opens - brackets); x, y and s simply add.
struct PAgg {
    u32 x, y, z, s;
};
SJ_HD PAgg pagg_identity() { return PAgg{0u, 0u, AM_ALL, 0u}; }
SJ_HD PAgg pagg_pack(const Agg &a) {
    return PAgg{a.w | (a.bc << 14), (u32)((a.d + (i32)a.bc) >> 1) | (a.nb << 13), a.am, a.s};
}
SJ_HD Agg pagg_unpack(const PAgg &v) {
    const u32 bc = v.x >> 14, op = v.y & 0x1fffu;
    return Agg{(i32)(2u * op) - (i32)bc, v.x & 0x3fffu, v.s, v.y >> 13, bc, v.z};
}
SJ_HD PAgg pagg_combine(const PAgg &a, const PAgg &b) {  // a in front of b
    return PAgg{a.x + b.x, a.y + b.y, am_combine(a.z, b.z), a.s + b.s};
}
// Packed scan element of a token from the kinds (ppk, pk, k, nk) in the four bytes of `win`, through the table
// (token_element is the reference form; the table is generated from the same rules).  Missing neighbours are
// sentinels: K_NONE in front of the first token, K_NL behind the last one (a newline run at the very end
// writes no root pair).
SJ_HD PAgg token_pelement(const u32 *elut, u32 win, u32 copied) {
    const u32 ppk = win & 0xffu, pk = (win >> 8) & 0xffu, k = (win >> 16) & 0xffu, nk = win >> 24;
    const u32 e = elut[grammar_lut_index((u8)k, (u8)pk, (u8)ppk)];
    u32 x = e & 0x4003u, y = (e >> 24) & 1u;
    if (k == K_NL && nk != K_NL) {  // the last newline of a run separates two records: root pair
        x += 2u;
        y |= 1u << 13;
    }
    return PAgg{x, y, e >> 26, copied};
}
// the kinds around token i of a kind array, with the sentinels
SJ_HD u32 kind_window(const u8 *kind, u32 i, u32 n) {
    const u32 ppk = i >= 2 ? kind[i - 2] : (u32)K_NONE, pk = i >= 1 ? kind[i - 1] : (u32)K_NONE;
    const u32 nk = i + 1 < n ? kind[i + 1] : (u32)K_NL;
    return ppk | (pk << 8) | ((u32)kind[i] << 16) | (nk << 24);
}

// ---- brackets: partners, contexts and root words over the compact bracket view -------------------------------------
// The c-th bracket token has depth br_depth[c] after it (level 0 of the min tree), its tape word at br_off[c]
// and br_info[c] = kind | gap_mask << 4 (the gap that ends with it).  Non-bracket tokens never change the
// depth, so previous-smaller-value queries over the compact view find the same brackets as over all tokens.
// One rule for every bracket: the gap that ends with it lies at the depth in front of it, and the container that owns
// that level is the bracket behind the last one in front with depth <= (depth in front - 1) -- the partner of a close,
// the parent of an open; the type of that container (the root context if the depth in front is not positive) must be in
// the set of contexts the gap allows.  A close also writes both tape words of its pair (payloads:
// annotate_previousloc, stage2_build_tape_amd64.go:335-336), and a pair at depth 0 -- a record, or the whole document
// -- the root words around it: the open-root word in front of the open bracket points behind the close-root word that
// follows the close bracket, and that one back at the open-root word (startContinue :196-221, succeed :428-442).
// Returns false on a context violation.
SJ_HD bool context_allowed(u32 mask, u8 ctx) { return (mask >> ctx) & 1u; }
SJ_HD bool bracket_resolve(const MinTree &mt, const u32 *br_off, const u8 *br_info, u32 c, u64 tape_base, u64 *tape) {
    const u8 k = br_info[c] & 15u;
    const u32 gap = (u32)(br_info[c] >> 4);
    const bool close = is_close(k);
    const i32 d = mt.lev[0][c];
    const i32 q = close ? d : d - 2;  // depth in front of the bracket - 1
    if (q < 0) return context_allowed(gap, CTX_ROOT);  // nothing is open in front of it (a close is rejected: it needs OBJ / ARR)
    const i64 j = psv(mt, (i64)c, q + 1) + 1;  // the open bracket that raised the depth to q + 1
    const u8 jk = br_info[j] & 15u;
    if (close) {
        const u64 oc = br_off[c], oj = br_off[j];
        tape[oc] = ((u64)(k == K_CLOSE_OBJ ? '}' : ']') << 56) | (tape_base + oj);
        tape[oj] = ((u64)(jk == K_OPEN_OBJ ? '{' : '[') << 56) | (tape_base + oc + 1);
        if (d == 0) {
            tape[oj - 1] = ((u64)'r' << 56) | (tape_base + oc + 2);
            tape[oc + 1] = ((u64)'r' << 56) | (tape_base + oj - 1);
        }
    }
    return context_allowed(gap, jk == K_OPEN_OBJ ? (u8)CTX_OBJ : (u8)CTX_ARR);
}

// ---- atoms and strings: tape words ---------------------------------------------------------------------------
SJ_HD u64 atom_word(u8 k) { return (u64)(k == K_TRUE ? 't' : (k == K_FALSE ? 'f' : 'n')) << 56; }
// parseString (stage2_build_tape_amd64.go:72-113): a copied string points into Strings.B (bit 55 set), the
// others into the message
SJ_HD u64 string_word(bool copied, u64 strings_off, u64 msg_off) {
    return ((u64)'"' << 56) | (copied ? STRINGBUFBIT + strings_off : msg_off);
}

}  // namespace sj
