// sj_chunk.h -- per-64-byte-chunk stage-1 arithmetic for gfx950 (one chunk per lane).
//
// Everything here is lane-local VALU work on one 64-byte chunk held in 16 dwords.
// The chunk is first transposed into 8 bit-planes (bit j of plane k == bit k of byte j)
// with bit butterflies and v_perm_b32 (transpose_planes; plane_of is the v_dot4 form of one plane); all byte classes are then boolean functions of the planes,
// evaluated 64 bytes at a time.  The mask algebra restates the semantics of the reference
// routines cited at each function (results must be bit-identical; the instruction
// sequences are not the reference's -- there is no PCLMUL / PSHUFB / movemask here).
//
// The functions are host+device so that the same code can be replayed on the CPU by the
// unit tests (csrc/host_selftest.cpp) without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define SJ_HD __host__ __device__ __forceinline__
#define SJ_HDC __host__ __device__ __forceinline__ constexpr
#else
#define SJ_HD inline
#define SJ_HDC inline constexpr
#endif

namespace sj {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

// ---- primitive helpers -----------------------------------------------------------------
SJ_HD u32 dot4(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    u32 s = c;
    for (int i = 0; i < 4; i++) s += ((a >> (8 * i)) & 0xff) * ((b >> (8 * i)) & 0xff);
    return s;
#endif
}
SJ_HD int popc64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
SJ_HD int ctz64(u64 x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}

SJ_HD int clz64(u64 x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

// ---- three-input boolean functions (v_bitop3_b32 on gfx950) --------------------------------
// TT is the truth table of f(a, b, c): bit (a*4 + b*2 + c) of TT is f's value.  Write TT as the same
// expression over the constants TA, TB, TC, e.g. "a & ~b | c" -> (TA & ~TB | TC) & 0xff.
static constexpr u32 TA = 0xF0, TB = 0xCC, TC = 0xAA;
template <u32 TT>
SJ_HD u32 bitop3(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT & 0xff);
#else
    u32 r = 0;
    for (int i = 0; i < 8; i++)
        if ((TT >> i) & 1u) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
#endif
}
template <u32 TT>
SJ_HD u64 bitop3(u64 a, u64 b, u64 c) {
    return ((u64)bitop3<TT>((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32)) << 32) |
           bitop3<TT>((u32)a, (u32)b, (u32)c);
}

// ---- bit-plane transposition -----------------------------------------------------------
// w[0..15] hold the chunk (little endian: byte j = (w[j>>2] >> 8*(j&3)) & 0xff).
// plane[k] bit j = bit k of byte j.
SJ_HD u32 alignbit(u32 hi, u32 lo, u32 sh) {  // ({hi,lo} >> sh)[31:0], sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (u32)((((u64)hi << 32) | lo) >> sh);
#endif
}

template <int K>
SJ_HD u64 plane_of(const u32 (&w)[16]) {
    const u32 m = 0x01010101u << K;
    u32 piece[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 a = dot4(w[2 * i] & m, 0x08040201u, 0u);
        piece[i] = dot4(w[2 * i + 1] & m, 0x80402010u, a);  // (8 plane bits) << K
    }
    // pieces are (bits << K) with bits < 256: pack pairs (no overlap), then funnel-shift the K away
    const u32 t01 = piece[0] | (piece[1] << 8), t23 = piece[2] | (piece[3] << 8);
    const u32 t45 = piece[4] | (piece[5] << 8), t67 = piece[6] | (piece[7] << 8);
    const u32 lo = alignbit(t23 >> 16, (t23 << 16) | t01, K);
    const u32 hi = alignbit(t67 >> 16, (t67 << 16) | t45, K);
    return ((u64)hi << 32) | lo;
}

// All eight planes at once, 128 instructions instead of 8 x 42.  Per half of the chunk (8 dwords, 32 bytes):
//   * 16 v_perm_b32 regroup the bytes into eight rows, row j = bytes (j, 8+j, 16+j, 24+j);
//   * three butterfly stages ACROSS the rows (distance 4, 2, 1; two shifts and two v_bitop3 selects per pair)
//     transpose the 8x8 bit matrix that the eight rows form in each of the four byte columns at once.
// Row k then is plane k of the 32 bytes in natural bit order: byte c of it holds bit k of bytes 8c .. 8c+7.
SJ_HD u32 perm_bytes(u32 hi, u32 lo, u32 sel) {  // v_perm_b32: selector bytes 0-3 pick from lo, 4-7 from hi
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const u64 both = ((u64)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; i++) r |= (u32)((both >> (8 * ((sel >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
    return r;
#endif
}
template <u32 M, int S>
SJ_HD void delta_swap_rows(u32 &a, u32 &b) {  // bits ~M of a <-> bits M of b (M << S == ~M)
    const u32 na = bitop3<((TA & TC) | (TB & ~TC))>(a, b << S, M);
    b = bitop3<((TA & ~TC) | (TB & TC))>(b, a >> S, M);
    a = na;
}
SJ_HD void transpose_planes(const u32 (&w)[16], u64 (&plane)[8]) {
    u32 r[2][8];
#pragma unroll
    for (int g = 0; g < 2; g++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {  // rows 0-3 from the even dwords of the half, rows 4-7 from the odd ones
            const u32 A = w[8 * g + h], B = w[8 * g + 2 + h], C = w[8 * g + 4 + h], D = w[8 * g + 6 + h];
            const u32 x0 = perm_bytes(B, A, 0x05010400u), x1 = perm_bytes(B, A, 0x07030602u);  // A0 B0 A1 B1 | A2 B2 A3 B3
            const u32 y0 = perm_bytes(D, C, 0x05010400u), y1 = perm_bytes(D, C, 0x07030602u);
            r[g][4 * h + 0] = perm_bytes(y0, x0, 0x05040100u);  // A0 B0 C0 D0
            r[g][4 * h + 1] = perm_bytes(y0, x0, 0x07060302u);
            r[g][4 * h + 2] = perm_bytes(y1, x1, 0x05040100u);
            r[g][4 * h + 3] = perm_bytes(y1, x1, 0x07060302u);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) delta_swap_rows<0x0f0f0f0fu, 4>(r[g][i], r[g][i + 4]);
#pragma unroll
        for (int i = 0; i < 8; i += 4) {
            delta_swap_rows<0x33333333u, 2>(r[g][i], r[g][i + 2]);
            delta_swap_rows<0x33333333u, 2>(r[g][i + 1], r[g][i + 3]);
        }
#pragma unroll
        for (int i = 0; i < 8; i += 2) delta_swap_rows<0x55555555u, 1>(r[g][i], r[g][i + 1]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) plane[k] = ((u64)r[1][k] << 32) | r[0][k];
}

struct Classes {
    u64 bs;      // '\\'
    u64 quote;   // '"'
    u64 structs; // { } [ ] : ,            (find_whitespace_and_structurals_amd64.s:62-103)
    u64 ws;      // space \t \n \r
    u64 ctrl;    // byte <= 0x1f           (find_quote_mask_and_bits_amd64.s:67-78)
    u64 nl;      // '\n'                   (find_newline_delimiters_amd64.s:16-28)
    u64 esc1;    // " \\ / b f n r t: the characters a simple escape may name (escape_map, parse_string_amd64.s:38-69);
                 // only the whole-parse kernel reads it (dead code elsewhere)
    u64 kp[4];   // bit planes of the token kind a byte starts (sj_stage2.h Kind: { 1 [ 2 } 3 ] 4 : 5 , 6 " 7 number 8
                 // t 9 f 10 n 11; 0 = no token can start here; '\n' is left out: the NDJSON kernel ORs c.nl into
                 // planes 2 and 3 for kind 12); whole-parse kernel only
};

// Every class is a conjunction of plane literals; the network below shares the common factors and
// spends one v_bitop3 per three inputs (21 per 32-bit half, +2 for the newline class).
SJ_HD Classes classify(const u32 (&w)[16]) {
    u64 pl[8];
    transpose_planes(w, pl);
    const u64 b0 = pl[0], b1 = pl[1], b2 = pl[2], b3 = pl[3], b4 = pl[4], b5 = pl[5], b6 = pl[6], b7 = pl[7];
    Classes c;
    const u64 h001 = bitop3<(~TA & ~TB & TC)>(b7, b6, b5);   // 0x20..0x3f
    const u64 h000 = bitop3<(~TA & ~TB & ~TC)>(b7, b6, b5);  // 0x00..0x1f
    const u64 t1 = bitop3<(~TA & TB & TC)>(b7, b6, b4);      // 01x1 xxxx
    const u64 m110 = bitop3<(TA & TB & ~TC)>(b3, b2, b1);    // xxxx 110x
    const u64 m1100 = m110 & ~b0;                            // xxxx 1100
    c.ctrl = h000;
    c.bs = bitop3<(TA & ~TB & TC)>(t1, b5, m1100);             // 0x5c = 0101 1100
    const u64 comma = bitop3<(TA & ~TB & TC)>(h001, b4, m1100);  // 0x2c = 0010 1100
    const u64 y = bitop3<(TA & TB & TC)>(t1, b3, b0);          // 01x1 1xx1
    const u64 brackets = bitop3<(TA & (TB ^ TC))>(y, b2, b1);  // 0x5b 0x5d 0x7b 0x7d
    const u64 c1 = bitop3<(TA & TB & TC)>(h001, b4, b3);       // 0011 1xxx
    const u64 c2 = bitop3<(TA & ~TB & TC)>(c1, b2, b1);        // 0011 101x
    const u64 s1 = brackets | comma;
    c.structs = bitop3<(TA | (TB & ~TC))>(s1, c2, b0);         // ... | 0x3a
    const u64 q1 = bitop3<(TA & ~TB & ~TC)>(h001, b4, b3);     // 0010 0xxx
    const u64 q2 = bitop3<(TA & ~TB & TC)>(q1, b2, b1);        // 0010 001x
    c.quote = q2 & ~b0;                                        // 0x22
    const u64 sp1 = bitop3<(TA & ~TB & ~TC)>(q1, b2, b1);      // 0010 000x
    const u64 w1 = bitop3<(TA & ~TB & TC)>(h000, b4, b3);      // 0000 1xxx
    // low three bits 001 (\t), 010 (\n), 101 (\r): minterms 1, 2, 5 of (b2, b1, b0)
    const u64 g = bitop3<((1u << 1) | (1u << 2) | (1u << 5))>(b2, b1, b0);
    const u64 sp = sp1 & ~b0;                                  // 0x20
    c.ws = bitop3<(TA | (TB & TC))>(sp, w1, g);
    const u64 h = bitop3<(1u << 2)>(b2, b1, b0);               // xxxx x010
    c.nl = w1 & h;                                             // 0x0a
    // " 22  \\ 5c  / 2f  b 62  f 66  n 6e  r 72  t 74
    const u64 l1111 = bitop3<(TA & TB & TC)>(b3, b2, b1);
    const u64 slash = bitop3<(TA & ~TB & TC)>(h001, b4, l1111) & b0;      // 0010 1111
    const u64 h011 = bitop3<(~TA & TB & TC)>(b7, b6, b5);                 // 011x xxxx
    const u64 s6 = bitop3<(TC & ~(TA & ~TB))>(b3, b2, b1);                // low nibble 0010 0110 1110 (b0 below)
    const u64 s7 = bitop3<(~TA & (TB ^ TC))>(b3, b2, b1);                 // low nibble 0010 0100
    const u64 t6 = bitop3<(TA & ~TB & TC)>(h011, b4, s6), t7 = bitop3<(TA & TB & TC)>(h011, b4, s7);
    const u64 letters = bitop3<((TA | TB) & ~TC)>(t6, t7, b0);
    c.esc1 = bitop3<(TA | TB | TC)>(c.quote, c.bs, slash) | letters;
    // token kinds as bit planes (the byte of a structural decides its kind; stage 2 reads them instead of the message)
    const u64 colon = c2 & ~b0;                                            // 0x3a
    const u64 base = bitop3<(TA & TB & ~TC)>(h011, b2, b0);                // 011x x1x0: t 74  f 66  n 6e
    const u64 tx = bitop3<(TA & ~TB & ~TC)>(b4, b3, b1);                   // b4 & ~b3 & ~b1
    const u64 kt = base & tx;                                              // t
    const u64 fn = bitop3<(TA & ~TB & TC)>(base, b4, b1);                  // f, n
    const u64 kn = fn & b3;                                                // n
    const u64 le9 = bitop3<(~TA | (~TB & ~TC))>(b3, b2, b1);               // low nibble <= 9
    const u64 digit = bitop3<(TA & TB & TC)>(h001, b4, le9);               // 0x30 .. 0x39
    const u64 minus = bitop3<(TA & ~TB & TC)>(h001, b4, m110) & b0;        // 0x2d = 0010 1101
    const u64 curly = brackets & b5;                                       // { }
    const u64 k0a = bitop3<(TA | TB | TC)>(curly, colon, c.quote);
    c.kp[0] = bitop3<(TA | TB | TC)>(k0a, kt, kn);                         // { } : " t n
    const u64 k1a = bitop3<(TA & ~(TB ^ TC))>(brackets, b5, b2);           // [ (b5 = 0, b2 = 0)  } (1, 1)
    c.kp[1] = bitop3<(TA | TB | TC)>(k1a, comma, c.quote) | fn;            // [ } , " f n
    const u64 k2a = bitop3<(TA & ~TB & TC)>(brackets, b5, b2);             // ]
    c.kp[2] = bitop3<(TA | TB | TC)>(k2a, colon, comma) | c.quote;         // ] : , "
    c.kp[3] = bitop3<(TA | TB | TC)>(digit, minus, kt) | fn;               // number t f n
    return c;
}

// ---- odd-length backslash runs ---------------------------------------------------------
// Semantics of find_odd_backslash_sequences_amd64.s:24-61: returns the mask of characters
// that directly follow an odd-length run of backslashes (runs may start in the previous
// chunk: carry_in = 1 iff the previous chunk ends inside an odd-length run).
// carry_out follows the reference's add-with-carry definition.
SJ_HD u64 odd_backslash_ends(u64 bs, u32 carry_in, u32 &carry_out) {
    const u64 even_bits = 0x5555555555555555ull, odd_bits = ~even_bits;
    const u64 prev = carry_in;
    const u64 start_edges = bs & ~(bs << 1);
    const u64 even_starts = start_edges & (even_bits ^ prev);
    const u64 odd_starts = start_edges & (odd_bits ^ prev);
    const u64 even_carries = bs + even_starts;
    u64 odd_carries = bs + odd_starts;
    carry_out = odd_carries < bs ? 1u : 0u;
    odd_carries |= prev;
    return ((even_carries & ~bs) & odd_bits) | ((odd_carries & ~bs) & even_bits);
}

// The same information in the form the kernels use: the mask of ESCAPED characters -- every character that
// directly follows an unescaped backslash (a "starter"): the second, fourth, ... backslash of a run and
// the character after an odd-length run.  At quote positions it equals odd_backslash_ends, so
// quote & ~escaped_mask == quote & ~odd_backslash_ends; bs & ~escaped_mask are the starters.
// carry_in = 1 iff the first character of the chunk is escaped (previous chunk ended in an odd run).
SJ_HD u64 escaped_mask(u64 bs, u32 carry_in) {
    const u64 even_bits = 0x5555555555555555ull;
    const u64 prev = carry_in;
    const u64 b = bs & ~prev;                    // an escaped backslash does not escape
    const u64 follows = (b << 1) | prev;         // characters that follow a backslash
    const u64 odd_starts = b & ~even_bits & ~follows;  // runs that begin on an odd bit
    const u64 seq_even = odd_starts + b;         // carry ripples through each such run
    const u64 invert = seq_even << 1;
    return (even_bits ^ invert) & follows;
}

// prefix XOR (the reference's VPCLMULQDQ by all-ones, find_quote_mask_and_bits_amd64.s:62-66)
SJ_HD u64 prefix_xor(u64 x) {
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    x ^= x << 32;
    return x;
}

// finalize_structurals_amd64.s:19-36 (+ ND newline OR, find_structural_bits_amd64.s:91-96)
SJ_HD u64 finalize(u64 structs, u64 ws, u64 quote_mask, u64 quote_bits, u32 pseudo_pred_in) {
    const u64 s0 = bitop3<((TA & ~TB) | TC)>(structs, quote_mask, quote_bits);  // (structs & ~qm) | quote_bits
    const u64 pseudo_pred = s0 | ws;
    const u64 shifted = (pseudo_pred << 1) | (u64)pseudo_pred_in;
    const u64 t = bitop3<(TA & ~TB & ~TC)>(shifted, ws, quote_mask);
    // (s0 | t) & ~(quote_bits & ~quote_mask): drop the closing quotes
    return bitop3<(TA & (~TB | TC))>(s0 | t, quote_bits, quote_mask);
}

// every byte 0x0a ('\n') of a word becomes 0x0d ('\r'), the other bytes stay (batch_api.hip packs documents with it)
SJ_HD u32 newlines_to_cr(u32 w) {
    const u32 x = w ^ 0x0a0a0a0au;
    const u32 z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);  // 0x80 in every byte that is '\n'
    return w ^ ((z >> 7) * 0x07u);                                          // 0x0a ^ 0x07 = 0x0d
}

}  // namespace sj
