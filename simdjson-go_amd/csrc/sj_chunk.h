// sj_chunk.h -- per-64-byte-chunk stage-1 arithmetic for gfx950 (one chunk per lane).
//
// Everything here is lane-local VALU work on one 64-byte chunk held in 16 dwords.
// The chunk is first transposed into 8 bit-planes (bit j of plane k == bit k of byte j)
// with v_dot4_u32_u8 gathers; all byte classes are then boolean functions of the planes,
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
#else
#define SJ_HD inline
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

struct Classes {
    u64 bs;      // '\\'
    u64 quote;   // '"'
    u64 structs; // { } [ ] : ,            (find_whitespace_and_structurals_amd64.s:62-103)
    u64 ws;      // space \t \n \r
    u64 ctrl;    // byte <= 0x1f           (find_quote_mask_and_bits_amd64.s:67-78)
    u64 nl;      // '\n'                   (find_newline_delimiters_amd64.s:16-28)
};

// keeps the instruction scheduler from interleaving all eight planes (register pressure)
#if defined(__HIP_DEVICE_COMPILE__) && defined(SJ_SCHED_FENCE)
#define SJ_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SJ_FENCE() ((void)0)
#endif

SJ_HD Classes classify(const u32 (&w)[16]) {
    const u64 b0 = plane_of<0>(w), b1 = plane_of<1>(w);
    SJ_FENCE();
    const u64 b2 = plane_of<2>(w), b3 = plane_of<3>(w);
    SJ_FENCE();
    const u64 b4 = plane_of<4>(w), b5 = plane_of<5>(w);
    SJ_FENCE();
    const u64 b6 = plane_of<6>(w), b7 = plane_of<7>(w);
    SJ_FENCE();
    const u64 n0 = ~b0, n1 = ~b1, n2 = ~b2, n3 = ~b3, n4 = ~b4, n5 = ~b5, n6 = ~b6, n7 = ~b7;
    Classes c;
    const u64 hi_001 = n7 & n6 & b5;  // 0x20..0x3f
    const u64 hi_000 = n7 & n6 & n5;  // 0x00..0x1f
    const u64 hi_01x = n7 & b6;       // 0x40..0x7f
    c.ctrl = hi_000;
    // 0x22 = 0010 0010
    c.quote = hi_001 & n4 & n3 & n2 & b1 & n0;
    // 0x5c = 0101 1100
    c.bs = hi_01x & n5 & b4 & b3 & b2 & n1 & n0;
    // 0x5b 0x5d 0x7b 0x7d = 01x1 1011 / 01x1 1101
    const u64 brackets = hi_01x & b4 & b3 & b0 & (b2 ^ b1);
    // 0x2c = 0010 1100, 0x3a = 0011 1010
    const u64 comma = hi_001 & n4 & b3 & b2 & n1 & n0;
    const u64 colon = hi_001 & b4 & b3 & n2 & b1 & n0;
    c.structs = brackets | comma | colon;
    // 0x20 ; 0x09 0x0a 0x0d = 0000 1001 / 1010 / 1101
    const u64 space = hi_001 & n4 & n3 & n2 & n1 & n0;
    const u64 ctl_ws = hi_000 & n4 & b3 & ((n1 & b0) | (n2 & b1 & n0));
    c.ws = space | ctl_ws;
    c.nl = hi_000 & n4 & b3 & n2 & b1 & n0;
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
    u64 s = (structs & ~quote_mask) | quote_bits;
    const u64 pseudo_pred = s | ws;
    const u64 shifted = (pseudo_pred << 1) | (u64)pseudo_pred_in;
    s |= shifted & ~ws & ~quote_mask;
    s &= ~(quote_bits & ~quote_mask);
    return s;
}

}  // namespace sj
