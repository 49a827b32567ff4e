// sj_tapewalk.h -- device helpers shared by the kernels that walk a finished tape (serialize.hip, marshal.hip):
// 2048-word tiles, block-wide scans, and the tag / raw-word classification.
//
// A tape entry is one word (brackets, roots, atoms) or two (strings: tag + length, numbers: tag + value).  The second
// word is raw 64-bit data whose top byte can look like any tag, so "is this word a tag?" is not a local question.
// Raw words only ever follow a two-word tag (" l u d) that is itself not raw; with c(i) = "the top byte of word i is
// one of \" l u d" and p = the last index below i with c(p) = 0 (an anchor; word 0, the opening root, is one):
//     word i is raw  <=>  i - p - 1 is odd.
// k_tw_last / k_tw_scan_last give every tile the last anchor in front of it; inside a tile a block max-scan does it.
// (Everything lives in an anonymous namespace: each .hip file that includes this header gets its own copy.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sj_stage2.h"

namespace {
using namespace sj;

static constexpr u64 TW_PAYLOAD = 0x00ffffffffffffffull;  // JSONVALUEMASK, parsed_json.go:27
static constexpr int TW_THREADS = 256, TW_ITEMS = 8, TW_TILE = TW_THREADS * TW_ITEMS;

__device__ __forceinline__ bool two_word_tag(u64 w) {
    const u32 t = (u32)(w >> 56);
    return t == '"' || t == 'l' || t == 'u' || t == 'd';
}

// block-wide exclusive scans over one value per thread (4 waves)
__device__ __forceinline__ long long block_excl_max(long long v, long long *s_w, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    long long incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const long long o = __shfl_up(incl, s, 64);
        if (lane >= s) incl = o > incl ? o : incl;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    long long before = -1;
    for (int w = 0; w < wave; w++) before = s_w[w] > before ? s_w[w] : before;
    long long ex = __shfl_up(incl, 1, 64);
    if (lane == 0) ex = -1;
    __syncthreads();
    return ex > before ? ex : before;
}
__device__ __forceinline__ unsigned long long block_excl_sum(unsigned long long v, unsigned long long *s_w, int tid,
                                                              unsigned long long *total) {
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned long long o = (unsigned long long)__shfl_up((long long)incl, s, 64);
        if (lane >= s) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned long long before = 0, tot = 0;
    for (int w = 0; w < TW_THREADS / 64; w++) {
        if (w < wave) before += s_w[w];
        tot += s_w[w];
    }
    if (total) *total = tot;
    __syncthreads();
    return before + incl - v;
}

// The anchor in front of the tile that starts at word tb, without the global pass (k_tw_last / k_tw_scan_last): the
// closest of the 64 words in front of the tile that is not a two-word tag.  -1: the tile is the first one; -2: none
// among the 64 (they are raw words that all look like string / number tags) -- the tile must then not classify
// anything (with a wrong anchor raw words would be read as tags and their neighbours as string offsets and lengths):
// it reports through a flag and the host repeats the walk with the global anchors.  Block-uniform result.
__device__ __forceinline__ long long tw_local_anchor(const u64 *tape, u64 tb, int tid, long long *s_carry) {
    if (tid < 64) {
        const bool have = tb >= 1 + (u64)tid;
        const bool c0 = have && !two_word_tag(tape[tb - 1 - (u64)tid]);
        const u64 b = __ballot(c0);
        if (tid == 0) *s_carry = b ? (long long)(tb - 1 - (u64)ctz64(b)) : (tb > 64 ? -2ll : -1ll);
    }
    __syncthreads();
    return *s_carry;
}

// In-place exclusive scan of a[0 .. n) by ONE 1024-thread block (the per-tile values of a walk: tens of thousands of
// elements).  Every wave owns a contiguous range and walks it 64 consecutive elements at a time (coalesced loads, a
// shuffle scan per step): pass 1 reduces the range, the 16 range totals meet in LDS, pass 2 scans and writes.
// MAX: running maximum (identity -1), else sum.  (The first version gave every THREAD a contiguous range: 64 different
// cache lines per load instruction, 180 us for 39 000 tiles.)
template <bool MAX>
__device__ __forceinline__ long long block1024_scan_array(long long *a, u32 n, long long *s_w /* [16] */, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const long long id = MAX ? -1ll : 0ll;
    auto op = [](long long x, long long y) { return MAX ? (x > y ? x : y) : x + y; };
    const u32 per = ((n + 15u) / 16u + 63u) / 64u * 64u;
    const u32 lo = (u32)wave * per < n ? (u32)wave * per : n, hi = lo + per < n ? lo + per : n;
    long long acc = id;
    for (u32 c0 = lo; c0 < hi; c0 += 64 * 8) {  // (eight loads in flight: one at a time is a memory round trip per 64 entries)
        long long v8[8];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const u32 i = c0 + (u32)g * 64u + (u32)lane;
            v8[g] = i < hi ? a[i] : id;
        }
#pragma unroll
        for (int g = 0; g < 8; g++) acc = op(acc, v8[g]);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc = op(acc, __shfl_xor(acc, s, 64));
    if (lane == 0) s_w[wave] = acc;
    __syncthreads();
    long long carry = id, total = id;
    for (int w = 0; w < 16; w++) {
        const long long x = s_w[w];
        if (w < wave) carry = op(carry, x);
        total = op(total, x);
    }
    for (u32 c0 = lo; c0 < hi; c0 += 64 * 8) {  // eight groups of 64 at a time: their loads are one round trip, not eight
        long long v8[8];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const u32 i = c0 + (u32)g * 64u + (u32)lane;
            v8[g] = i < hi ? a[i] : id;
        }
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const u32 i = c0 + (u32)g * 64u + (u32)lane;
            long long incl = v8[g];
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const long long o = __shfl_up(incl, s, 64);
                if (lane >= s) incl = op(o, incl);
            }
            long long ex = __shfl_up(incl, 1, 64);
            if (lane == 0) ex = id;
            if (i < hi) a[i] = op(carry, ex);
            carry = op(carry, __shfl(incl, 63, 64));
        }
    }
    __syncthreads();  // (s_w may be used again)
    return total;
}

__global__ __launch_bounds__(TW_THREADS) void k_tw_last(const u64 *tape, u64 n, long long *tile_last) {
    __shared__ long long s_w[TW_THREADS / 64];
    const int tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * TW_TILE + (u64)tid * TW_ITEMS;
    long long last = -1;
#pragma unroll
    for (int k = 0; k < TW_ITEMS; k++)
        if (base + k < n && !two_word_tag(tape[base + k])) last = (long long)(base + k);
    // block maximum
    const int lane = tid & 63;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const long long o = __shfl_xor(last, s, 64);
        last = o > last ? o : last;
    }
    if (lane == 0) s_w[tid >> 6] = last;
    __syncthreads();
    if (tid == 0) {
        long long m = s_w[0];
        for (int w = 1; w < TW_THREADS / 64; w++) m = s_w[w] > m ? s_w[w] : m;
        tile_last[blockIdx.x] = m;
    }
}

// one block: tile_last[t] := the last anchor in front of tile t (exclusive running maximum)
__global__ __launch_bounds__(1024) void k_tw_scan_last(long long *tile_last, u32 tiles) {
    __shared__ long long s_w[16];
    block1024_scan_array<true>(tile_last, tiles, s_w, (int)threadIdx.x);
}

}  // namespace
