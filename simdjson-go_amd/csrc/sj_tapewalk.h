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
    __shared__ long long s_m[1024];
    const u32 tid = threadIdx.x, per = (tiles + 1023u) / 1024u;
    const u32 lo = tid * per < tiles ? tid * per : tiles, hi = lo + per < tiles ? lo + per : tiles;
    long long m = -1;
    for (u32 t = lo; t < hi; t++) m = tile_last[t] > m ? tile_last[t] : m;
    s_m[tid] = m;
    __syncthreads();
    if (tid == 0) {
        long long run = -1;
        for (int k = 0; k < 1024; k++) {
            const long long v = s_m[k];
            s_m[k] = run;
            run = v > run ? v : run;
        }
    }
    __syncthreads();
    long long run = s_m[tid];
    for (u32 t = lo; t < hi; t++) {
        const long long v = tile_last[t];
        tile_last[t] = run;
        run = v > run ? v : run;
    }
}

}  // namespace
