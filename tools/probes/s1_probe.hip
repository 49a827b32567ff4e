// s1_probe.hip -- micro-probes for the stage-1 kernel design (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I simdjson-go_amd/csrc tools/probes/s1_probe.hip -o /tmp/s1_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "sj_chunk.h"
using namespace sj;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// V0: lane<->chunk loads (4 x dwordx4 per lane at stride 64), xor reduce
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void v0_lanechunk_read(const u8 *base, u64 nchunks, u32 *out) {
    const u64 c = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= nchunks) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(base + c * 64);
    uint4 a = p[0], b = p[1], d = p[2], e = p[3];
    u32 x = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ d.x ^ d.y ^ d.z ^ d.w ^ e.x ^ e.y ^ e.z ^ e.w;
    if (x == 0x12345678u) out[0] = x;
}
// V1: fully coalesced streaming read (lane reads 16 B, 4 rows of 1 KiB per wave)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void v1_coalesced_read(const u8 *base, u64 nchunks, u32 *out) {
    const u64 blockbase = (u64)blockIdx.x * BLOCK * 64;
    const uint4 *p = reinterpret_cast<const uint4 *>(base + blockbase);
    u32 x = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 a = p[k * BLOCK + threadIdx.x];
        x ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (x == 0x12345678u) out[0] = x;
}
// V2: lane<->chunk load + classify + popcount (no scans)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void v2_lanechunk_classify(const u8 *base, u64 nchunks, u32 *out) {
    const u64 c = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= nchunks) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(base + c * 64);
    u32 w[16];
#pragma unroll
    for (int k = 0; k < 4; k++) { uint4 v = p[k]; w[4*k]=v.x; w[4*k+1]=v.y; w[4*k+2]=v.z; w[4*k+3]=v.w; }
    Classes cl = classify(w);
    u32 co;
    u64 oe = odd_backslash_ends(cl.bs, 0, co);
    u64 qb = cl.quote & ~oe;
    u64 qm = prefix_xor(qb);
    u64 s = finalize(cl.structs, cl.ws, qm, qb, 1);
    u32 n = popc64(s) + popc64(cl.ctrl & qm);
    if (n == 0x12345678u) out[0] = n;
}
// V3: coalesced load -> LDS -> lane<->chunk read (b128, xor-swizzled) + classify
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void v3_lds_classify(const u8 *base, u64 nchunks, u32 *out) {
    __shared__ uint4 tile[BLOCK * 4];
    const u64 blockbase = (u64)blockIdx.x * BLOCK * 64;
    const uint4 *p = reinterpret_cast<const uint4 *>(base + blockbase);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int g = k * BLOCK + threadIdx.x;   // 16-byte granule index within the tile
        const int chunk = g >> 2, slot = g & 3;
        tile[chunk * 4 + (slot ^ ((chunk >> 2) & 3))] = p[g];
    }
    __syncthreads();
    u32 w[16];
    const int chunk = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) { uint4 v = tile[chunk * 4 + (k ^ ((chunk >> 2) & 3))]; w[4*k]=v.x; w[4*k+1]=v.y; w[4*k+2]=v.z; w[4*k+3]=v.w; }
    Classes cl = classify(w);
    u32 co;
    u64 oe = odd_backslash_ends(cl.bs, 0, co);
    u64 qb = cl.quote & ~oe;
    u64 qm = prefix_xor(qb);
    u64 s = finalize(cl.structs, cl.ws, qm, qb, 1);
    u32 n = popc64(s) + popc64(cl.ctrl & qm);
    if (n == 0x12345678u) out[0] = n;
}
// V4: lane<->chunk load + classify + flatten with the real store pattern (offsets from a fake running count)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void v4_classify_flatten(const u8 *base, u64 nchunks, u32 *out) {
    const u64 c = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= nchunks) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(base + c * 64);
    u32 w[16];
#pragma unroll
    for (int k = 0; k < 4; k++) { uint4 v = p[k]; w[4*k]=v.x; w[4*k+1]=v.y; w[4*k+2]=v.z; w[4*k+3]=v.w; }
    Classes cl = classify(w);
    u32 co;
    u64 oe = odd_backslash_ends(cl.bs, 0, co);
    u64 qb = cl.quote & ~oe;
    u64 qm = prefix_xor(qb);
    u64 s = finalize(cl.structs, cl.ws, qm, qb, 1);
    u64 o = c * 6;  // ~ average density, keeps the address pattern realistic
    while (s) { out[o++] = (u32)(c * 64) + ctz64(s); s &= s - 1; }
}

template <typename F>
static float timeit(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/tmp/c2.bin";
    FILE *f = fopen(path, "rb"); if (!f) { printf("no input %s\n", path); return 1; }
    fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<u8> h(n + 65536, 0x20); if (fread(h.data(), 1, n, f) != n) return 1; fclose(f);
    u8 *d; CK(hipMalloc(&d, h.size())); CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
    u32 *out; CK(hipMalloc(&out, (n / 64 + 1024) * 6 * 4 + (1 << 20)));
    const u64 nchunks = n / 64;
    printf("input %zu bytes, %llu chunks\n", n, (unsigned long long)nchunks);
#define RUN(name, K, B) { const u32 blocks = (u32)(nchunks / B); float ms = timeit([&] { hipLaunchKernelGGL((K<B>), dim3(blocks), dim3(B), 0, 0, d, nchunks, out); }, 20); \
        printf("%-28s block %4d : %8.4f ms  %8.1f GB/s\n", name, B, ms, (double)blocks * B * 64 / ms / 1e6); }
    RUN("v0 lanechunk read", v0_lanechunk_read, 256); RUN("v0 lanechunk read", v0_lanechunk_read, 512);
    RUN("v1 coalesced read", v1_coalesced_read, 256); RUN("v1 coalesced read", v1_coalesced_read, 512);
    RUN("v2 lanechunk classify", v2_lanechunk_classify, 256); RUN("v2 lanechunk classify", v2_lanechunk_classify, 512);
    RUN("v3 lds classify", v3_lds_classify, 256); RUN("v3 lds classify", v3_lds_classify, 512);
    RUN("v4 classify+flatten", v4_classify_flatten, 256); RUN("v4 classify+flatten", v4_classify_flatten, 512);
    return 0;
}
