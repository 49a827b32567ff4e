// phasea_rate.hip -- VALU-only throughput of the per-chunk stage-1 math (no memory in the loop).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I simdjson-go_amd/csrc tools/probes/phasea_rate.hip -o tools/probes/phasea_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "sj_chunk.h"
using namespace sj;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const u32 *in, u32 *out, int iters) {
    u32 w[16];
    for (int j = 0; j < 16; j++) w[j] = in[(threadIdx.x * 16 + j) & 4095];
    u32 acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) w[j] ^= acc + j;   // 16 VALU, keeps the loop body alive
        const Classes c = classify(w);
        if (MODE == 0) {
            acc += (u32)popc64(c.bs ^ c.quote ^ c.structs ^ c.ws ^ c.ctrl ^ c.nl);
        } else {
            u32 co;
            const u64 qb = c.quote & ~odd_backslash_ends(c.bs, acc & 1, co);
            u64 qm = prefix_xor(qb);
            u64 a = finalize(c.structs, c.ws, qm, qb, acc & 1);
            u64 b = finalize(c.structs, c.ws, ~qm, qb, acc & 1);
            acc += (u32)popc64(a) + ((u32)popc64(b) << 16) + (u32)popc64(c.ctrl & qm) + (u32)popc64(c.nl) + co;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    u32 *in, *out;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMemset(in, 0x5a, 4096 * 4)); CK(hipMalloc(&out, 4 << 20));
    const int iters = 400;
    for (int mode = 0; mode < 2; mode++)
        for (int wps = 1; wps <= 8; wps *= 2) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&](int it) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, in, out, it);
                                        else hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, in, out, it); };
            launch(5); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // chunk-iterations per SIMD = wps * iters ; report ns per chunk-iteration per SIMD and the stage-1 GB/s this VALU rate would allow
            const double ns = ms * 1e6 / ((double)wps * iters);
            printf("mode %d  %d waves/SIMD: %.3f ms  %.1f ns per wave-chunk per SIMD  -> %.0f GB/s if all 1024 SIMDs did only this\n",
                   mode, wps, ms, ns, 4096.0 / ns * 1024);
        }
    return 0;
}
