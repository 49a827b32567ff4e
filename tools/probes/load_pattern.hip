// load_pattern.hip -- how fast can 256 persistent 1024-thread blocks stream a buffer in stage 1's tile order?
//   PAT 0: each lane loads its own 64-byte chunk (4 x 16 B, lanes 64 B apart)     -- what phase A does
//   PAT 1: each load instruction covers 1 KiB contiguously (lane l: 16 B at q*1024 + 16 l)
//   SYNC : a block barrier per tile (the tile pipeline) or free-running waves
//   DEPTH: units in flight per wave (1 = loads of the next unit issued when the current one is consumed)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/load_pattern.hip -o /tmp/load_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;

template <int PAT>
__device__ __forceinline__ void issue(const unsigned char *base, u64 unit, int lane, uint4 (&v)[4]) {
    const unsigned char *p = base + unit * 4096;
#pragma unroll
    for (int q = 0; q < 4; q++)
        v[q] = PAT == 0 ? *reinterpret_cast<const uint4 *>(p + lane * 64 + q * 16)
                        : *reinterpret_cast<const uint4 *>(p + q * 1024 + lane * 16);
}
__device__ __forceinline__ u32 eat(const uint4 (&v)[4], int math) {
    u32 x = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) x ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
    for (int i = 0; i < math; i++) x = x * 1664525u + 1013904223u;  // dependent VALU chain standing in for the math
    return x;
}

template <int PAT, int SYNC, int DEPTH>
__global__ __launch_bounds__(1024) void k(const unsigned char *base, u64 units, u32 *out, int math) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 tiles = units / 32;
    u32 acc = 0;
    uint4 a[4], b[4];
    // unit sequence of this wave: tile t = blockIdx + k * gridDim, passes 0 and 1
    u64 t = blockIdx.x;
    if (t >= tiles) return;
    issue<PAT>(base, t * 32 + wave, lane, a);
    if (DEPTH == 2) issue<PAT>(base, t * 32 + 16 + wave, lane, b);
    for (; t < tiles; t += gridDim.x) {
        const u64 tn = t + gridDim.x;
        if (DEPTH == 1) {
            uint4 c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = a[q];
            issue<PAT>(base, t * 32 + 16 + wave, lane, a);
            acc += eat(c, math);
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = a[q];
            if (tn < tiles) issue<PAT>(base, tn * 32 + wave, lane, a);
            acc += eat(c, math);
        } else {
            uint4 c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = a[q];
            if (tn < tiles) issue<PAT>(base, tn * 32 + wave, lane, a);
            acc += eat(c, math);
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = b[q];
            if (tn < tiles) issue<PAT>(base, tn * 32 + 16 + wave, lane, b);
            acc += eat(c, math);
        }
        if (SYNC) __syncthreads();
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const u64 bytes = argc > 1 ? strtoull(argv[1], 0, 10) : 269025391ull;
    const u64 units = bytes / 4096 / 32 * 32;
    unsigned char *buf; u32 *out;
    CK(hipMalloc(&buf, units * 4096)); CK(hipMemset(buf, 0x5a, units * 4096)); CK(hipMalloc(&out, 256 * 1024 * 4));
    for (int math = 0; math <= 1500; math += 750)
    for (int v = 0; v < 8; v++) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&]() {
            switch (v) {
            case 0: hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            case 1: hipLaunchKernelGGL((k<0, 1, 1>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            case 2: hipLaunchKernelGGL((k<0, 0, 2>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            case 3: hipLaunchKernelGGL((k<0, 1, 2>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            case 4: hipLaunchKernelGGL((k<1, 0, 1>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            case 5: hipLaunchKernelGGL((k<1, 1, 1>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            case 6: hipLaunchKernelGGL((k<1, 0, 2>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            case 7: hipLaunchKernelGGL((k<1, 1, 2>), dim3(256), dim3(1024), 0, 0, buf, units, out, math); break;
            }
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 10; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("math %4d  pattern %d sync %d depth %d: %.4f ms  %.0f GB/s\n", math, v >> 2, v & 1, ((v >> 1) & 1) + 1, ms, units * 4096.0 / ms / 1e6);
    }
    return 0;
}
