// mixed_stream.hip -- what does this part sustain for stage 1's TRAFFIC MIX?  256 persistent 1024-thread blocks read a buffer in
// stage 1's tile order (a lane loads its own 64-byte chunk, one 4 KiB unit per wave in flight) and write W bytes of output per
// 4 KiB unit read, contiguously per wave with 16-byte stores (what the flatten's copy-out does) -- no math, no look-back, no
// barrier.  Stage 1 writes 4 S / N bytes per input byte: 0.35 on twitter.json (W = 1434), 0.86 on parking-citations ND (W = 3520);
// a copy is W = 4096.  Prints GB/s of input, of output and of both for three buffer sizes; NT = streaming stores.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mixed_stream.hip -o /tmp/mixed_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;

template <bool NT>
__global__ __launch_bounds__(1024) void k(const unsigned char *base, u64 units, unsigned char *out, u32 wbytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 tiles = units / 32;
    uint4 a[4];
    auto issue = [&](u64 unit) {
        const unsigned char *p = base + unit * 4096 + lane * 64;
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] = *reinterpret_cast<const uint4 *>(p + q * 16);
    };
    u64 t = blockIdx.x;
    if (t >= tiles) return;
    issue(t * 32 + wave);
    for (; t < tiles; t += gridDim.x) {
        for (int pass = 0; pass < 2; pass++) {
            const u64 unit = t * 32 + pass * 16 + wave;
            uint4 c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = a[q];
            const u64 tn = pass == 0 ? t : t + gridDim.x;
            if (tn < tiles) issue(tn * 32 + (pass == 0 ? 16 : 0) + wave);
            // the unit's output: wbytes bytes at unit * wbytes, 16 bytes per lane per round
            unsigned char *o = out + unit * (u64)wbytes;
            const uint4 v = make_uint4(c[0].x ^ c[1].y, c[2].z ^ c[3].w, c[0].w ^ c[2].x, c[1].z ^ c[3].y);
            for (u32 i = (u32)lane * 16u; i + 16u <= wbytes; i += 1024u) {
                if (NT) {
                    u32 *d = reinterpret_cast<u32 *>(o + i);
                    __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1);
                    __builtin_nontemporal_store(v.z, d + 2); __builtin_nontemporal_store(v.w, d + 3);
                } else {
                    *reinterpret_cast<uint4 *>(o + i) = v;
                }
            }
        }
    }
}

int main() {
    const u64 sizes[3] = {67572106ull, 269025391ull, 1073575501ull};
    const u32 ws[4] = {0u, 1424u, 3520u, 4096u};
    unsigned char *buf, *out;
    CK(hipMalloc(&buf, sizes[2] + 4096)); CK(hipMemset(buf, 0x5a, sizes[2] + 4096)); CK(hipMalloc(&out, sizes[2] + 4096));
    for (int s = 0; s < 3; s++) {
        const u64 units = sizes[s] / 4096 / 32 * 32;
        for (int w = 0; w < 4; w++)
            for (int nt = 0; nt < 2; nt++) {
                if (ws[w] == 0 && nt) continue;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                auto launch = [&]() {
                    if (nt) hipLaunchKernelGGL((k<true>), dim3(256), dim3(1024), 0, 0, buf, units, out, ws[w]);
                    else hipLaunchKernelGGL((k<false>), dim3(256), dim3(1024), 0, 0, buf, units, out, ws[w]);
                };
                for (int i = 0; i < 3; i++) launch();
                CK(hipEventRecord(e0, 0));
                const int reps = 20;
                for (int i = 0; i < reps; i++) launch();
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                const double in = (double)units * 4096, o = (double)units * ws[w];
                printf("in %6.0f MB  out/unit %4u B (%.2f of input) %s  %.4f ms  in %6.0f GB/s  out %6.0f GB/s  total %6.0f GB/s\n",
                       in / 1e6, ws[w], ws[w] / 4096.0, nt ? "nt   " : "plain", ms, in / ms / 1e6, o / ms / 1e6, (in + o) / ms / 1e6);
            }
    }
    return 0;
}
