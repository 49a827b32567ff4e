// unaligned.hip -- does gfx950 global memory take unaligned 8-byte loads/stores? (probe, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void k(const unsigned char *src, unsigned char *dst, int so, int d_o) {
    const int i = threadIdx.x;
    const unsigned long long v = *reinterpret_cast<const unsigned long long *>(src + so + 8 * i);
    *reinterpret_cast<unsigned long long *>(dst + d_o + 8 * i) = v;
}
int main() {
    unsigned char h[2048], o[2048], *s, *d;
    for (int i = 0; i < 2048; i++) h[i] = (unsigned char)(i * 7 + 3);
    hipMalloc(&s, 2048); hipMalloc(&d, 2048);
    hipMemcpy(s, h, 2048, hipMemcpyHostToDevice);
    int bad = 0;
    for (int so = 0; so < 8; so++) for (int d_o = 0; d_o < 8; d_o++) {
        hipMemset(d, 0, 2048);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, so, d_o);
        if (hipDeviceSynchronize() != hipSuccess) { printf("fault at %d %d\n", so, d_o); return 1; }
        hipMemcpy(o, d, 2048, hipMemcpyDeviceToHost);
        if (memcmp(o + d_o, h + so, 512) != 0) { bad++; printf("mismatch so=%d do=%d\n", so, d_o); }
    }
    printf("unaligned 8-byte global load/store: %s\n", bad ? "BROKEN" : "ok");
    return 0;
}
