// unaligned_lds.hip -- does gfx950 LDS take unaligned 4-byte stores / loads? (probe, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void k(const unsigned *src, unsigned char *dst, int off, int stride) {
    __shared__ __attribute__((aligned(16))) unsigned char s[4096];
    const int i = threadIdx.x;
    for (int j = i; j < 1024; j += 64) reinterpret_cast<unsigned *>(s)[j] = 0;
    __syncthreads();
    // lane i stores a dword at byte offset off + stride * i (overlapping tails: later lanes win where they overlap? no
    // order between lanes of one instruction -- use stride >= 4 for a defined result)
    *reinterpret_cast<unsigned *>(s + off + stride * i) = src[i];
    __syncthreads();
    const unsigned v = *reinterpret_cast<const unsigned *>(s + off + stride * i);  // unaligned load back
    reinterpret_cast<unsigned *>(dst)[i] = v;
    for (int j = i; j < 1024; j += 64) reinterpret_cast<unsigned *>(dst + 256)[j] = reinterpret_cast<unsigned *>(s)[j];
}
int main() {
    unsigned h[64], *s;
    unsigned char o[256 + 4096], *d;
    for (int i = 0; i < 64; i++) h[i] = 0x01020304u * (unsigned)(i + 1) + 0x10203040u;
    hipMalloc(&s, 256); hipMalloc(&d, sizeof o);
    hipMemcpy(s, h, 256, hipMemcpyHostToDevice);
    int bad = 0;
    for (int off = 0; off < 4; off++) for (int stride = 4; stride <= 7; stride++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, off, stride);
        if (hipDeviceSynchronize() != hipSuccess) { printf("fault at off=%d stride=%d\n", off, stride); return 1; }
        hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; i++) {
            unsigned back, mem;
            memcpy(&back, o + 4 * i, 4);
            memcpy(&mem, o + 256 + off + stride * i, 4);
            if (back != h[i] || mem != h[i]) { bad++; if (bad < 8) printf("mismatch off=%d stride=%d lane=%d %08x %08x want %08x\n", off, stride, i, back, mem, h[i]); }
        }
    }
    printf("unaligned 4-byte LDS store/load: %s\n", bad ? "BROKEN" : "ok");
    return 0;
}
