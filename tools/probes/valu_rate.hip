// valu_rate.hip -- issue cost of the VALU instructions stage 1 is made of (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// 8 independent chains per op so that dependent latency never limits
#define KERNEL(NAME, BODY)                                                                    \
    __global__ void NAME(unsigned *out, unsigned long long *cyc, int iters) {                \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7;                                                                 \
        unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3;                                \
        unsigned b = out[0], c = out[1];                                                      \
        unsigned long long t0 = __builtin_readcyclecounter();                                 \
        for (int i = 0; i < iters; i++) {                                                     \
            REP8(BODY)                                                                        \
        }                                                                                     \
        unsigned long long t1 = __builtin_readcyclecounter();                                 \
        out[2 + threadIdx.x % 7] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(q0 ^ q1 ^ q2 ^ q3);   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                      \
    }

#define OP3(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                             op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define OP4(op) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                             op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define OP2(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" \
                             op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define OPQ(op) asm volatile(op " %0, 3, %0\n" op " %1, 3, %1\n" op " %2, 3, %2\n" op " %3, 3, %3\n" \
                             op " %0, 5, %0\n" op " %1, 5, %1\n" op " %2, 5, %2\n" op " %3, 5, %3\n" \
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
#define OPQA(op) asm volatile(op " %0, %0, 3, %0\n" op " %1, %1, 3, %1\n" op " %2, %2, 3, %2\n" op " %3, %3, 3, %3\n" \
                             op " %0, %0, 1, %0\n" op " %1, %1, 1, %1\n" op " %2, %2, 1, %2\n" op " %3, %3, 1, %3\n" \
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));

KERNEL(k_and, OP3("v_and_b32"))
KERNEL(k_xor, OP3("v_xor_b32"))
KERNEL(k_add, OP3("v_add_u32"))
KERNEL(k_fma, OP4("v_fma_f32"))
KERNEL(k_dot4, OP4("v_dot4_u32_u8"))
KERNEL(k_or3, OP4("v_or3_b32"))
KERNEL(k_and_or, OP4("v_and_or_b32"))
KERNEL(k_bfi, OP4("v_bfi_b32"))
KERNEL(k_perm, OP4("v_perm_b32"))
KERNEL(k_alignbit, OP4("v_alignbit_b32"))
KERNEL(k_lshl_or, OP4("v_lshl_or_b32"))
KERNEL(k_lshl_add, OP4("v_lshl_add_u32"))
KERNEL(k_bcnt, OP3("v_bcnt_u32_b32"))
KERNEL(k_ffbl, OP2("v_ffbl_b32"))
KERNEL(k_mul_lo, OP3("v_mul_lo_u32"))
KERNEL(k_mul_u24, OP3("v_mul_u32_u24"))
KERNEL(k_mad_u24, OP4("v_mad_u32_u24"))
KERNEL(k_lshl64, OPQ("v_lshlrev_b64"))
KERNEL(k_lshladd64, OPQA("v_lshl_add_u64"))
KERNEL(k_sad, OP4("v_sad_u8"))
KERNEL(k_msad, OP4("v_msad_u8"))
KERNEL(k_bfe, OP4("v_bfe_u32"))


#define OP3L(op, lit) asm volatile(op " %0, " lit ", %0\n" op " %1, " lit ", %1\n" op " %2, " lit ", %2\n" op " %3, " lit ", %3\n" \
                             op " %4, " lit ", %4\n" op " %5, " lit ", %5\n" op " %6, " lit ", %6\n" op " %7, " lit ", %7\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define OPCMP(op) asm volatile(op " vcc, %0, %8\n" op " vcc, %1, %8\n" op " vcc, %2, %8\n" op " vcc, %3, %8\n" \
                             op " vcc, %4, %8\n" op " vcc, %5, %8\n" op " vcc, %6, %8\n" op " vcc, %7, %8\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define OPVCC(op) asm volatile(op " %0, %0, %8, vcc\n" op " %1, %1, %8, vcc\n" op " %2, %2, %8, vcc\n" op " %3, %3, %8, vcc\n" \
                             op " %4, %4, %8, vcc\n" op " %5, %5, %8, vcc\n" op " %6, %6, %8, vcc\n" op " %7, %7, %8, vcc\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define OPCO(op) asm volatile(op " %0, vcc, %0, %8\n" op " %1, vcc, %1, %8\n" op " %2, vcc, %2, %8\n" op " %3, vcc, %3, %8\n" \
                             op " %4, vcc, %4, %8\n" op " %5, vcc, %5, %8\n" op " %6, vcc, %6, %8\n" op " %7, vcc, %7, %8\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define OPCOC(op) asm volatile(op " %0, vcc, %0, %8, vcc\n" op " %1, vcc, %1, %8, vcc\n" op " %2, vcc, %2, %8, vcc\n" op " %3, vcc, %3, %8, vcc\n" \
                             op " %4, vcc, %4, %8, vcc\n" op " %5, vcc, %5, %8, vcc\n" op " %6, vcc, %6, %8, vcc\n" op " %7, vcc, %7, %8, vcc\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define OPDPP(op) asm volatile(op " %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                             op " %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                             op " %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                             op " %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define OPSDWA(op) asm volatile(op " %0, %0, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" op " %1, %1, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" \
                             op " %2, %2, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" op " %3, %3, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" \
                             op " %4, %4, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" op " %5, %5, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" \
                             op " %6, %6, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" op " %7, %7, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
KERNEL(k_or, OP3("v_or_b32"))
KERNEL(k_not, OP2("v_not_b32"))
KERNEL(k_mov, OP2("v_mov_b32"))
KERNEL(k_lshl, OP3("v_lshlrev_b32"))
KERNEL(k_lshr, OP3("v_lshrrev_b32"))
KERNEL(k_sub, OP3("v_sub_u32"))
KERNEL(k_min, OP3("v_min_u32"))
KERNEL(k_xnor, OP3("v_xnor_b32"))
KERNEL(k_and_lit, OP3L("v_and_b32", "0x01010101"))
KERNEL(k_and_inl, OP3L("v_and_b32", "15"))
KERNEL(k_lshl_inl, OP3L("v_lshlrev_b32", "3"))
KERNEL(k_cmp_eq, OPCMP("v_cmp_eq_u32"))
KERNEL(k_cndmask, OPVCC("v_cndmask_b32"))
KERNEL(k_add_co, OPCO("v_add_co_u32"))
KERNEL(k_addc_co, OPCOC("v_addc_co_u32"))
KERNEL(k_add_dpp, OPDPP("v_add_u32_dpp"))
KERNEL(k_and_sdwa, OPSDWA("v_and_b32_sdwa"))
KERNEL(k_mul_f32, OP3("v_mul_f32"))
KERNEL(k_add_f32, OP3("v_add_f32"))
KERNEL(k_xad, OP4("v_xad_u32"))
KERNEL(k_add3, OP4("v_add3_u32"))
KERNEL(k_dot4c, OP3("v_dot4c_i32_i8"))
KERNEL(k_dot8, OP4("v_dot8_u32_u4"))
KERNEL(k_mbcnt, OP3("v_mbcnt_lo_u32_b32"))

int main() {
    unsigned *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 4096)); CK(hipMemset(out, 0, 4096)); CK(hipMalloc(&cyc, 8 * 4096));
    const int iters = 200;
    unsigned long long h[4096];
#define RUN(K, waves_per_simd) { \
        const int threads = 64 * 4 * waves_per_simd; \
        hipLaunchKernelGGL(K, dim3(256), dim3(threads), 0, 0, out, cyc, iters); \
        CK(hipDeviceSynchronize()); \
        hipLaunchKernelGGL(K, dim3(256), dim3(threads), 0, 0, out, cyc, iters); \
        CK(hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost)); \
        double s = 0; for (int i = 0; i < 256; i++) s += (double)h[i]; s /= 256; \
        printf("%-14s waves/SIMD %d: %7.2f cycles per wave-instruction, %5.2f per instr per SIMD\n", #K, waves_per_simd, \
               s / (iters * 64.0), s / (iters * 64.0) / waves_per_simd); }
#define THR(K) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); const int it2 = 2000; \
        hipLaunchKernelGGL(K, dim3(256 * 8), dim3(256), 0, 0, out, cyc, 10); CK(hipDeviceSynchronize()); \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(K, dim3(256 * 8), dim3(256), 0, 0, out, cyc, it2); CK(hipEventRecord(e1)); \
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        const double winst = 256.0 * 8 * 4 * it2 * 64; /* wave-instructions */ \
        printf("%-14s full chip: %.3f ms, %.2f ns per wave-instr per SIMD = %.2f cycles @2.4GHz\n", #K, ms, \
               ms * 1e6 / (winst / 1024.0), ms * 1e6 / (winst / 1024.0) * 2.4); }
#define THRW(K, W) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); const int it2 = 2000; \
        hipLaunchKernelGGL(K, dim3(256), dim3(256 * W), 0, 0, out, cyc, 10); CK(hipDeviceSynchronize()); \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(K, dim3(256), dim3(256 * W), 0, 0, out, cyc, it2); CK(hipEventRecord(e1)); \
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        CK(hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost)); double sc = 0; for (int i = 0; i < 256; i++) sc += (double)h[i]; sc /= 256; \
        const double winst = 256.0 * W * 4 * it2 * 64; \
        printf("%-10s %d waves/SIMD: %.3f ms  %.2f ns/winst/SIMD = %.2f cyc@2.4G ; memtime ticks/ns = %.3f\n", #K, W, ms, \
               ms * 1e6 / (winst / 1024.0), ms * 1e6 / (winst / 1024.0) * 2.4, sc / (ms * 1e6)); }
#define ALL(K) THRW(K, 1) THRW(K, 2) THRW(K, 3) THRW(K, 4)
    ALL(k_and) ALL(k_dot4) ALL(k_lshl)
    return 0;
}
