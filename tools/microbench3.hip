// microbench3.hip - cost of scalar-register / VCC operands and of VOP3 forms in VALU code on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define R8(A, B) A "\n" B "\n" A "\n" B "\n" A "\n" B "\n" A "\n" B "\n"
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OPS8(fmt) asm volatile(fmt : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc", "s4", "s5", "s6", "s7", "s8", "s9")
            if (KIND == 0) OPS8("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9");
            if (KIND == 1) OPS8("v_mul_f32 %0, s4, %0\n v_add_f32 %1, s4, %1\n v_mul_f32 %2, s4, %2\n v_add_f32 %3, s4, %3\n v_mul_f32 %4, s4, %4\n v_add_f32 %5, s4, %5\n v_mul_f32 %6, s4, %6\n v_add_f32 %7, s4, %7");
            if (KIND == 2) OPS8("v_mul_f32 %0, 2.0, %0\n v_add_f32 %1, 0.5, %1\n v_mul_f32 %2, 2.0, %2\n v_add_f32 %3, 0.5, %3\n v_mul_f32 %4, 2.0, %4\n v_add_f32 %5, 0.5, %5\n v_mul_f32 %6, 2.0, %6\n v_add_f32 %7, 0.5, %7");
            if (KIND == 3) OPS8("v_mul_f32_e64 %0, %0, %8\n v_add_f32_e64 %1, %1, %9\n v_mul_f32_e64 %2, %2, %8\n v_add_f32_e64 %3, %3, %9\n v_mul_f32_e64 %4, %4, %8\n v_add_f32_e64 %5, %5, %9\n v_mul_f32_e64 %6, %6, %8\n v_add_f32_e64 %7, %7, %9");
            if (KIND == 4) OPS8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %9, vcc");
            if (KIND == 5) OPS8("v_cndmask_b32_e64 %0, %0, %8, s[4:5]\n v_cndmask_b32_e64 %1, %1, %9, s[6:7]\n v_cndmask_b32_e64 %2, %2, %8, s[4:5]\n v_cndmask_b32_e64 %3, %3, %9, s[6:7]\n v_cndmask_b32_e64 %4, %4, %8, s[4:5]\n v_cndmask_b32_e64 %5, %5, %9, s[6:7]\n v_cndmask_b32_e64 %6, %6, %8, s[4:5]\n v_cndmask_b32_e64 %7, %7, %9, s[6:7]");
            if (KIND == 6) OPS8("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %9\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %9\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %9\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %9");
            if (KIND == 7) OPS8("v_cmp_lt_f32_e64 s[4:5], %0, %8\n v_cmp_lt_f32_e64 s[6:7], %1, %9\n v_cmp_lt_f32_e64 s[8:9], %2, %8\n v_cmp_lt_f32_e64 s[4:5], %3, %9\n v_cmp_lt_f32_e64 s[6:7], %4, %8\n v_cmp_lt_f32_e64 s[8:9], %5, %9\n v_cmp_lt_f32_e64 s[4:5], %6, %8\n v_cmp_lt_f32_e64 s[6:7], %7, %9");
            if (KIND == 8) OPS8("v_fma_f32 %0, %0, s4, 1.0\n v_fma_f32 %1, %1, s4, 1.0\n v_fma_f32 %2, %2, s4, 1.0\n v_fma_f32 %3, %3, s4, 1.0\n v_fma_f32 %4, %4, s4, 1.0\n v_fma_f32 %5, %5, s4, 1.0\n v_fma_f32 %6, %6, s4, 1.0\n v_fma_f32 %7, %7, s4, 1.0");
            if (KIND == 9) OPS8("v_max_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_sub_f32 %3, %3, %9\n v_max_f32 %4, %4, %8\n v_sub_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_sub_f32 %7, %7, %9");
            if (KIND == 10) OPS8("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9");
            if (KIND == 11) OPS8("v_mul_f32 %0, 0x42080000, %0\n v_fmac_f32 %1, %8, %9\n v_mul_f32 %2, 0x42080000, %2\n v_fmac_f32 %3, %8, %9\n v_mul_f32 %4, 0x42080000, %4\n v_fmac_f32 %5, %8, %9\n v_mul_f32 %6, 0x42080000, %6\n v_fmac_f32 %7, %8, %9");
            if (KIND == 12) OPS8("v_fmaak_f32 %0, %0, %8, 0x3f800000\n v_fmamk_f32 %1, %1, 0x42080000, %9\n v_fmaak_f32 %2, %2, %8, 0x3f800000\n v_fmamk_f32 %3, %3, 0x42080000, %9\n v_fmaak_f32 %4, %4, %8, 0x3f800000\n v_fmamk_f32 %5, %5, 0x42080000, %9\n v_fmaak_f32 %6, %6, %8, 0x3f800000\n v_fmamk_f32 %7, %7, 0x42080000, %9");
            if (KIND == 13) OPS8("v_floor_f32 %0, %0\n v_mul_f32 %1, %1, %9\n v_floor_f32 %2, %2\n v_mul_f32 %3, %3, %9\n v_floor_f32 %4, %4\n v_mul_f32 %5, %5, %9\n v_floor_f32 %6, %6\n v_mul_f32 %7, %7, %9");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *out; CK(hipMalloc(&out, 8192 * 256 * sizeof(float)));
    const char *names[] = {"mul/add vgpr (VOP2)", "mul/add same SGPR src0", "mul/add inline const", "mul/add VOP3 (e64) vgpr", "cndmask vcc (VOP2)",
                           "cndmask e64 sgpr-pair", "v_cmp -> vcc", "v_cmp_e64 -> sgpr pair", "fma with SGPR operand", "max/sub/min (VOP2)", "v_med3 (VOP3)",
                           "mul literal + fmac", "fmaak/fmamk literal", "floor + mul alternating"};
    for (int wps = 4; wps <= 8; wps *= 2) {
        int grid = 256 * wps, iters = 8000;
        for (int kind = 0; kind < 14; ++kind) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
#define L(K) case K: hipLaunchKernelGGL(k<K>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                switch (kind) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double insts_per_simd = (double)iters * 64 * wps;
            printf("waves/SIMD %d  %-26s %.3f ms  %.3f ns per wave-inst per SIMD\n", wps, names[kind], ms, ms * 1e6 / insts_per_simd);
        }
    }
    return 0;
}
