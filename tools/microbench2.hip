// microbench2.hip - VALU issue rate vs occupancy and encoding on gfx950 (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (KIND == 1) asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                                             "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            else if (KIND == 3) asm volatile("v_mul_f32 %0, 0x3f800001, %0\n v_add_f32 %1, 0x3a83126f, %1\n v_mul_f32 %2, 0x3f800001, %2\n v_add_f32 %3, 0x3a83126f, %3\n"
                                             "v_mul_f32 %4, 0x3f800001, %4\n v_add_f32 %5, 0x3a83126f, %5\n v_mul_f32 %6, 0x3f800001, %6\n v_add_f32 %7, 0x3a83126f, %7\n"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            else if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %9\n"
                                             "v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %9\n"
                                             : "+v"(x0) : "v"(a), "v"(b), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));   // one dependent chain
            else if (KIND == 5) asm volatile("v_mul_f32 %0, s4, %0\n v_add_f32 %1, s5, %1\n v_mul_f32 %2, s4, %2\n v_add_f32 %3, s5, %3\n"
                                             "v_mul_f32 %4, s4, %4\n v_add_f32 %5, s5, %5\n v_mul_f32 %6, s4, %6\n v_add_f32 %7, s5, %7\n"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));   // SGPR operand
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *out; CK(hipMalloc(&out, 8192 * 256 * sizeof(float)));
    const char *names[] = {"v_fma_f32 (VOP3)", "v_fmac_f32 (VOP2)", "mul/add VOP2", "mul/add literal", "dependent mul chain", "mul/add SGPR src"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        int grid = 256 * wps, iters = 10000;
        for (int kind = 0; kind < 6; ++kind) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (kind) {
                case 0: hipLaunchKernelGGL(valu_kernel<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 1: hipLaunchKernelGGL(valu_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 2: hipLaunchKernelGGL(valu_kernel<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 3: hipLaunchKernelGGL(valu_kernel<3>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 4: hipLaunchKernelGGL(valu_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 5: hipLaunchKernelGGL(valu_kernel<5>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double insts_per_simd = (double)iters * 64 * wps;          // wave-instructions issued per SIMD
            printf("waves/SIMD %d  %-22s %.3f ms  %.2f ns per wave-inst per SIMD (= %.2f cyc at 2.4 GHz)\n", wps, names[kind], ms,
                   ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
        }
    }
    return 0;
}
