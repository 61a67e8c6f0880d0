// microbench.hip - gfx950 probes that size the integrator's ceilings (not product code):
//   valu    issue rate of v_fma_f32 / v_pk_fma_f32 / v_mul+v_add / v_floor / v_cndmask chains
//   gather  random 8-B and 16-B gathers from tables of 1..64 MiB (L2 / Infinity Cache)
//   copy    float4 streaming copy, plain vs non-temporal
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/bin/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v2f pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {   // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) {   // 4 v_pk_fma_f32 (8 lanes-worth of fma)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (KIND == 2) {   // 4 mul + 4 add
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 3) {   // 8 v_floor_f32
                asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n"
                             "v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 4) {   // 4 v_pk_mul_f32 (8 lanes-worth)
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (KIND == 5) {   // 8 v_cvt_i32_f32
                asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n"
                             "v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T *table, unsigned mask, float *out, int per_thread, unsigned seed)
{
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
    float acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        s = s * 1664525u + 1013904223u;
        unsigned idx = (s >> 4) & mask;
        T v = table[idx];
        acc += v.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// cache-policy variants of the 8-B gather (4 loads in flight per lane, then one wait)
template <int POL>
__global__ __launch_bounds__(256) void gather_policy_kernel(const v2f *table, unsigned mask, float *out, int per_thread, unsigned seed)
{
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
    float acc = 0;
    for (int i = 0; i < per_thread; i += 4) {
        const v2f *p[4];
        for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; p[j] = table + ((s >> 4) & mask); }
        v2f v0, v1, v2, v3;
#define LD4(suffix) asm volatile("global_load_dwordx2 %0, %4, off " suffix "\n global_load_dwordx2 %1, %5, off " suffix "\n" \
                                 "global_load_dwordx2 %2, %6, off " suffix "\n global_load_dwordx2 %3, %7, off " suffix "\n s_waitcnt vmcnt(0)" \
                                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory")
        if (POL == 0) LD4("");
        else if (POL == 1) LD4("nt");
        else if (POL == 2) LD4("sc1");
        else if (POL == 3) LD4("sc0 sc1");
        else if (POL == 4) LD4("sc0");
        else LD4("sc1 nt");
#undef LD4
        acc += v0.x + v1.x + v2.x + v3.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const v4f *in, v4f *out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
        else out[i] = in[i];
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main()
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *out; CK(hipMalloc(&out, 2048 * 256 * 8 * sizeof(float)));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);

    // ---- VALU ----
    const char *names[] = {"v_fma_f32 x8", "v_pk_fma_f32 x4", "mul/add x8", "v_floor_f32 x8", "v_pk_mul_f32 x4", "v_cvt_i32_f32 x8"};
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        int grid = 256 * waves_per_simd;     // 256-thread blocks: 4 waves per block = 1 per SIMD
        int iters = 20000;
        for (int kind = 0; kind < 6; ++kind) {
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
                CK(hipEventRecord(e1));
                float ms = time_ms(e0, e1);
                if (rep == 1) {
                    double groups = (double)iters * 8;                    // asm groups per wave
                    double insts = groups * (kind == 1 || kind == 4 ? 4 : 8);
                    double waves = (double)grid * 4;
                    double lane_ops = insts * waves * 64 * (kind == 1 || kind == 4 ? 2 : 1);
                    printf("valu %-18s waves/SIMD %d: %.3f ms  %.2f T lane-ops/s  (%.2f T wave-inst*64/s)\n", names[kind], waves_per_simd, ms,
                           lane_ops / ms / 1e9, insts * waves * 64 / ms / 1e9);
                }
            }
        }
    }

    // ---- gather ----
    for (int elem = 8; elem <= 16; elem += 8) {
        for (size_t mib = 1; mib <= 64; mib *= 2) {
            size_t bytes = mib << 20;
            void *table; CK(hipMalloc(&table, bytes)); CK(hipMemset(table, 0, bytes));
            unsigned mask = (unsigned)(bytes / elem - 1);
            int grid = 256 * 8, per_thread = 64;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (elem == 8) hipLaunchKernelGGL(gather_kernel<float2>, dim3(grid), dim3(256), 0, 0, (const float2 *)table, mask, out, per_thread, 17u + rep);
                else hipLaunchKernelGGL(gather_kernel<float4>, dim3(grid), dim3(256), 0, 0, (const float4 *)table, mask, out, per_thread, 17u + rep);
                CK(hipEventRecord(e1));
                float ms = time_ms(e0, e1);
                if (rep == 2) printf("gather %2d-B from %3zu MiB: %.3f ms  %.2f G gathers/s\n", elem, mib, ms, (double)grid * 256 * per_thread / ms / 1e6);
            }
            CK(hipFree(table));
        }
    }

    // ---- gather cache policies (8-B, 16 MiB table) ----
    {
        size_t bytes = (size_t)16 << 20;
        void *table; CK(hipMalloc(&table, bytes)); CK(hipMemset(table, 0, bytes));
        unsigned mask = (unsigned)(bytes / 8 - 1);
        const char *pn[] = {"default", "nt", "sc1", "sc0 sc1", "sc0", "sc1 nt"};
        int grid = 256 * 8, per_thread = 64;
        for (int pol = 0; pol < 6; ++pol)
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                switch (pol) {
                case 0: hipLaunchKernelGGL(gather_policy_kernel<0>, dim3(grid), dim3(256), 0, 0, (const v2f *)table, mask, out, per_thread, 5u + rep); break;
                case 1: hipLaunchKernelGGL(gather_policy_kernel<1>, dim3(grid), dim3(256), 0, 0, (const v2f *)table, mask, out, per_thread, 5u + rep); break;
                case 2: hipLaunchKernelGGL(gather_policy_kernel<2>, dim3(grid), dim3(256), 0, 0, (const v2f *)table, mask, out, per_thread, 5u + rep); break;
                case 3: hipLaunchKernelGGL(gather_policy_kernel<3>, dim3(grid), dim3(256), 0, 0, (const v2f *)table, mask, out, per_thread, 5u + rep); break;
                case 4: hipLaunchKernelGGL(gather_policy_kernel<4>, dim3(grid), dim3(256), 0, 0, (const v2f *)table, mask, out, per_thread, 5u + rep); break;
                case 5: hipLaunchKernelGGL(gather_policy_kernel<5>, dim3(grid), dim3(256), 0, 0, (const v2f *)table, mask, out, per_thread, 5u + rep); break;
                }
                CK(hipEventRecord(e1));
                float ms = time_ms(e0, e1);
                if (rep == 2) printf("gather 8-B 16 MiB policy %-8s: %.3f ms  %.2f G gathers/s\n", pn[pol], ms, (double)grid * 256 * per_thread / ms / 1e6);
            }
        CK(hipFree(table));
    }

    // ---- copy ----
    size_t n = (size_t)16 << 20;    // 16M float4 = 256 MiB each way
    v4f *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
    for (int nt = 0; nt < 2; ++nt)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (nt) hipLaunchKernelGGL(copy_kernel<true>, dim3(2048), dim3(256), 0, 0, a, b, n);
            else hipLaunchKernelGGL(copy_kernel<false>, dim3(2048), dim3(256), 0, 0, a, b, n);
            CK(hipEventRecord(e1));
            float ms = time_ms(e0, e1);
            if (rep == 2) printf("copy float4 %s 256 MiB: %.3f ms  %.2f TB/s (read+write)\n", nt ? "nontemporal" : "plain      ", ms, 2.0 * n * 16 / ms / 1e9);
        }
    return 0;
}
