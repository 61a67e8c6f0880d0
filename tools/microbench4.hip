// microbench4.hip - what the memory side allows for "stream 16 B in + random 8-B gather + stream 16 B out".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: copy only; 1: + random gather; 2: + gather with the table index taken from the data (dependent, like the integrator)
// gather through an explicit cache policy (no prefetch; vmcnt(0) before use)
template <int POL>
__global__ __launch_bounds__(256) void kpol(const v4f *in, v4f *out, const v2f *table, unsigned mask, unsigned n)
{
    unsigned stride = gridDim.x * 256u;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += stride) {
        v4f st = __builtin_nontemporal_load(in + idx);
        unsigned h = __float_as_uint(st.x);
        h = (h ^ (h >> 15)) * 2246822519u;
        const v2f *p = table + ((h >> 7) & mask);
        v2f f;
        if (POL == 0) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(p) : "memory");
        if (POL == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(p) : "memory");
        if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(p) : "memory");
        if (POL == 3) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(p) : "memory");
        if (POL == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(p) : "memory");
        if (POL == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(p) : "memory");
        st.z += f.x; st.w += f.y;
        st.x += st.z; st.y += st.w;
        __builtin_nontemporal_store(st, out + idx);
    }
}

template <int MODE, bool PREFETCH>
__global__ __launch_bounds__(256) void k(const v4f *in, v4f *out, const v2f *table, unsigned mask, unsigned n)
{
    unsigned stride = gridDim.x * 256u;
    unsigned idx = blockIdx.x * 256u + threadIdx.x;
    v4f nxt = {0, 0, 0, 0};
    if (PREFETCH && idx < n) nxt = __builtin_nontemporal_load(in + idx);
    for (; idx < n; idx += stride) {
        v4f st;
        if (PREFETCH) { st = nxt; if (idx + stride < n) nxt = __builtin_nontemporal_load(in + idx + stride); }
        else st = __builtin_nontemporal_load(in + idx);
        if (MODE >= 1) {
            unsigned h = MODE == 2 ? __float_as_uint(st.x) : idx * 2654435761u;
            h = (h ^ (h >> 15)) * 2246822519u;
            v2f f = table[(h >> 7) & mask];
            st.z += f.x; st.w += f.y;
        }
        st.x += st.z; st.y += st.w;
        __builtin_nontemporal_store(st, out + idx);
    }
}

int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned n = 1u << 24;
    v4f *a, *b; CK(hipMalloc(&a, (size_t)n * 16)); CK(hipMalloc(&b, (size_t)n * 16));
    {   // random-ish float data
        float *h = (float *)malloc((size_t)n * 16);
        unsigned s = 12345; for (size_t i = 0; i < (size_t)n * 4; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 16777216.0f); }
        CK(hipMemcpy(a, h, (size_t)n * 16, hipMemcpyHostToDevice)); free(h);
    }
    for (size_t mib = 1; mib <= 16; mib *= 4) {
        size_t bytes = mib << 20;
        v2f *table; CK(hipMalloc(&table, bytes)); CK(hipMemset(table, 0, bytes));
        unsigned mask = (unsigned)(bytes / 8 - 1);
        for (int mode = 0; mode < 3; ++mode)
            for (int pf = 0; pf < 2; ++pf) {
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
#define L(M, P) hipLaunchKernelGGL((k<M, P>), dim3(2048), dim3(256), 0, 0, a, b, table, mask, n)
                    if (mode == 0) { if (pf) L(0, true); else L(0, false); }
                    if (mode == 1) { if (pf) L(1, true); else L(1, false); }
                    if (mode == 2) { if (pf) L(2, true); else L(2, false); }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                }
                printf("table %2zu MiB  mode %d (%s)  prefetch %d: %.4f ms\n", mib, mode,
                       mode == 0 ? "copy" : mode == 1 ? "copy+gather(idx hash)" : "copy+gather(data-dependent)", pf, ms);
            }
        const char *pn[] = {"default", "sc1", "sc0 sc1", "nt", "sc0", "sc0 sc1 nt"};
        for (int pol = 0; pol < 6; ++pol) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
#define P(K) case K: hipLaunchKernelGGL((kpol<K>), dim3(2048), dim3(256), 0, 0, a, b, table, mask, n); break;
                switch (pol) { P(0) P(1) P(2) P(3) P(4) P(5) }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("table %2zu MiB  copy+gather policy %-10s: %.4f ms\n", mib, pn[pol], ms);
        }
        CK(hipFree(table));
    }
    return 0;
}
