// microbench6.hip - state layout probe: RGBA32F texels (one 16-B access per lane) vs four f32 planes
// (struct-of-arrays) for the integrator's stream (read 4 floats, write 4 floats per particle, nt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void aos(const v4f *in, v4f *out, unsigned n)
{
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        v4f s = __builtin_nontemporal_load(in + i);
        s.x += s.z; s.y += s.w;
        __builtin_nontemporal_store(s, out + i);
    }
}
// one particle per lane, four 4-B planes
__global__ __launch_bounds__(256) void soa1(const float *in, float *out, unsigned n)
{
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        float x = __builtin_nontemporal_load(in + i), y = __builtin_nontemporal_load(in + n + i);
        float z = __builtin_nontemporal_load(in + 2u * n + i), w = __builtin_nontemporal_load(in + 3u * n + i);
        __builtin_nontemporal_store(x + z, out + i); __builtin_nontemporal_store(y + w, out + n + i);
        __builtin_nontemporal_store(z, out + 2u * n + i); __builtin_nontemporal_store(w, out + 3u * n + i);
    }
}
// four particles per lane, 16-B accesses to each plane
__global__ __launch_bounds__(256) void soa4(const v4f *in, v4f *out, unsigned n4)
{
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        v4f x = __builtin_nontemporal_load(in + i), y = __builtin_nontemporal_load(in + n4 + i);
        v4f z = __builtin_nontemporal_load(in + 2u * n4 + i), w = __builtin_nontemporal_load(in + 3u * n4 + i);
        __builtin_nontemporal_store(x + z, out + i); __builtin_nontemporal_store(y + w, out + n4 + i);
        __builtin_nontemporal_store(z, out + 2u * n4 + i); __builtin_nontemporal_store(w, out + 3u * n4 + i);
    }
}
int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned n = 1u << 24;
    void *a, *b; CK(hipMalloc(&a, (size_t)n * 16)); CK(hipMalloc(&b, (size_t)n * 16)); CK(hipMemset(a, 0, (size_t)n * 16));
    const char *names[] = {"RGBA32F texel per lane (1 x dwordx4)", "4 planes, 1 particle per lane (4 x dword)", "4 planes, 4 particles per lane (4 x dwordx4)"};
    for (int v = 0; v < 3; ++v) {
        float ms = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(aos, dim3(2048), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, n);
            if (v == 1) hipLaunchKernelGGL(soa1, dim3(2048), dim3(256), 0, 0, (const float *)a, (float *)b, n);
            if (v == 2) hipLaunchKernelGGL(soa4, dim3(2048), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, n / 4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%-46s %.4f ms  %.2f TB/s\n", names[v], ms, 2.0 * n * 16 / ms / 1e9);
    }
    return 0;
}
