// microbench5.hip - streaming copy (16 B in, 16 B out, nt) when the grid is split into G groups, each
// grid-striding over its own contiguous 1/G of the range (G = 1: plain grid-stride; G = 2048: one
// contiguous chunk per workgroup).  How many concurrent stream windows does HBM like?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_groups(const v4f *in, v4f *out, unsigned n, unsigned groups, int interleaved)
{
    // interleaved: group = blockIdx % groups (XCD-style); else group = blockIdx / (gridDim/groups)
    unsigned per_group_blocks = gridDim.x / groups;
    unsigned g = interleaved ? blockIdx.x % groups : blockIdx.x / per_group_blocks;
    unsigned rank = interleaved ? blockIdx.x / groups : blockIdx.x % per_group_blocks;
    unsigned per = n / groups;
    unsigned lo = g * per, hi = lo + per;
    unsigned stride = per_group_blocks * 256u;
    unsigned idx = lo + rank * 256u + threadIdx.x;
    v4f nxt = {0, 0, 0, 0};
    if (idx < hi) nxt = __builtin_nontemporal_load(in + idx);
    for (; idx < hi; idx += stride) {
        v4f st = nxt;
        if (idx + stride < hi) nxt = __builtin_nontemporal_load(in + idx + stride);
        st.x += st.z; st.y += st.w;
        __builtin_nontemporal_store(st, out + idx);
    }
}

int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned n = 1u << 24;
    v4f *a, *b; CK(hipMalloc(&a, (size_t)n * 16)); CK(hipMalloc(&b, (size_t)n * 16)); CK(hipMemset(a, 0, (size_t)n * 16));
    for (int inter = 0; inter < 2; ++inter)
        for (unsigned groups = 1; groups <= 2048; groups *= 4) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(copy_groups, dim3(2048), dim3(256), 0, 0, a, b, n, groups, inter);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("groups %4u  %s: %.4f ms  (%.2f TB/s)\n", groups, inter ? "group = block %% G" : "group = block / (2048/G)", ms, 2.0 * n * 16 / ms / 1e9);
        }
    return 0;
}
