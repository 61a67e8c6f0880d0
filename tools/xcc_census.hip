// xcc_census.hip - checks HW_REG_XCC_ID reads and the blockIdx -> XCD dealing on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void census(unsigned *xcc_of_block)
{
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}
int main()
{
    const int n = 2048;
    unsigned *d, h[n];
    CK(hipMalloc(&d, n * sizeof(unsigned)));
    hipLaunchKernelGGL(census, dim3(n), dim3(256), 0, 0, d);
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int count[16] = {0}, match = 0;
    for (int b = 0; b < n; ++b) { count[h[b] & 15]++; match += (h[b] == (unsigned)(b % 8)); }
    printf("blocks per XCC id:"); for (int i = 0; i < 16; ++i) printf(" %d", count[i]); printf("\n");
    printf("blocks with xcc == blockIdx %% 8: %d of %d; first 16:", match, n);
    for (int b = 0; b < 16; ++b) printf(" %u", h[b]); printf("\n");
    return 0;
}
