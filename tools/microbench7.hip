// microbench7: throughput of device-scope atomic adds (returning / not) on hipMalloc memory: how many per second the chip
// completes when they are spread over a region of R bytes, `per` of them in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PER, bool RET>
__global__ __launch_bounds__(256) void atomics(unsigned *a, unsigned mask, unsigned *out, unsigned stride)
{
    unsigned i = blockIdx.x * 256u + threadIdx.x, h = i * 2654435761u, sum = 0;
    unsigned r[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        h = h * 1664525u + 1013904223u;
        unsigned at = ((h >> 8) & mask) * stride;
        if (RET) r[k] = atomicAdd(a + at, 1u);
        else { atomicAdd(a + at, 1u); r[k] = 0; }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) sum += r[k];
    if (sum == 0xffffffffu) out[i] = sum;
}

template <int PER, bool RET>
static void run(unsigned *a, unsigned *out, unsigned words, unsigned stride, unsigned threads, int active_lanes_note)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = threads / 256u;
    atomics<PER, RET><<<blocks, 256>>>(a, words - 1u, out, stride);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 5; ++it) atomics<PER, RET><<<blocks, 256>>>(a, words - 1u, out, stride);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = 5.0 * threads * PER;
    printf("  %s  per-thread %d  region %8u words x stride %3u  threads %8u : %7.2f G atomics/s  (%.1f us per launch)\n", RET ? "returning" : "no return", PER, words, stride, threads, n / ms / 1e6, ms / 5 * 1e3);
}

int main()
{
    unsigned *a, *out;
    CK(hipMalloc(&a, 1u << 30)); CK(hipMemset(a, 0, 1u << 30));
    CK(hipMalloc(&out, 1u << 26));
    for (unsigned threads : {1u << 16, 1u << 20, 1u << 22}) {
        for (unsigned words : {1u << 17, 1u << 24}) {
            run<1, true>(a, out, words, 1, threads, 64);
            run<4, true>(a, out, words, 1, threads, 64);
            run<4, false>(a, out, words, 1, threads, 64);
        }
        run<4, true>(a, out, 1u << 17, 16, threads, 64);      // one counter per 64 B
        run<4, true>(a, out, 1u << 13, 32, threads, 64);      // 8192 counters, one per 128-byte line
        run<4, true>(a, out, 1u << 13, 1, threads, 64);       // 8192 counters side by side (32 KB)
        run<4, true>(a, out, 1u << 6, 64, threads, 64);       // 64 counters 256 B apart
    }
    return 0;
}
