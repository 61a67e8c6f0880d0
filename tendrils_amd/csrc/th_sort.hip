// th_sort.hip - stable LSD radix sort of (key, 32-bit value) pairs for gfx950, used by the flow deposit to bring the
// fragments of Tendrils.draw() (src/index.js:295-303) into (flow texel, stream order) order.
//
// GL blends the fragments of one texel in primitive order; the fragment array is produced in that order, so a STABLE
// sort by texel is all that is needed (th_deposit.hip).  Digits of kRadixBits bits, least significant first; per pass
//   radix_hist_kernel     the digit histogram of every 4096-element block of the pass's input (counts[digit][block]:
//                         digit-major, so one flat exclusive scan gives every block's first output slot per digit)
//   radix_scatter_kernel  block-local stable ranking - every wave owns 1024 consecutive elements and ranks them 64
//                         at a time: same-digit lanes of a round are matched with one ballot per digit bit, earlier
//                         rounds through the wave's private LDS counters, earlier waves through the per-wave
//                         histograms - then the block is laid out by digit in LDS and written from there, so that a
//                         digit's run leaves the block as one contiguous store.
// No atomics decide an output position: the result is the same on every run.
#include "th_kernels.hpp"
#include "th_math.hpp"

namespace th {
namespace {

constexpr uint32_t kRadixBlock = 4096;       // elements per workgroup: 4 waves x 16 rounds x 64 lanes
constexpr int kRadixMaxPasses = 8;

TH_D uint32_t digit_of(uint32_t k, int shift, uint32_t mask) { return (k >> shift) & mask; }
TH_D uint32_t digit_of(unsigned long long k, int shift, uint32_t mask) { return (uint32_t)(k >> shift) & mask; }

struct RadixPlan {
    int passes;
    int shift[kRadixMaxPasses];
    uint32_t bits[kRadixMaxPasses];
};

template <typename K>
__global__ __launch_bounds__(256) void radix_hist_kernel(const K *keys, uint32_t n, int shift, uint32_t bits, uint32_t nblocks, uint32_t *counts)
{
    __shared__ uint32_t h[1u << kRadixBits];
    const uint32_t mask = (1u << bits) - 1u;
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRadixBlock;
    // all 16 keys of a lane first (unconditional, clamped: a load under a branch is awaited before the next is issued)
    K key[kRadixBlock / 256u];
#pragma unroll
    for (uint32_t k = 0; k < kRadixBlock / 256u; ++k) {
        const uint32_t i = base + k * 256u + threadIdx.x;
        key[k] = keys[i < n ? i : n - 1u];
    }
#pragma unroll
    for (uint32_t k = 0; k < kRadixBlock / 256u; ++k) {
        const uint32_t i = base + k * 256u + threadIdx.x;
        if (i < n) atomicAdd(&h[digit_of(key[k], shift, mask)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < (1u << bits)) counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <typename K>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const K *keys_in, const uint32_t *vals_in, K *keys_out, uint32_t *vals_out,
                                                            uint32_t n, int shift, uint32_t bits, uint32_t nblocks, const uint32_t *base)
{
    constexpr uint32_t D = 1u << kRadixBits;
    __shared__ uint32_t wave_hist[4][D];       // per wave: elements of each digit; then: first slot of (wave, digit) in the block
    __shared__ uint32_t digit_start[D];        // first slot of each digit inside the block
    __shared__ uint32_t block_base[D];         // this block's first output slot of each digit (one global read per digit, up front:
                                               // read per element in the output loop it was a dependent round trip per iteration)
    __shared__ K stage_k[kRadixBlock];
    __shared__ uint32_t stage_v[kRadixBlock];
    const uint32_t mask = (1u << bits) - 1u, digits = 1u << bits;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t block_first = blockIdx.x * kRadixBlock;
    const uint32_t in_block = n - block_first < kRadixBlock ? n - block_first : kRadixBlock;
    for (uint32_t i = threadIdx.x; i < 4u * D; i += 256u) (&wave_hist[0][0])[i] = 0u;
    if (threadIdx.x < digits) block_base[threadIdx.x] = base[(size_t)threadIdx.x * nblocks + blockIdx.x];
    __syncthreads();

    // this wave's 1024 consecutive elements, 16 rounds of 64
    K key[16];
    uint32_t val[16];
    const uint32_t wave_first = wave * 1024u;
    // (all 32 loads of a lane in flight together: unconditional, clamped - 16 round trips in a row were the pass's time)
    const uint32_t *vals = vals_in ? vals_in : reinterpret_cast<const uint32_t *>(keys_in);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t e = wave_first + (uint32_t)r * 64u + lane;
        const uint32_t at = block_first + (e < in_block ? e : 0u);
        key[r] = keys_in[at];
        const uint32_t v = vals[at];
        val[r] = vals_in ? v : at;                  // no values: the elements' own positions
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t e = wave_first + (uint32_t)r * 64u + lane;
        if (e < in_block) atomicAdd(&wave_hist[wave][digit_of(key[r], shift, mask)], 1u);
    }
    __syncthreads();
    // block-local layout: digit d starts at digit_start[d]; inside it the waves follow each other
    if (threadIdx.x < digits) {
        const uint32_t d = threadIdx.x;
        const uint32_t c0 = wave_hist[0][d], c1 = wave_hist[1][d], c2 = wave_hist[2][d], c3 = wave_hist[3][d];
        digit_start[d] = c0 + c1 + c2 + c3;           // (total for now)
        wave_hist[0][d] = 0u; wave_hist[1][d] = c0; wave_hist[2][d] = c0 + c1; wave_hist[3][d] = c0 + c1 + c2;
    }
    __syncthreads();
    if (threadIdx.x < 64u) {          // exclusive scan of the digit totals by one wave (D <= 256: 4 per lane)
        uint32_t t[D / 64u], s = 0;
#pragma unroll
        for (uint32_t k = 0; k < D / 64u; ++k) { const uint32_t d = lane * (D / 64u) + k; t[k] = d < digits ? digit_start[d] : 0u; s += t[k]; }
        uint32_t incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)lane >= o) incl += up; }
        uint32_t run = incl - s;
#pragma unroll
        for (uint32_t k = 0; k < D / 64u; ++k) { const uint32_t d = lane * (D / 64u) + k; if (d < digits) digit_start[d] = run; run += t[k]; }
    }
    __syncthreads();
    if (threadIdx.x < digits) {
        const uint32_t d = threadIdx.x, s = digit_start[d];
        wave_hist[0][d] += s; wave_hist[1][d] += s; wave_hist[2][d] += s; wave_hist[3][d] += s;
    }
    __syncthreads();

    // stable ranks: round by round, the wave's own counters advance in program order
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t e = wave_first + (uint32_t)r * 64u + lane;
        const bool valid = e < in_block;
        const uint32_t d = digit_of(key[r], shift, mask);
        // lanes of this round with the same digit: one ballot per digit bit
        unsigned long long same = __ballot(valid);
        for (uint32_t b = 0; b < bits; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        if (valid) {
            const uint32_t earlier = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            const uint32_t pos = wave_hist[wave][d] + earlier;
            stage_k[pos] = key[r];
            stage_v[pos] = val[r];
        }
        // the last lane of every group moves the wave's counter past the group (after every lane has read it)
        __builtin_amdgcn_wave_barrier();
        if (valid && (same >> lane) == 1ull) wave_hist[wave][d] += (uint32_t)__builtin_popcountll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // out: element i of the block layout belongs to digit d -> slot base[d][block] + (i - digit_start[d])
    for (uint32_t i = threadIdx.x; i < in_block; i += 256u) {
        const K k = stage_k[i];
        const uint32_t d = digit_of(k, shift, mask);
        const uint32_t at = block_base[d] + (i - digit_start[d]);
        keys_out[at] = k;
        vals_out[at] = stage_v[i];
    }
}

// ---- exclusive scan of a flat u32 array (1024 elements per block, three small kernels) ----------------------------
constexpr uint32_t kScanBlock = 1024;

__global__ __launch_bounds__(256) void sort_scan_local_kernel(uint32_t *data, uint32_t *block_sums, uint32_t n)
{
    __shared__ uint32_t sh[256];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? data[base + k] : 0u; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {
        uint32_t add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) data[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = sh[255];
}

// exclusive scan of the block sums by ONE workgroup of 1024 threads (each walks a contiguous share)
__global__ __launch_bounds__(1024) void sort_scan_sums_kernel(uint32_t *block_sums, uint32_t nblocks)
{
    __shared__ uint32_t sh[1024];
    const uint32_t per = (nblocks + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per < nblocks ? threadIdx.x * per : nblocks, hi = lo + per < nblocks ? lo + per : nblocks;
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; ++k) s += block_sums[k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        uint32_t add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t v = block_sums[k]; block_sums[k] = run; run += v; }
}

__global__ __launch_bounds__(256) void sort_scan_add_kernel(uint32_t *data, const uint32_t *block_sums, uint32_t n)
{
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) data[base + k] += add;
}

RadixPlan make_plan(int begin_bit, int end_bit)
{
    RadixPlan p{};
    const int total = end_bit - begin_bit;
    p.passes = (total + (int)kRadixBits - 1) / (int)kRadixBits;
    if (p.passes < 1) p.passes = 1;
    // equal digits (21 bits -> 3 x 7 rather than 8 + 8 + 5)
    const int per = (total + p.passes - 1) / p.passes;
    int at = begin_bit;
    for (int q = 0; q < p.passes; ++q) {
        const int b = end_bit - at < per ? end_bit - at : per;
        p.shift[q] = at; p.bits[q] = (uint32_t)(b > 0 ? b : 1);
        at += b;
    }
    return p;
}

uint32_t radix_blocks(uint32_t n) { return (n + kRadixBlock - 1) / kRadixBlock; }

}  // namespace

size_t radix_sort_temp_bytes(uint32_t n, int begin_bit, int end_bit)
{
    const RadixPlan p = make_plan(begin_bit, end_bit);
    const size_t per_pass = ((size_t)1 << kRadixBits) * radix_blocks(n);
    const size_t sums = (per_pass + kScanBlock - 1) / kScanBlock;
    (void)p;
    return (per_pass + sums + 64) * sizeof(uint32_t);
}

// Sorts n pairs by key bits [begin_bit, end_bit), stable.  The passes ping-pong between the (a) and (b) buffers;
// returns 0 when the result is in (a), 1 when it is in (b).  iota: the values are the elements' positions 0..n-1 (vals_a is
// then only written).
template <typename K>
static int radix_sort_pairs(K *keys_a, uint32_t *vals_a, K *keys_b, uint32_t *vals_b, uint32_t n, int begin_bit, int end_bit,
                            void *temp, bool iota, hipStream_t s)
{
    if (n == 0) return 0;
    const RadixPlan plan = make_plan(begin_bit, end_bit);
    const uint32_t nblocks = radix_blocks(n);
    const size_t per_pass = ((size_t)1 << kRadixBits) * nblocks;
    uint32_t *counts = static_cast<uint32_t *>(temp);
    uint32_t *sums = counts + per_pass;
    for (int q = 0; q < plan.passes; ++q) {
        K *ki = (q & 1) ? keys_b : keys_a, *ko = (q & 1) ? keys_a : keys_b;
        const uint32_t *vi = (q & 1) ? vals_b : (q == 0 && iota ? nullptr : vals_a);
        uint32_t *vo = (q & 1) ? vals_a : vals_b;
        const uint32_t used = (1u << plan.bits[q]) * nblocks;
        const uint32_t scan_blocks = (used + kScanBlock - 1) / kScanBlock;
        hipLaunchKernelGGL((radix_hist_kernel<K>), dim3(nblocks), dim3(256), 0, s, ki, n, plan.shift[q], plan.bits[q], nblocks, counts);
        hipLaunchKernelGGL(sort_scan_local_kernel, dim3(scan_blocks), dim3(256), 0, s, counts, sums, used);
        hipLaunchKernelGGL(sort_scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, scan_blocks);
        hipLaunchKernelGGL(sort_scan_add_kernel, dim3(scan_blocks), dim3(256), 0, s, counts, sums, used);
        hipLaunchKernelGGL((radix_scatter_kernel<K>), dim3(nblocks), dim3(256), 0, s, ki, vi, ko, vo, n, plan.shift[q], plan.bits[q],
                           nblocks, counts);
    }
    return plan.passes & 1;
}

int launch_radix_sort_u32(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint32_t n, int begin_bit,
                          int end_bit, void *temp, bool iota, hipStream_t s)
{
    return radix_sort_pairs<uint32_t>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, temp, iota, s);
}

int launch_radix_sort_u64(unsigned long long *keys_a, uint32_t *vals_a, unsigned long long *keys_b, uint32_t *vals_b, uint32_t n,
                          int begin_bit, int end_bit, void *temp, bool iota, hipStream_t s)
{
    return radix_sort_pairs<unsigned long long>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, temp, iota, s);
}

}  // namespace th
