// th_kernels.hip - gfx950 (CDNA4, wave64) kernels of the Tendrils particle path.
//
// Compiled with -ffp-contract=off: see th_math.hpp.  One thread integrates one
// particle; the state ring is RGBA32F texels (16 B per lane per access =
// global_load/store_dwordx4, the widest coalesced form), the flow field is a
// random 16-B gather served by L2 / Infinity Cache, the simplex-noise gradient
// table sits in LDS.
//
// Reference behaviour implemented here (paths relative to the reference tree):
//   integrator      src/logic.frag:45-101  (+ src/flow/flow-at-screen-pos.glsl:13-27,
//                   src/flow/get.glsl:3-5, src/map/pos-to-uv.glsl:6-8, glsl-noise simplex/3d)
//   launch shape    one invocation per state texel (src/particles.js:132-143)
#include "th_kernels.hpp"
#include "th_math.hpp"
#include "th_logic.hpp"

#include <type_traits>
#include <cstdlib>

namespace th {


static int grid_for(size_t n, int blocks_per_cu)
{
    size_t blocks = (n + 255) / 256;
    size_t cap = (size_t)256 * blocks_per_cu;
    return (int)(blocks < cap ? (blocks ? blocks : 1) : cap);
}



__global__ __launch_bounds__(256) void pack_state_kernel(uint2 *dst, const float4 *src, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = pack_state(src[i]);
}

__global__ __launch_bounds__(256) void unpack_state_kernel(float4 *dst, const uint2 *src, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = unpack_state(src[i]);
}

void launch_pack_state(void *dst, const float4 *src, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(pack_state_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, (uint2 *)dst, src, n);
}

void launch_unpack_state(float4 *dst, const void *src, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(unpack_state_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, dst, (const uint2 *)src, n);
}

__global__ __launch_bounds__(256) void hash_tables_kernel(float4 *block)
{
    uint32_t *a = reinterpret_cast<uint32_t *>(block), *b = a + kPermA;
    for (int k = threadIdx.x; k < kPermA; k += 256) a[k] = 4u * (uint32_t)permute_int((float)k);
    for (int k = threadIdx.x; k < kPermB; k += 256) b[k] = 16u * (uint32_t)((int)permute_int((float)k) - kLutMin);
}
void launch_hash_tables(float4 *block, hipStream_t s) { hipLaunchKernelGGL(hash_tables_kernel, dim3(1), dim3(256), 0, s, block); }
int hash_table_vectors() { return kHashVec; }



// Plain grid-stride, the next state texel of each lane prefetched one iteration ahead.  Slots are in texel order (slot ==
// particle id) or - p.perm - in a tile-sorted order (the particle of slot s is perm[s]): a wave's taps then fall into
// one neighbourhood of the decoded field, and the gather costs what a staged window would.  The hash
// stages of the noise run through the LDS tables of the fused kernel.
template <bool FAST, bool NOISE, bool TARGET, bool POW2, bool DECODED, bool PTAB>
__global__ __launch_bounds__(256) void logic_kernel(const LogicParams p)
{
    const float time = p.time_dev ? *p.time_dev : p.u.time;     // captured-graph replays keep `time` in device memory
    // PTAB (sorted slots): hash stages from LDS tables.  In texel order the pass waits for its random taps, not for
    // arithmetic, and the tables' LDS traffic only costs (0.188 vs 0.197 ms at C3)
    __shared__ float4 smem[NOISE ? (PTAB ? kHashVec : 0) + kLutSize : 1];
    const float4 *lut = smem + (NOISE && PTAB ? kHashVec : 0);
    const HashTables tabs{reinterpret_cast<const uint32_t *>(smem), reinterpret_cast<const uint32_t *>(smem) + kPermA};
    if constexpr (NOISE) {
        if constexpr (PTAB) fill_hash_tables(smem, p.lut);
        else for (int k = threadIdx.x; k < kLutSize; k += 256) smem[k] = p.lut[k];
        __syncthreads();
    }
    const uint32_t stride = gridDim.x * 256u, end = p.count;
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    float4 nxt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t *perm = p.perm;
    uint32_t pnxt = idx;
    if (idx < end) { nxt = load_stream(&p.in[idx]); if (perm) pnxt = __builtin_nontemporal_load(&perm[idx]); }
    for (; idx < end; idx += stride) {
        float4 st = nxt;
        const uint32_t pid = perm ? pnxt : idx;
        if (idx + stride < end) { nxt = load_stream(&p.in[idx + stride]); if (perm) pnxt = __builtin_nontemporal_load(&perm[idx + stride]); }
        const float4 r = integrate<FAST, NOISE, TARGET, POW2, DECODED, NOISE && PTAB>(p, lut, st, pid, time, &tabs);
        store_stream(&p.out[idx], r);
        if (p.seen) {           // (uniform; a frame loop: LogicParams::seen)
            // hidden for sure: both ends beyond one edge (a NaN compares false: seen)
            const bool hidden = (st.x < p.seen_xlo && r.x < p.seen_xlo) || (st.x > p.seen_xhi && r.x > p.seen_xhi) ||
                                (st.y < p.seen_ylo && r.y < p.seen_ylo) || (st.y > p.seen_yhi && r.y > p.seen_yhi);
            const unsigned long long any = __ballot(!hidden);
            if (__lane_id() == (uint32_t)__builtin_ctzll(__ballot(true))) p.seen[idx >> 6] = any ? 1u : 0u;
        }
    }
}

// Packed-state integrator (TH_STATE_F16): same per-particle arithmetic on the decoded texel, 8 B in / 8 B out.
template <bool FAST, bool NOISE, bool TARGET, bool POW2, bool DECODED, bool PTAB>
__global__ __launch_bounds__(256) void logic_packed_kernel(const LogicParams p)
{
    const float time = p.time_dev ? *p.time_dev : p.u.time;
    __shared__ float4 smem[NOISE ? (PTAB ? kHashVec : 0) + kLutSize : 1];      // (PTAB: sorted slots, as logic_kernel)
    const float4 *lut = smem + (NOISE && PTAB ? kHashVec : 0);
    const HashTables tabs{reinterpret_cast<const uint32_t *>(smem), reinterpret_cast<const uint32_t *>(smem) + kPermA};
    if constexpr (NOISE) {
        if constexpr (PTAB) fill_hash_tables(smem, p.lut);
        else for (int k = threadIdx.x; k < kLutSize; k += 256) smem[k] = p.lut[k];
        __syncthreads();
    }
    const v2u *in = reinterpret_cast<const v2u *>(p.in);
    v2u *out = reinterpret_cast<v2u *>(p.out);
    const uint32_t stride = gridDim.x * 256u;
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    v2u nxt = {0x80008000u, 0u};
    const uint32_t *perm = p.perm;          // tile-sorted slots: the particle of slot s is perm[s] (as logic_kernel)
    uint32_t pnxt = idx;
    if (idx < p.count) { nxt = __builtin_nontemporal_load(&in[idx]); if (perm) pnxt = __builtin_nontemporal_load(&perm[idx]); }
    for (; idx < p.count; idx += stride) {
        v2u w = nxt;
        const uint32_t pid = perm ? pnxt : idx;
        if (idx + stride < p.count) { nxt = __builtin_nontemporal_load(&in[idx + stride]); if (perm) pnxt = __builtin_nontemporal_load(&perm[idx + stride]); }
        float4 r = integrate<FAST, NOISE, TARGET, POW2, DECODED, NOISE && PTAB>(p, lut, unpack_state(make_uint2(w.x, w.y)), pid, time, &tabs);
        uint2 q = pack_state(r);
        v2u qq = {q.x, q.y};
        __builtin_nontemporal_store(qq, &out[idx]);
    }
}

// Generic kernel: reference-order evaluation of every texel (used when the host
// cannot establish the fast path's preconditions, e.g. non-finite uniforms).
__global__ __launch_bounds__(256) void logic_generic_kernel(const LogicParams p)
{
    const float time = p.time_dev ? *p.time_dev : p.u.time;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < p.count; idx += stride) {
        float4 st = p.in[idx];
        uint32_t y = idx / p.width, x = idx - y * p.width;
        p.out[idx] = logic_texel_ref(p, x, y + p.row0, st, idx, time);
    }
}

template <bool FAST, bool NOISE, bool TARGET>
static void launch_logic_p2(const LogicParams &p, bool pow2, bool decoded, hipStream_t s)
{
    // 7 workgroups per CU are resident: 2048 workgroups (8 per CU) ran as 1792 + a second round of 256; 20 per CU ends evenly
    const int grid = grid_for(p.count, 20);
#define TH_GO(P2, DEC) do { if (p.perm) hipLaunchKernelGGL((logic_kernel<FAST, NOISE, TARGET, P2, DEC, true>), dim3(grid), dim3(256), 0, s, p); \
                            else hipLaunchKernelGGL((logic_kernel<FAST, NOISE, TARGET, P2, DEC, false>), dim3(grid), dim3(256), 0, s, p); } while (0)
    if (pow2) { if (decoded) TH_GO(true, true); else TH_GO(true, false); }
    else { if (decoded) TH_GO(false, true); else TH_GO(false, false); }
#undef TH_GO
}

template <bool FAST, bool NOISE, bool TARGET>
static void launch_packed_p2(const LogicParams &p, bool pow2, bool decoded, hipStream_t s)
{
    // 64 VGPRs in texel order (8 workgroups per CU resident: a grid of 8 per CU), 68 with the hash tables over sorted slots (7)
    int grid = grid_for(p.count, p.perm ? 20 : 8);
#define TH_GO(P2, DEC) do { if (p.perm) hipLaunchKernelGGL((logic_packed_kernel<FAST, NOISE, TARGET, P2, DEC, true>), dim3(grid), dim3(256), 0, s, p); \
                            else hipLaunchKernelGGL((logic_packed_kernel<FAST, NOISE, TARGET, P2, DEC, false>), dim3(grid), dim3(256), 0, s, p); } while (0)
    if (pow2) { if (decoded) TH_GO(true, true); else TH_GO(true, false); }
    else { if (decoded) TH_GO(false, true); else TH_GO(false, false); }
#undef TH_GO
}

void launch_logic(const LogicParams &p, int mode, bool noise, bool target, bool pow2, bool decoded,
                  bool generic, bool packed, hipStream_t s)
{
    if (generic) {          // texel-order f32 only (the host unpacks around it)
        hipLaunchKernelGGL(logic_generic_kernel, dim3(grid_for(p.count, 8)), dim3(256), 0, s, p);
        return;
    }
    const bool fast = mode == TH_MODE_FAST;
#define TH_DISPATCH(F, N, T)                                             \
    do {                                                                 \
        if (packed) launch_packed_p2<F, N, T>(p, pow2, decoded, s);      \
        else launch_logic_p2<F, N, T>(p, pow2, decoded, s);              \
    } while (0)
    if (fast) {
        if (noise) { if (target) TH_DISPATCH(true, true, true); else TH_DISPATCH(true, true, false); }
        else { if (target) TH_DISPATCH(true, false, true); else TH_DISPATCH(true, false, false); }
    } else {
        if (noise) { if (target) TH_DISPATCH(false, true, true); else TH_DISPATCH(false, true, false); }
        else { if (target) TH_DISPATCH(false, false, true); else TH_DISPATCH(false, false, false); }
    }
#undef TH_DISPATCH
}

// ---------------------------------------------------------------------------
// Temporal fusion for th_step_n: `nsteps` consecutive steps of one particle in ONE pass.  Particles
// are independent and flow / targets do not change inside a th_step_n call, so the intermediate
// states need not travel through HBM: the pass reads state 0 once and writes only the two states a
// 2-buffer ring keeps (state nsteps -> out, state nsteps-1 -> out_prev; each lane touches only its
// own texel, so out_prev may alias the input).  Streamed bytes per particle-step drop from 32 to
// 48/nsteps; every step still performs its own flow tap and the full arithmetic, in the same
// operation order - results are bit-identical to nsteps separate launches.
// The flow tap reads the RGBA32F texel and decodes per particle (one decoded plane per step time
// would multiply the gather footprint by nsteps).
// ---------------------------------------------------------------------------
constexpr bool kFusedPermTable = true;      // hash stages through the LDS tables (snoise_corners_tab)

// The statistics of th_stats (stats_kernel below: same classification, same arithmetic per particle) taken by the launch that
// has the state in registers: a lane's share, then one StatsPartial per workgroup (plain store; folded in a fixed order by
// launch_stats_fold) - instead of a pass that reads the whole state again (268 MB at C3).
// Nothing of it lives in registers across integrate(), no LDS, no barrier: right after its stores a wave reduces its 64
// particles' share and lane 0 stores it as the wave's own partial (a workgroup's tail - barrier, fold by one thread, store -
// cost 3 % of a launch of 65 536 short workgroups).  `first`: the wave's first round overwrites, later rounds add.
TH_D void stats_take(StatsPartial *slot, const float4 &v, float limit, bool first)
{
    const float cap = limit * (1.0f - 9.5367431640625e-07f);
    const bool is_live = v.x != kInert || v.y != kInert;
    const bool is_nan = (v.x != v.x) || (v.y != v.y) || (v.z != v.z) || (v.w != v.w);
    const float sp = __builtin_sqrtf(v.z * v.z + v.w * v.w);
    const bool fin = is_live && !is_nan && sp <= 3.4028234e38f;
    StatsPartial t;
    t.live = __builtin_popcountll(__ballot(is_live)); t.nan = __builtin_popcountll(__ballot(is_nan));
    t.capped = __builtin_popcountll(__ballot(fin && sp >= cap));
    double sum = fin ? (double)sp : 0.0;
    float mx = fin ? sp : 0.0f;
    // (through the LDS crossbar: this launch is bound by VALU issue, and the same reduction on the VALU's data-parallel
    // path - row scans, row broadcasts - cost twice as much: profiles/r4_b_fused_statistics.txt)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        const float om = __shfl_xor(mx, o); mx = om > mx ? om : mx;
    }
    if (__lane_id() == (uint32_t)__builtin_ctzll(__ballot(true))) {
        t.sum_speed = sum; t.max_speed = (double)mx;
        if (!first) {
            const StatsPartial o = *slot;
            t.live += o.live; t.nan += o.nan; t.capped += o.capped; t.sum_speed += o.sum_speed;
            t.max_speed = o.max_speed > t.max_speed ? o.max_speed : t.max_speed;
        }
        *slot = t;
    }
}

// a wave that met no particle (beyond the last slot of its share) leaves an empty partial: no memset in front of the launch
TH_D void stats_none(StatsPartial *slot, bool first)
{
    if (__ballot(!first) == 0ull && __lane_id() == 0u) *slot = StatsPartial{0ull, 0ull, 0ull, 0.0, 0.0};
}

template <bool FAST, bool NOISE, bool TARGET, bool POW2, bool BUCKETED, bool STATS>
__global__ __launch_bounds__(256) void logic_fused_kernel(const LogicParams p)
{
    // one LDS block: [permA | permB | gradient table], the hash tables first so that their reads need no base offset
    __shared__ float4 smem[NOISE ? kHashVec + kLutSize : 1];
    const float4 *lut = smem + (NOISE ? kHashVec : 0);
    const HashTables tabs{reinterpret_cast<const uint32_t *>(smem), reinterpret_cast<const uint32_t *>(smem) + kPermA};
    if constexpr (NOISE) {
        fill_hash_tables(smem, p.lut);
        __syncthreads();
    }
    uint32_t idx, stride, end;
    if constexpr (!BUCKETED) {
        idx = blockIdx.x * 256u + threadIdx.x;
        stride = gridDim.x * 256u;
        end = p.count;
    } else {
        const uint32_t group = blockIdx.x & 7u, rank = blockIdx.x >> 3, per = (p.count + 7u) >> 3;
        const uint32_t lo = group * per;
        idx = lo + rank * 256u + threadIdx.x;
        stride = (gridDim.x >> 3) * 256u;
        end = lo + per < p.count ? lo + per : p.count;
    }
    float4 nxt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t npid = idx;
    [[maybe_unused]] bool first = true;
    if (idx < end) {
        nxt = load_stream(&p.in[idx]);
        if constexpr (BUCKETED) npid = __builtin_nontemporal_load(&p.perm[idx]);
    }
    for (; idx < end; idx += stride) {
        float4 st = nxt, prev = nxt;
        uint32_t pid = BUCKETED ? npid : idx;
        if (idx + stride < end) {
            nxt = load_stream(&p.in[idx + stride]);
            if constexpr (BUCKETED) npid = __builtin_nontemporal_load(&p.perm[idx + stride]);
        }
        for (uint32_t k = 0; k < p.nsteps; ++k) {
            prev = st;
            st = integrate<FAST, NOISE, TARGET, POW2, false, kFusedPermTable>(p, lut, st, pid, p.times[k], &tabs);
        }
        store_stream(&p.out_prev[idx], prev);
        store_stream(&p.out[idx], st);
        if constexpr (STATS) { stats_take(&p.stats_part[blockIdx.x * 4u + (threadIdx.x >> 6)], st, p.u.speedLimit, first); first = false; }
    }
    if constexpr (STATS) stats_none(&p.stats_part[blockIdx.x * 4u + (threadIdx.x >> 6)], first);
}

// Packed ring (TH_STATE_F16): the same fusion on 8-B texels.  The storage quantisation is part of every step
// (a step reads what the previous one stored), so each intermediate state goes through pack -> unpack in
// registers: bit-identical to nsteps logic_packed_kernel launches.
template <bool FAST, bool NOISE, bool TARGET, bool POW2, bool BUCKETED, bool STATS>
__global__ __launch_bounds__(256, 5) void logic_fused_packed_kernel(const LogicParams p)
{
    __shared__ float4 smem[NOISE ? kHashVec + kLutSize : 1];
    const float4 *lut = smem + (NOISE ? kHashVec : 0);
    const HashTables tabs{reinterpret_cast<const uint32_t *>(smem), reinterpret_cast<const uint32_t *>(smem) + kPermA};
    if constexpr (NOISE) {
        fill_hash_tables(smem, p.lut);
        __syncthreads();
    }
    const v2u *in = reinterpret_cast<const v2u *>(p.in);
    v2u *out = reinterpret_cast<v2u *>(p.out), *out_prev = reinterpret_cast<v2u *>(p.out_prev);
    uint32_t idx, stride, end;
    if constexpr (!BUCKETED) {
        idx = blockIdx.x * 256u + threadIdx.x;
        stride = gridDim.x * 256u;
        end = p.count;
    } else {            // tile-sorted slots (p.perm), dealt to the 8 XCD groups in eighths like the f32 pass
        const uint32_t group = blockIdx.x & 7u, rank = blockIdx.x >> 3, per = (p.count + 7u) >> 3;
        const uint32_t lo = group * per;
        idx = lo + rank * 256u + threadIdx.x;
        stride = (gridDim.x >> 3) * 256u;
        end = lo + per < p.count ? lo + per : p.count;
    }
    v2u nxt = {0x80008000u, 0u};
    uint32_t npid = idx;
    [[maybe_unused]] bool first = true;
    if (idx < end) {
        nxt = __builtin_nontemporal_load(&in[idx]);
        if constexpr (BUCKETED) npid = __builtin_nontemporal_load(&p.perm[idx]);
    }
    for (; idx < end; idx += stride) {
        uint2 w = make_uint2(nxt.x, nxt.y), wprev = w;
        const uint32_t pid = BUCKETED ? npid : idx;
        if (idx + stride < end) {
            nxt = __builtin_nontemporal_load(&in[idx + stride]);
            if constexpr (BUCKETED) npid = __builtin_nontemporal_load(&p.perm[idx + stride]);
        }
        // (between two fused steps the state is what the ring WOULD hold of it - quantize_state = unpack of pack - in registers as
        // floats; the words are made once, for the two states that leave)
        float4 st = unpack_state(w);
        for (uint32_t k = 0; k < p.nsteps; ++k) {
            // (the state before the last step leaves as words at once: carried through the loop as four more floats the kernel
            // had 101 VGPRs - four waves per SIMD instead of five)
            if (k + 1u == p.nsteps && k) wprev = pack_state(st);
            st = quantize_state(integrate<FAST, NOISE, TARGET, POW2, false, kFusedPermTable>(p, lut, st, pid, p.times[k], &tabs));
        }
        if (p.nsteps) w = pack_state(st);                         // (one step: wprev stays the word that came in, whatever it decodes to)
        v2u a = {wprev.x, wprev.y}, b = {w.x, w.y};
        __builtin_nontemporal_store(a, &out_prev[idx]);
        __builtin_nontemporal_store(b, &out[idx]);
        // (the statistics of a packed ring are those of what its texels decode to - what th_stats reads through its f32 view)
        if constexpr (STATS) { stats_take(&p.stats_part[blockIdx.x * 4u + (threadIdx.x >> 6)], st, p.u.speedLimit, first); first = false; }
    }
    if constexpr (STATS) stats_none(&p.stats_part[blockIdx.x * 4u + (threadIdx.x >> 6)], first);
}

// Launch shape of the fused passes: one 256-slot workgroup per 256 particles (no persistent grid).  A fused pass
// is issue-bound and runs 5 workgroups per CU (82 VGPRs): a persistent grid of 2048 workgroups left the CUs
// unevenly loaded at the end of the pass (1280 resident + a 768-workgroup second round); with one short
// workgroup per 256 slots the dispatcher keeps every CU full until the last few (C3, exact: 0.098 -> 0.086 ms per
// step, profiles/r2_a_grid_sweep.txt).
static int fused_grid(uint32_t count, bool bucketed)
{
    const uint32_t blocks = (count + 255u) / 256u;
    // bucketed: 8 XCD groups, group g = blockIdx % 8 sweeps its eighth of the slots
    const uint32_t grid = bucketed ? 8u * ((((count + 7u) >> 3) + 255u) / 256u) : blocks;
    return (int)(grid ? grid : 1u);
}

template <bool FAST, bool NOISE, bool TARGET>
static void launch_fused_p2(const LogicParams &p, bool pow2, bool packed, hipStream_t s)
{
    if (packed) {
        const bool sorted = p.perm != nullptr;
        const int pgrid = fused_grid(p.count, sorted);
#define TH_GO(P2, BK)                                                                                                                 \
    do {                                                                                                                              \
        if (p.stats_part) hipLaunchKernelGGL((logic_fused_packed_kernel<FAST, NOISE, TARGET, P2, BK, true>), dim3(pgrid), dim3(256), 0, s, p);   \
        else hipLaunchKernelGGL((logic_fused_packed_kernel<FAST, NOISE, TARGET, P2, BK, false>), dim3(pgrid), dim3(256), 0, s, p);               \
    } while (0)
        if (pow2) { if (sorted) TH_GO(true, true); else TH_GO(true, false); }
        else { if (sorted) TH_GO(false, true); else TH_GO(false, false); }
#undef TH_GO
        return;
    }
    const bool bucketed = p.perm != nullptr;
    const int grid = fused_grid(p.count, bucketed);
#define TH_GO(P2, BK)                                                                                                         \
    do {                                                                                                                      \
        if (p.stats_part) hipLaunchKernelGGL((logic_fused_kernel<FAST, NOISE, TARGET, P2, BK, true>), dim3(grid), dim3(256), 0, s, p);   \
        else hipLaunchKernelGGL((logic_fused_kernel<FAST, NOISE, TARGET, P2, BK, false>), dim3(grid), dim3(256), 0, s, p);               \
    } while (0)
    if (pow2) { if (bucketed) TH_GO(true, true); else TH_GO(true, false); }
    else { if (bucketed) TH_GO(false, true); else TH_GO(false, false); }
#undef TH_GO
}

void launch_logic_fused(const LogicParams &p, int mode, bool noise, bool target, bool pow2, bool packed, hipStream_t s)
{
    const bool fast = mode == TH_MODE_FAST;
#define TH_DISPATCH(F, N, T) launch_fused_p2<F, N, T>(p, pow2, packed, s)
    if (fast) {
        if (noise) { if (target) TH_DISPATCH(true, true, true); else TH_DISPATCH(true, true, false); }
        else { if (target) TH_DISPATCH(true, false, true); else TH_DISPATCH(true, false, false); }
    } else {
        if (noise) { if (target) TH_DISPATCH(false, true, true); else TH_DISPATCH(false, true, false); }
        else { if (target) TH_DISPATCH(false, false, true); else TH_DISPATCH(false, false, false); }
    }
#undef TH_DISPATCH
}

// ---------------------------------------------------------------------------
// Tile-sorted slot order.  Particles are independent, so the ring buffers may hold them in any
// SLOT order as long as `perm[slot]` names the particle (its texel: gl_FragCoord, uv, the index i
// and the targets texel all derive from it).  Sorting the slots by the kTile x kTile tile of the
// flow field the particle taps turns the one random gather of the integrator (src/flow/
// flow-at-screen-pos.glsl:13-27) into a workgroup-local LDS lookup:
//   key      tile of the tap texel (same arithmetic as the tap itself); particles that tap nothing
//            (inert, NaN / infinite position) form one extra class after the last tile
//   sort     counting sort: tile_hist_kernel -> tile_scan_kernel (tile starts, rank cursors and the
//            chunk table: <= kTileChunk slots of ONE tile per workgroup) -> the slots are assigned by
//            the kernel that moves the state anyway: logic_sorted_kernel<.., SCATTER = true> writes the
//            step's OUTPUT at the new slots (no extra pass over the state), tile_scatter_kernel is
//            the plain move for launches that do not step (th_step_n's fused passes)
//   launch   one workgroup per chunk stages its tile + kTileHalo texels of the decoded plane in LDS;
//            chunks are dealt to the XCDs in eighths (blockIdx % 8 = XCD group, checked with
//            HW_REG_XCC_ID in tools/xcc_census.hip), so one XCD's L2 serves one band of the field.
// Slot order within a tile is whatever the rank atomics produce (it differs from run to run);
// results do not depend on it: every particle reads only its own texel (src/logic.frag:48,75,85).
// ---------------------------------------------------------------------------
// Sort class of particle `pid` at (px, py): 2 * tile + (its line never draws).  Whether draw() can make a line of a particle
// at all is a fixed property of its row in the state texture (g.row_draws; th_api.hip: line_rows) - about half of the rows
// cannot: their two vertices read the same texel of the same buffer (src/state/state-at-frame.glsl:12-22).  Keeping the two
// kinds apart inside every tile lets the draw passes (th_bins.hip) run whole waves of lines that exist, not half-empty ones.
TH_D uint32_t tile_key(const TileGeom &g, float px, float py, uint32_t pid)
{
    const uint32_t row = g.row0 + (g.pow2w ? pid >> g.log2w : pid / g.width);
    const uint32_t idle = ((g.row_draws[row >> 5] >> (row & 31u)) & 1u) ^ 1u;
    const bool taps = __builtin_fabsf(px) < __builtin_inff() && __builtin_fabsf(py) < __builtin_inff() &&
                      (px != kInert || py != kInert);
    if (!taps) return 2u * g.ntiles + idle;
    // the tap texel of integrate(): (pos * viewSize + 1) * (0.5 * size), clamped, truncated
    const int tx = (int)__builtin_amdgcn_fmed3f((px * g.view_x + 1.0f) * g.half_fw, 0.0f, g.fwm1);
    const int ty = (int)__builtin_amdgcn_fmed3f((py * g.view_y + 1.0f) * g.half_fh, 0.0f, g.fhm1);
    return 2u * ((uint32_t)(ty >> kTileShift) * g.tiles_x + (uint32_t)(tx >> kTileShift)) + idle;
}

// Runs of equal keys among the valid lanes of a wave (valid lanes are a prefix of the wave).  In sorted input a
// wave holds one or two runs, so a counter per run instead of per lane keeps the atomics off the hot tiles.
//   head      this lane starts a run
//   rank      position of this lane inside its run
//   length    (head lanes) lanes in the run
//   head_lane lane that starts this lane's run
struct WaveRuns { bool head; uint32_t rank, length, head_lane; };
TH_D WaveRuns wave_runs(uint32_t key, bool valid)
{
    const uint32_t lane = __lane_id();
    const uint32_t prev = __shfl_up(key, 1);
    const unsigned long long vmask = __ballot(valid);
    const bool head = valid && (lane == 0 || key != prev);
    const unsigned long long heads = __ballot(head);
    WaveRuns r;
    r.head = head;
    const unsigned long long below = heads & ((2ull << lane) - 1ull);          // heads at or below this lane
    r.head_lane = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
    r.rank = lane - r.head_lane;
    const unsigned long long above = lane < 63u ? (heads >> (lane + 1u)) : 0ull;   // heads after this lane
    const uint32_t next = above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : (uint32_t)__builtin_popcountll(vmask);
    r.length = next - lane;
    return r;
}

// Counting through a small LDS table.  Global atomics on a few thousand adjacent counters are slow (16.8 M particles,
// one atomic per run of a wave: 4.7 ms per pass), and a workgroup sweeping sorted slots meets only its own tile and
// the tiles next to it, so every workgroup first counts in an open-addressed LDS table of kBinSlots keys and touches
// the global counters once per distinct key; runs that find the table full go to the global counter directly.
constexpr uint32_t kBinEmpty = 0xffffffffu;
struct ChunkBins { uint32_t key[kBinSlots], count[kBinSlots], base[kBinSlots]; };

TH_D void bins_clear(ChunkBins &t)
{
    if (threadIdx.x < kBinSlots) { t.key[threadIdx.x] = kBinEmpty; t.count[threadIdx.x] = 0u; t.base[threadIdx.x] = 0u; }
}
// table slot of `key` (claimed when absent and `insert`), -1 = not in the table / table full
TH_D int bins_slot(ChunkBins &t, uint32_t key, bool insert)
{
    uint32_t h = (key * 2654435761u) >> (32 - kBinSlotsLog2);
    for (uint32_t n = 0; n < kBinSlots; ++n, h = (h + 1u) & (kBinSlots - 1u)) {
        const uint32_t k = *(volatile uint32_t *)&t.key[h];
        if (k == key) return (int)h;
        if (k == kBinEmpty) {
            if (!insert) return -1;
            const uint32_t old = atomicCAS(&t.key[h], kBinEmpty, key);
            if (old == kBinEmpty || old == key) return (int)h;
        }
    }
    return -1;
}
// The global counters themselves are kept in kSortReplicas copies kMaxTileBins words apart (a few thousand adjacent
// words sit on a handful of memory channels, where the atomics of every workgroup would queue up: 16.8 M particles
// in texel order, 4.7 ms per counting pass on one copy against 0.4 ms on 64).  The pass that counts and the pass that
// hands out the slots must pick the same copy for a particle: `rep` is the input slot's 4096-block for passes over
// linear blocks, the chunk number for passes that work from a chunk's record.
TH_D uint32_t replica_of(uint32_t block) { return (block & (kSortReplicas - 1u)) * kMaxTileBins; }

// count one run of a wave (call from its head lane)
TH_D void bins_count(ChunkBins &t, uint32_t *global_hist, uint32_t rep, uint32_t key, uint32_t length)
{
    const int h = bins_slot(t, key, true);
    if (h >= 0) atomicAdd(&t.count[h], length); else atomicAdd(&global_hist[rep + key], length);
}
// after a barrier: the table's totals to the global histogram (and, optionally, the table itself to `rec`)
TH_D void bins_flush(const ChunkBins &t, uint32_t *global_hist, uint32_t rep, ChunkRecord *rec)
{
    if (threadIdx.x < kBinSlots) {
        const uint32_t k = t.key[threadIdx.x], n = t.count[threadIdx.x];
        if (k != kBinEmpty && n) atomicAdd(&global_hist[rep + k], n);
        if (rec) { rec->key[threadIdx.x] = k; rec->count[threadIdx.x] = n; }
    }
}

// position of slot `at` of a state buffer in either storage format
template <bool PACKED>
TH_D float2 slot_position(const float4 *state, uint32_t at)
{
    if constexpr (PACKED) {
        const float4 s = unpack_state(make_uint2(reinterpret_cast<const uint2 *>(state)[at].x, 0u));
        return make_float2(s.x, s.y);
    } else return *reinterpret_cast<const float2 *>(&state[at]);
}

template <bool PACKED>
__global__ __launch_bounds__(256) void tile_hist_kernel(const TileSortParams b)
{
    __shared__ ChunkBins bins;
    bins_clear(bins);
    __syncthreads();
    const uint32_t base = blockIdx.x * kTileChunk;
    // all 16 positions of a lane first (unconditional, clamped: nothing waits under a branch), then the counting
    float2 pos[kTileChunk / 256u];
    uint32_t pid[kTileChunk / 256u];
#pragma unroll
    for (uint32_t k = 0; k < kTileChunk / 256u; ++k) {
        const uint32_t s = base + k * 256u + threadIdx.x, at = s < b.count ? s : b.count - 1u;
        pos[k] = slot_position<PACKED>(b.state, at);
        pid[k] = b.perm_in ? b.perm_in[at] : at;
    }
#pragma unroll
    for (uint32_t k = 0; k < kTileChunk / 256u; ++k) {
        const uint32_t s = base + k * 256u + threadIdx.x;
        const bool valid = s < b.count;
        const uint32_t key = valid ? tile_key(b.g, pos[k].x, pos[k].y, pid[k]) : 0u;
        const WaveRuns r = wave_runs(key, valid);
        if (r.head) bins_count(bins, b.hist, replica_of(blockIdx.x), key, r.length);
    }
    __syncthreads();
    bins_flush(bins, b.hist, replica_of(blockIdx.x), b.block_records ? &b.block_records[blockIdx.x] : nullptr);
}

// Exclusive scan of the histogram (bins = tiles + the no-tap class, summed over the copies) into the first slot of every
// bin, the rank cursors of every copy (copy r of a bin starts where copies < r end) and the chunk table; clears the
// histogram.  Three launches: one workgroup doing all of it walked 64 copies x 4 bins per thread, one dependent global access
// after the other, twice - 70-220 us in which nothing else ran, inside the one frame in 64 that re-sorts.
//   tile_scan_sum_kernel    a thread per bin: the copies' counts (64 independent, coalesced loads) summed; every copy's cursor
//                           relative to the bin's first slot; the histogram cleared; the bin's total to b.totals
//   tile_scan_kernel        one workgroup: the totals scanned (4 bins per thread), every bin's first slot to b.starts, the chunk table
//   tile_scan_place_kernel  a thread per bin: the bin's first slot added to its copies' cursors
__global__ __launch_bounds__(256) void tile_scan_sum_kernel(const TileSortParams b)
{
    const uint32_t bins = 2u * b.g.ntiles + 2u, k = blockIdx.x * 256u + threadIdx.x;
    if (k >= bins) return;
    uint32_t h[kSortReplicas];
#pragma unroll
    for (uint32_t r = 0; r < kSortReplicas; ++r) h[r] = b.hist[(size_t)r * kMaxTileBins + k];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t r = 0; r < kSortReplicas; ++r) {
        const size_t w = (size_t)r * kMaxTileBins + k;
        b.hist[w] = 0u;
        b.cursor[w] = sum;
        sum += h[r];
    }
    b.totals[k] = sum;
}

__global__ __launch_bounds__(1024) void tile_scan_kernel(const TileSortParams b)
{
    __shared__ uint32_t part_n[1024], part_c[1024];
    const uint32_t bins = 2u * b.g.ntiles + 2u, per = (bins + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per < bins ? threadIdx.x * per : bins, hi = lo + per < bins ? lo + per : bins;
    uint32_t n = 0, c = 0;
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t h = b.totals[k];
        n += h; c += (h + kTileChunk - 1u) / kTileChunk;
    }
    part_n[threadIdx.x] = n; part_c[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {          // Hillis-Steele inclusive scan of the per-thread totals
        uint32_t an = 0, ac = 0;
        if (threadIdx.x >= off) { an = part_n[threadIdx.x - off]; ac = part_c[threadIdx.x - off]; }
        __syncthreads();
        part_n[threadIdx.x] += an; part_c[threadIdx.x] += ac;
        __syncthreads();
    }
    uint32_t slot = part_n[threadIdx.x] - n, chunk = part_c[threadIdx.x] - c;
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t h = b.totals[k];
        b.starts[k] = slot;
        for (uint32_t done = 0; done < h; done += kTileChunk, ++chunk)
            b.chunks[chunk] = TileChunk{slot + done, h - done < kTileChunk ? h - done : kTileChunk, k, 0u};
        slot += h;
    }
    if (threadIdx.x == 1023u) *b.nchunks = part_c[1023];
}

__global__ __launch_bounds__(256) void tile_scan_place_kernel(const TileSortParams b)
{
    const uint32_t bins = 2u * b.g.ntiles + 2u, k = blockIdx.x * 256u + threadIdx.x;
    if (k >= bins) return;
    const uint32_t start = b.starts[k];
#pragma unroll
    for (uint32_t r = 0; r < kSortReplicas; ++r) b.cursor[(size_t)r * kMaxTileBins + k] += start;
}

// new slot of every valid lane: one returning atomic per run of equal (copy, key) words
TH_D uint32_t reserve_slots(uint32_t *cursor, uint32_t word, bool valid)
{
    const WaveRuns r = wave_runs(word, valid);
    uint32_t first = 0;
    if (r.head) first = atomicAdd(&cursor[word], r.length);
    return __shfl(first, r.head_lane) + r.rank;
}
// the same through a workgroup's table of reserved ranges (keys outside the table: the global cursor of copy `rep`)
TH_D uint32_t reserve_slots(ChunkBins &t, uint32_t *cursor, uint32_t rep, uint32_t key, bool valid)
{
    const WaveRuns r = wave_runs(key, valid);
    uint32_t first = 0;
    if (r.head) {
        const int h = bins_slot(t, key, false);
        first = h >= 0 ? t.base[h] + atomicAdd(&t.count[h], r.length) : atomicAdd(&cursor[rep + key], r.length);
    }
    return __shfl(first, r.head_lane) + r.rank;
}

// The plain move: state (and particle ids) of slot order `perm_in` (nullptr = texel order) to the sorted slots.  With the
// block's table of (tile, count) from tile_hist_kernel every tile's range is reserved once per workgroup (one atomic each,
// all in flight together) and the ranks come from LDS; without it - and for tiles that did not fit the table - every
// wave run waits for its own round trip to the global cursor (16 of them in a row per lane: 0.93 ms at C3 against 0.3).
template <bool PACKED>
__global__ __launch_bounds__(256) void tile_scatter_kernel(const TileSortParams b)
{
    using Texel = std::conditional_t<PACKED, v2u, v4f>;          // 8-byte packed texels or RGBA32F
    __shared__ ChunkBins bins;
    const bool tabled = b.block_records != nullptr;
    if (tabled) {
        if (threadIdx.x < kBinSlots) {
            const uint32_t k = b.block_records[blockIdx.x].key[threadIdx.x], n = b.block_records[blockIdx.x].count[threadIdx.x];
            bins.key[threadIdx.x] = k;
            bins.count[threadIdx.x] = 0u;
            bins.base[threadIdx.x] = (k != kBinEmpty && n) ? atomicAdd(&b.cursor[replica_of(blockIdx.x) + k], n) : 0u;
        }
        __syncthreads();
    }
    const uint32_t base = blockIdx.x * kTileChunk;
    const uint32_t *ids = b.perm_in ? b.perm_in : reinterpret_cast<const uint32_t *>(b.state);     // (texel order: a word read anyway;
                                                                                                      // both formats hold >= count words)
    // four slots of a lane per round, their loads in flight together (unconditional, clamped)
    constexpr uint32_t kRound = 4;
    for (uint32_t k0 = 0; k0 < kTileChunk / 256u; k0 += kRound) {
        Texel st[kRound];
        uint32_t pid[kRound];
#pragma unroll
        for (uint32_t q = 0; q < kRound; ++q) {
            const uint32_t s = base + (k0 + q) * 256u + threadIdx.x, at = s < b.count ? s : b.count - 1u;
            st[q] = __builtin_nontemporal_load(&reinterpret_cast<const Texel *>(b.state)[at]);
            pid[q] = __builtin_nontemporal_load(&ids[at]);
        }
#pragma unroll
        for (uint32_t q = 0; q < kRound; ++q) {
            const uint32_t s = base + (k0 + q) * 256u + threadIdx.x;
            const bool valid = s < b.count;
            float px, py;
            if constexpr (PACKED) { const float4 u = unpack_state(make_uint2(st[q].x, 0u)); px = u.x; py = u.y; }
            else { px = st[q].x; py = st[q].y; }
            const uint32_t key = valid ? tile_key(b.g, px, py, b.perm_in ? pid[q] : s) : 0u;
            const uint32_t d = tabled ? reserve_slots(bins, b.cursor, replica_of(blockIdx.x), key, valid)
                                      : reserve_slots(b.cursor, replica_of(blockIdx.x) + key, valid);
            if (valid) { reinterpret_cast<Texel *>(b.state_out)[d] = st[q]; b.perm_out[d] = b.perm_in ? pid[q] : s; }
        }
    }
}

// back to texel order: dst[perm[s]] = src[s]
template <typename Texel>
__global__ __launch_bounds__(256) void unpermute_state_kernel(Texel *dst, const Texel *src, const uint32_t *perm, uint32_t n)
{
    for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < n; s += gridDim.x * 256u) dst[perm[s]] = src[s];
}

// texel order into a slot order: dst[s] = src[perm[s]]
template <typename Texel>
__global__ __launch_bounds__(256) void permute_state_kernel(Texel *dst, const Texel *src, const uint32_t *perm, uint32_t n)
{
    for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < n; s += gridDim.x * 256u) dst[s] = src[perm[s]];
}

static int tile_grid(uint32_t count) { return (int)((count + kTileChunk - 1) / kTileChunk); }

void launch_tile_hist(const TileSortParams &b, hipStream_t s)
{
    if (b.packed) hipLaunchKernelGGL(tile_hist_kernel<true>, dim3(tile_grid(b.count)), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(tile_hist_kernel<false>, dim3(tile_grid(b.count)), dim3(256), 0, s, b);
}

void launch_tile_scan(const TileSortParams &b, hipStream_t s)
{
    const uint32_t bins = 2u * b.g.ntiles + 2u, grid = (bins + 255u) / 256u;
    hipLaunchKernelGGL(tile_scan_sum_kernel, dim3(grid), dim3(256), 0, s, b);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, b);
    hipLaunchKernelGGL(tile_scan_place_kernel, dim3(grid), dim3(256), 0, s, b);
}

void launch_tile_scatter(const TileSortParams &b, hipStream_t s)
{
    if (b.packed) hipLaunchKernelGGL(tile_scatter_kernel<true>, dim3(tile_grid(b.count)), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(tile_scatter_kernel<false>, dim3(tile_grid(b.count)), dim3(256), 0, s, b);
}

void launch_unpermute_state(float4 *dst, const float4 *src, const uint32_t *perm, uint32_t n, bool packed, hipStream_t s)
{
    if (packed) hipLaunchKernelGGL(unpermute_state_kernel<uint2>, dim3(grid_for(n, 8)), dim3(256), 0, s, reinterpret_cast<uint2 *>(dst),
                                   reinterpret_cast<const uint2 *>(src), perm, n);
    else hipLaunchKernelGGL(unpermute_state_kernel<float4>, dim3(grid_for(n, 8)), dim3(256), 0, s, dst, src, perm, n);
}

void launch_permute_state(float4 *dst, const float4 *src, const uint32_t *perm, uint32_t n, bool packed, hipStream_t s)
{
    if (packed) hipLaunchKernelGGL(permute_state_kernel<uint2>, dim3(grid_for(n, 8)), dim3(256), 0, s, reinterpret_cast<uint2 *>(dst),
                                   reinterpret_cast<const uint2 *>(src), perm, n);
    else hipLaunchKernelGGL(permute_state_kernel<float4>, dim3(grid_for(n, 8)), dim3(256), 0, s, dst, src, perm, n);
}

// ---------------------------------------------------------------------------
// The single-step integrator over sorted slots.
//   IN_TILED  the input is in a tile-sorted order (p.perm, p.chunks): one chunk per workgroup, the chunk's
//             tile + halo staged in LDS, taps from there.  Otherwise the input is in texel order (the first
//             sort of a context): chunks are runs of kTileChunk texels, taps are global gathers.
//   SCATTER   the output goes to the slots of a NEW sort (key = tile of the INPUT position, counted by
//             tile_hist_kernel over the same buffer), with the particle ids in p.perm_out.  Otherwise
//             the output keeps the input's slot.
// The hash stages of the noise run through the LDS tables of the fused kernel (snoise_corners_tab).
// ---------------------------------------------------------------------------
//   COUNT     (in-place passes) also histogram the tiles of the OUTPUT positions - the input of the next pass - and
//             leave every chunk's table of (tile, count) in p.records, so that a SCATTER pass that follows needs no
//             counting pass of its own and reserves its slots once per workgroup and tile (p.use_records)
// The flow taps are gathered from the decoded plane like the plain passes' in between (round 2 also built an LDS-staged
// window of the chunk's tile + halo here; gathered taps over the same slot order won the A/B - profiles/HISTORY.md - and
// the window was removed in round 3).
template <bool FAST, bool NOISE, bool TARGET, bool POW2, bool IN_TILED, bool SCATTER, bool COUNT>
__global__ __launch_bounds__(256, 5) void logic_sorted_kernel(const LogicParams p)
{
    __shared__ ChunkBins bins;
    __shared__ float4 smem[NOISE ? kHashVec + kLutSize : 1];
    const float time = p.time_dev ? *p.time_dev : p.u.time;
    const float4 *lut = smem + (NOISE ? kHashVec : 0);
    const HashTables tabs{reinterpret_cast<const uint32_t *>(smem), reinterpret_cast<const uint32_t *>(smem) + kPermA};

    TileChunk ch;
    uint32_t c = blockIdx.x;
    if constexpr (IN_TILED) {
        // chunk of this workgroup: the chunk table is dealt to the 8 XCD groups in eighths
        const uint32_t nchunks = *p.nchunks, per = (nchunks + 7u) >> 3;
        c = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per || c >= nchunks) return;
        ch = p.chunks[c];
    } else {
        const uint32_t start = blockIdx.x * kTileChunk;
        if (start >= p.count) return;
        ch = TileChunk{start, p.count - start < kTileChunk ? p.count - start : kTileChunk, 0u, 0u};
    }
    const bool use_records = SCATTER && IN_TILED && p.use_records;
    if constexpr (COUNT) bins_clear(bins);
    if constexpr (SCATTER && IN_TILED) {
        // the table the previous (COUNT) pass left for this chunk: reserve every tile's range with one atomic
        if (use_records && threadIdx.x < kBinSlots) {
            const uint32_t k = p.records[c].key[threadIdx.x], n = p.records[c].count[threadIdx.x];
            bins.key[threadIdx.x] = k;
            bins.count[threadIdx.x] = 0u;
            bins.base[threadIdx.x] = (k != kBinEmpty && n) ? atomicAdd(&p.cursor[replica_of(c) + k], n) : 0u;
        }
    }
    if constexpr (NOISE) fill_hash_tables(smem, p.lut);
    if constexpr (NOISE || IN_TILED || COUNT) __syncthreads();        // (tables, window, bins; the record-based reservations of SCATTER)

    // The chunk is swept 256 slots at a time, two iterations of state loads ahead.  The loads are unconditional - lanes and
    // iterations beyond the chunk read its first slot instead (one line per wave) - because a load under a branch
    // makes hipcc wait for it at the join.  Workgroups start at different offsets into their chunks (and wrap
    // around): chunks of one tile are exact multiples of 64 KiB apart, and workgroups sweeping them in lockstep
    // would all sit on the same memory channels.
    const uint32_t iters = (ch.count + 255u) >> 8;
    const uint32_t first = (blockIdx.x * 5u) % iters;
    const uint32_t end = ch.start + ch.count;
    auto slot_of = [&](uint32_t it) { uint32_t k = first + it; k = k >= iters ? k - iters : k; return ch.start + (k << 8) + threadIdx.x; };
    auto load_slot = [&](uint32_t it, float4 &st, uint32_t &pid) {
        uint32_t s = slot_of(it);
        s = (it < iters && s < end) ? s : ch.start;
        st = load_stream(&p.in[s]);
        if constexpr (IN_TILED) pid = __builtin_nontemporal_load(&p.perm[s]);
    };
    auto body = [&](uint32_t it, float4 st, uint32_t pid_in) __attribute__((always_inline)) {
        const uint32_t slot = slot_of(it);
        const bool valid = slot < end;
        const uint32_t pid = IN_TILED ? pid_in : slot;
        uint32_t dst = slot;
        // (whole waves reach this point together: the rank reservation and the counting are wave-wide operations)
        if constexpr (SCATTER) {
            const uint32_t key = valid ? tile_key(p.geom, st.x, st.y, pid) : 0u;
            // (without a record the histogram came from tile_hist_kernel: copies by the input slot's 4096-block)
            if constexpr (IN_TILED) dst = use_records ? reserve_slots(bins, p.cursor, replica_of(c), key, valid)
                                                      : reserve_slots(p.cursor, replica_of(slot >> 12) + key, valid);
            else dst = reserve_slots(p.cursor, replica_of(blockIdx.x) + key, valid);
        }
        float4 r = st;
        if (valid) {
            r = integrate<FAST, NOISE, TARGET, POW2, true, true>(p, lut, st, pid, time, &tabs);
            // (scattered runs start at any slot: plain stores, so that L2 can merge the partial lines two runs share)
            if constexpr (SCATTER) { p.out[dst] = r; p.perm_out[dst] = pid; if (p.in_moved) p.in_moved[dst] = st; }
            else store_stream(&p.out[dst], r);
        }
        if constexpr (COUNT) {
            const uint32_t key = valid ? tile_key(p.geom, r.x, r.y, pid) : 0u;
            const WaveRuns w = wave_runs(key, valid);
            if (w.head) bins_count(bins, p.hist, replica_of(c), key, w.length);
        }
    };
    // three register sets, loop unrolled by three: every load goes into the set the previous body has just
    // consumed, so no value that is still in flight is ever moved between registers
    float4 q0, q1, q2;
    uint32_t p0 = 0, p1 = 0, p2 = 0;
    load_slot(0, q0, p0);
    load_slot(1, q1, p1);
    for (uint32_t it = 0; it < iters; it += 3u) {
        load_slot(it + 2u, q2, p2);
        body(it, q0, p0);
        if (it + 1u >= iters) break;
        load_slot(it + 3u, q0, p0);
        body(it + 1u, q1, p1);
        if (it + 2u >= iters) break;
        load_slot(it + 4u, q1, p1);
        body(it + 2u, q2, p2);
    }
    if constexpr (COUNT) {
        __syncthreads();
        bins_flush(bins, p.hist, replica_of(c), &p.records[c]);
    }
}

template <bool FAST, bool NOISE, bool TARGET>
static void launch_sorted_p2(const LogicParams &p, bool pow2, bool in_tiled, bool scatter, bool count, uint32_t max_chunks, hipStream_t s)
{
    // IN_TILED: an upper bound of the chunk count (the real one lives on the device), rounded up to the 8 XCD groups
    const int grid = in_tiled ? (int)(((max_chunks + 7u) & ~7u)) : tile_grid(p.count);
#define TH_GO(P2, IT, SC, CN) hipLaunchKernelGGL((logic_sorted_kernel<FAST, NOISE, TARGET, P2, IT, SC, CN>), dim3(grid), dim3(256), 0, s, p)
#define TH_GO_P2(IT, SC, CN) do { if (pow2) TH_GO(true, IT, SC, CN); else TH_GO(false, IT, SC, CN); } while (0)
    if (in_tiled) {
        if (scatter) TH_GO_P2(true, true, false);
        else TH_GO_P2(true, false, true);        // (in place: the counting pass before a re-sort; plain passes run logic_kernel)
    } else TH_GO_P2(false, true, false);
#undef TH_GO_P2
#undef TH_GO
}

void launch_logic_sorted(const LogicParams &p, int mode, bool noise, bool target, bool pow2, bool in_tiled, bool scatter,
                         bool count, uint32_t max_chunks, hipStream_t s)
{
    const bool fast = mode == TH_MODE_FAST;
#define TH_DISPATCH(F, N, T) launch_sorted_p2<F, N, T>(p, pow2, in_tiled, scatter, count, max_chunks, s)
    if (fast) {
        if (noise) { if (target) TH_DISPATCH(true, true, true); else TH_DISPATCH(true, true, false); }
        else { if (target) TH_DISPATCH(true, false, true); else TH_DISPATCH(true, false, false); }
    } else {
        if (noise) { if (target) TH_DISPATCH(false, true, true); else TH_DISPATCH(false, true, false); }
        else { if (target) TH_DISPATCH(false, false, true); else TH_DISPATCH(false, false, false); }
    }
#undef TH_DISPATCH
}

// Per-texel flow decode for one step (src/flow/get.glsl:3-5).  Sampling is NEAREST and get() is
// pointwise, so decoding per texel is bit-identical to decoding per particle; it halves the
// footprint of the random gather (16 -> 8 B per texel).
__global__ __launch_bounds__(256) void flow_decode_kernel(const float4 *flow, float2 *dec, size_t n, float time,
                                                          const float *time_dev, float decay)
{
    if (time_dev) time = *time_dev;
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float4 f = flow[i];
        float k = __builtin_fmaxf(0.0f, 1.0f - ((time - f.z) * decay));
        dec[i] = make_float2(f.x * k, f.y * k);
    }
}

// the flow texels' x, y, z alone, 12 B apart (LogicParams::flow3)
__global__ __launch_bounds__(256) void flow_pack3_kernel(const float4 *flow, float *xyz, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float4 f = flow[i];
        xyz[3 * i] = f.x; xyz[3 * i + 1] = f.y; xyz[3 * i + 2] = f.z;
    }
}
void launch_flow_pack3(const float4 *flow, float *xyz, size_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(flow_pack3_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, flow, xyz, n);
}

void launch_flow_decode(const float4 *flow, float2 *dec, size_t n, float time, const float *time_dev, float decay,
                        hipStream_t s)
{
    if (n) hipLaunchKernelGGL(flow_decode_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, flow, dec, n, time, time_dev, decay);
}

// ---------------------------------------------------------------------------
// Optical-flow producer: src/optical-flow/index.frag:55-81, one thread per flow texel,
// alpha-blended into the flow texture (SRC_ALPHA, ONE_MINUS_SRC_ALPHA: src/index.js:267-268).
// Ten NEAREST/CLAMP taps of the two RGBA8 frames; neighbouring lanes read neighbouring
// texels, so every tap is a coalesced 4-byte-per-lane load served by L2.
// ---------------------------------------------------------------------------
// NEAREST + CLAMP_TO_EDGE index into an 8-bit-per-channel texture, with the coordinate
// precision of the captured reference run: clamp to [0, 1), truncate to 16 fractional bits,
// texel = (coord16 * size) >> 16 (equals floor(u*size) except within 2^-16 of a texel boundary).
TH_D int nearest_texel_fx16(float u, unsigned n)
{
    float c = __builtin_amdgcn_fmed3f(u, 0.0f, 65535.0f / 65536.0f);
    unsigned fx = (unsigned)(c * 65536.0f);
    return (int)((fx * n) >> 16);
}

// gray value of frame texel (tx, ty): texture2D on RGBA8 then grayScale() (src/utils/gray-scale.glsl:2).
// UNORM8 -> float as (c*257) * (1/65535): what the captured reference run did; equals c/255 within 1 ulp.
TH_D float gray_texel(const uchar4 *img, int w, int tx, int ty)
{
    uchar4 t = img[ty * w + tx];
    const float k = 1.0f / 65535.0f;
    float r = ((float)t.x * 257.0f) * k, g = ((float)t.y * 257.0f) * k, b = ((float)t.z * 257.0f) * k;
    return r * 0.3f + g * 0.59f + b * 0.11f;
}

// shader arithmetic after the ten taps (src/optical-flow/index.frag:69-80) + the blend
TH_D void optical_flow_finish(const OpticalFlowParams &p, int idx, float gx, float gy, float diff)
{
    const th_optical_flow_uniforms &u = p.u;
    float gm = __builtin_sqrtf((gx * gx) + (gy * gy) + u.lambda);
    float vx = (diff * (gx / gm)) * u.speed, vy = (diff * (gy / gm)) * u.speed;
    // bezier(vec3(0,0,1), t) (src/utils/bezier.glsl:9-13), literal operation order
    float t = __builtin_sqrtf(vx * vx + vy * vy) / u.speedLimit;
    float ut = 1.0f - t;
    float bz = (0.0f * ut + 0.0f * t) * ut + (0.0f * ut + 1.0f * t) * t;
    float fx = bz * vx, fy = bz * vy;
    // flow(vel, speedLimit): vec4(vel, time, min(length(vel)/speedLimit, 1))  src/flow/apply/state.glsl:5-16
    float a = __builtin_fminf(__builtin_sqrtf(fx * fx + fy * fy) / u.speedLimit, 1.0f);
    // gl.blendFunc(SRC_ALPHA, ONE_MINUS_SRC_ALPHA) on the float target
    float4 d = p.flow[idx];
    float ia = 1.0f - a;
    p.flow[idx] = make_float4(fx * a + d.x * ia, fy * a + d.y * ia, u.time * a + d.z * ia, a * a + d.w * ia);
}

// varying uv -> frame uv of output texel coordinate c along one axis (src/optical-flow/index.frag:56):
// NDC of the pixel centre as the rasteriser interpolates it, (c+0.5)*A - 1 with A = fl(2/size)
// (DESIGN.md "optical flow"), then posToUV(uv*scaleUV/viewSize).
TH_D float frame_uv(int c, float grad, float scale, float view)
{
    float u = ((float)c + 0.5f) * grad - 1.0f;
    float q = u * scale / view;
    return (q + 1.0f) * 0.5f;              // posToUV, exactly (1*(v+1))/2
}

constexpr int kOfTileW = 32, kOfTileH = 8;          // output texels per workgroup (32 x 8 lanes)
constexpr int kOfLdsTexels = 3072;                   // staged frame texels per frame (12 KiB per frame)

// One workgroup = a 32 x 8 tile of flow texels.  When the ten taps of the whole tile fall inside a
// small box of frame texels (offset of a few texels: the reference default `offset` scaled to
// texels, SURVEY.md C3 offset = 1/1920), both frames' box is converted to gray ONCE and staged in
// LDS - each staged texel then serves ~5 taps of neighbouring lanes and the UNORM8->gray arithmetic
// runs once per frame texel instead of once per tap.  Otherwise (the demo's offset = 0.1 UV = 192
// texels at 1080p: no reuse inside a tile) every tap is a coalesced global load served by L2.
// Tap texel indices are computed identically on both paths, so the result is bit-identical.
__global__ __launch_bounds__(256) void optical_flow_kernel(const OpticalFlowParams p)
{
    __shared__ float lds_view[kOfLdsTexels], lds_last[kOfLdsTexels];
    __shared__ int box[4];                               // x0, x1, y0, y1 (frame texels, inclusive)
    const th_optical_flow_uniforms &u = p.u;
    const int lx = threadIdx.x & (kOfTileW - 1), ly = threadIdx.x >> 5;
    const int tiles_x = (p.out_w + kOfTileW - 1) / kOfTileW;
    const int bx = (blockIdx.x % tiles_x) * kOfTileW, by = (blockIdx.x / tiles_x) * kOfTileH;
    const int x = bx + lx, y = by + ly;
    const float o = u.offset;

    // box of the tile's taps: the tap index is monotonic in the output coordinate for a fixed offset,
    // so the extremes are reached at the tile's first / last column (row) with offsets -o, 0, +o
    if (threadIdx.x < 12) {
        const int axis = threadIdx.x / 6, rest = threadIdx.x % 6, edge = rest / 3, k = rest % 3;
        const float off = k == 0 ? -o : (k == 1 ? 0.0f : o);
        int c = axis == 0 ? (edge ? min(bx + kOfTileW, p.out_w) - 1 : bx) : (edge ? min(by + kOfTileH, p.out_h) - 1 : by);
        float s = axis == 0 ? frame_uv(c, p.grad_x, u.scaleUV[0], u.viewSize[0]) : frame_uv(c, p.grad_y, u.scaleUV[1], u.viewSize[1]);
        int t = nearest_texel_fx16(s + off, (unsigned)(axis == 0 ? p.fr_w : p.fr_h));
        // (threads 0..11 are one wave: its LDS operations execute in program order)
        if (threadIdx.x == 0) { box[0] = t; box[1] = t; }
        if (threadIdx.x == 6) { box[2] = t; box[3] = t; }
        atomicMin(&box[2 * axis], t);
        atomicMax(&box[2 * axis + 1], t);
    }
    __syncthreads();
    const int x0 = box[0], y0 = box[2], bw = box[1] - box[0] + 1, bh = box[3] - box[2] + 1;
    const bool tiled = bw * bh <= kOfLdsTexels;          // workgroup-uniform
    if (tiled) {
        for (int t = threadIdx.x; t < bw * bh; t += 256) {
            int ty = t / bw, tx = t - ty * bw;
            lds_view[t] = gray_texel(p.view, p.fr_w, x0 + tx, y0 + ty);
            lds_last[t] = gray_texel(p.last, p.fr_w, x0 + tx, y0 + ty);
        }
    }
    __syncthreads();
    if (x >= p.out_w || y >= p.out_h) return;

    const float sx = frame_uv(x, p.grad_x, u.scaleUV[0], u.viewSize[0]);
    const float sy = frame_uv(y, p.grad_y, u.scaleUV[1], u.viewSize[1]);
    const int txm = nearest_texel_fx16(sx - o, (unsigned)p.fr_w), txc = nearest_texel_fx16(sx, (unsigned)p.fr_w),
              txp = nearest_texel_fx16(sx + o, (unsigned)p.fr_w);
    const int tym = nearest_texel_fx16(sy - o, (unsigned)p.fr_h), tyc = nearest_texel_fx16(sy, (unsigned)p.fr_h),
              typ = nearest_texel_fx16(sy + o, (unsigned)p.fr_h);
    float gx, gy, diff;
    if (tiled) {
#define V(tx, ty) lds_view[((ty) - y0) * bw + ((tx) - x0)]
#define L(tx, ty) lds_last[((ty) - y0) * bw + ((tx) - x0)]
        gx = (V(txp, tyc) - V(txm, tyc)) + (L(txp, tyc) - L(txm, tyc));          // :63-64
        gy = (V(txc, typ) - V(txc, tym)) + (L(txc, typ) - L(txc, tym));          // :66-67
        diff = V(txc, tyc) - L(txc, tyc);                                         // :72
#undef V
#undef L
    } else {
#define V(tx, ty) gray_texel(p.view, p.fr_w, tx, ty)
#define L(tx, ty) gray_texel(p.last, p.fr_w, tx, ty)
        gx = (V(txp, tyc) - V(txm, tyc)) + (L(txp, tyc) - L(txm, tyc));
        gy = (V(txc, typ) - V(txc, tym)) + (L(txc, typ) - L(txc, tym));
        diff = V(txc, tyc) - L(txc, tyc);
#undef V
#undef L
    }
    optical_flow_finish(p, y * p.out_w + x, gx, gy, diff);
}

void launch_optical_flow(const OpticalFlowParams &p, hipStream_t s)
{
    if (p.out_w <= 0 || p.out_h <= 0) return;
    const int tiles = ((p.out_w + kOfTileW - 1) / kOfTileW) * ((p.out_h + kOfTileH - 1) / kOfTileH);
    hipLaunchKernelGGL(optical_flow_kernel, dim3(tiles), dim3(256), 0, s, p);
}

// ---------------------------------------------------------------------------
// Respawn passes (src/spawn/ball/index.frag, src/spawn/pixels/frag/*).  The GLSL hash
// fract(sin(x)*43758.5453) (glsl-random 0.0.5) and angleToVec() amplify the last bits of
// the platform's sin/cos; this build pins them to one fixed fp64 sequence (quadrant
// reduction, Taylor polynomials in fma form, one rounding to fp32) so that host and device
// agree bit-for-bit and the value is within one fp32 rounding of the true one.
// ---------------------------------------------------------------------------
TH_D void sincos_pinned(float xf, float &s_out, float &c_out)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    double x = (double)xf;
    double k = __builtin_rint(x * TWO_OVER_PI);
    double r = __builtin_fma(-k, PIO2_HI, x);
    r = __builtin_fma(-k, PIO2_LO, r);
    double r2 = r * r;
    double ps = -1.0 / 1307674368000.0;
    ps = __builtin_fma(ps, r2, 1.0 / 6227020800.0);
    ps = __builtin_fma(ps, r2, -1.0 / 39916800.0);
    ps = __builtin_fma(ps, r2, 1.0 / 362880.0);
    ps = __builtin_fma(ps, r2, -1.0 / 5040.0);
    ps = __builtin_fma(ps, r2, 1.0 / 120.0);
    ps = __builtin_fma(ps, r2, -1.0 / 6.0);
    double sr = __builtin_fma(ps * r2, r, r);
    double pc = 1.0 / 20922789888000.0;
    pc = __builtin_fma(pc, r2, -1.0 / 87178291200.0);
    pc = __builtin_fma(pc, r2, 1.0 / 479001600.0);
    pc = __builtin_fma(pc, r2, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, r2, 1.0 / 40320.0);
    pc = __builtin_fma(pc, r2, -1.0 / 720.0);
    pc = __builtin_fma(pc, r2, 1.0 / 24.0);
    pc = __builtin_fma(pc, r2, -0.5);
    double cr = __builtin_fma(pc, r2, 1.0);
    long long q = (long long)k & 3;
    double s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    double c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
    s_out = (float)s;
    c_out = (float)c;
}

TH_D float mod_glsl(float x, float y) { return x - y * th_floor(x / y); }
TH_D float fract_glsl(float x) { return x - th_floor(x); }

// glsl-random 0.0.5
TH_D float random_glsl(float cox, float coy)
{
    float dt = cox * 12.9898f + coy * 78.233f;
    float sn = mod_glsl(dt, 3.14f);
    float s, c;
    sincos_pinned(sn, s, c);
    return fract_glsl(s * 43758.5453f);
}

// src/spawn/ball/index.frag:11-19
__global__ __launch_bounds__(256) void spawn_ball_kernel(const SpawnBallParams p)
{
    const float tau = 6.28318530717958647692f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < p.count; idx += gridDim.x * 256u) {
        uint32_t y = idx / p.width, x = idx - y * p.width;
        float fx = (float)x + 0.5f, fy = (float)(y + p.row0) + 0.5f;
        float r0 = random_glsl(fx * 1.7654f + 2.3675f, fy * 1.7654f + 2.3675f);
        float r1 = random_glsl(fx * 1.23494f + 0.36434f, fy * 1.23494f + 0.36434f);
        float r2 = random_glsl(fx * 0.327789f + 3.498787f, fy * 0.327789f + 3.498787f);
        float r3 = random_glsl(fx * 9.0374f + 0.2773f, fy * 9.0374f + 0.2773f);
        float s0, c0, s1, c1;
        sincos_pinned(r0 * tau, s0, c0);
        sincos_pinned(r2 * tau, s1, c1);
        p.out[idx] = make_float4(c0 * r1 * p.u.radius, s0 * r1 * p.u.radius, c1 * r3 * p.u.speed, s1 * r3 * p.u.speed);
    }
}

void launch_spawn_ball(const SpawnBallParams &p, hipStream_t s)
{
    if (p.count) hipLaunchKernelGGL(spawn_ball_kernel, dim3(grid_for(p.count, 8)), dim3(256), 0, s, p);
}

TH_D int nearest_texel_f32(float u, float nf, float nm1)
{
    return (int)__builtin_amdgcn_fmed3f(th_floor(u * nf), 0.0f, nm1);
}

// filter/pass/vignette.glsl:9-11 with curve (0.1,1,1), mid 0.5, limit 0.6 (spawn/pixels/vignette-head.glsl:4-6)
TH_D float spawn_vignette(float u, float v)
{
    float dx = u - 0.5f, dy = v - 0.5f;
    float amt = __builtin_fminf(1.0f - (__builtin_sqrtf(dx * dx + dy * dy) / 0.6f), 1.0f);
    float ut = 1.0f - amt;
    float bz = (0.1f * ut + 1.0f * amt) * ut + (1.0f * ut + 1.0f * amt) * amt;
    return __builtin_fmaxf(0.0f, bz);
}

// spawn/pixels/apply/color.glsl:13-17 over the vignette pass; rgb2hsv = libs/glsl-hsv/rgb-hsv.glsl:4-11
TH_D float4 spawn_apply_color(float4 t, float vg, float time, float px, float py)
{
    const float r = t.x * vg, g = t.y * vg, b = t.z * vg, a = t.w * vg;
    float p0, p1, p2, p3;
    if (g < b) { p0 = b; p1 = g; p2 = -1.0f; p3 = 2.0f / 3.0f; } else { p0 = g; p1 = b; p2 = 0.0f; p3 = -1.0f / 3.0f; }
    float q0, q1, q2, q3;
    if (r < p0) { q0 = p0; q1 = p1; q2 = p3; q3 = r; } else { q0 = r; q1 = p1; q2 = p2; q3 = p0; }
    const float e = 1.0e-10f;
    const float d = q0 - __builtin_fminf(q3, q1);
    const float h = __builtin_fabsf(q2 + (q3 - q1) / (6.0f * d + e)), s = d / (q0 + e), v = q0;
    float sn, cs;
    sincos_pinned((h + (time * 0.00003f)) * 6.28318530717958647692f, sn, cs);
    return make_float4(px, py, ((cs * s) * v) * a, ((sn * s) * v) * a);
}

// spawnToPos (src/spawn/pixels/frag/head.frag:28-34): jitter, uvToPos, flipUV*spawnSize, mat3 transform
TH_D void spawn_to_pos(const th_spawn_sample_uniforms &u, float su, float sv, float tt, float &px, float &py)
{
    float ra = random_glsl(su - 1.2345f + tt, sv - 1.2345f + tt);
    float rb = random_glsl(su + 1.2345f + tt, sv + 1.2345f + tt);
    float ox = (-u.jitter[0]) * (1.0f - ra) + u.jitter[0] * ra;
    float oy = (-u.jitter[1]) * (1.0f - rb) + u.jitter[1] * rb;
    float qx = -1.0f + (2.0f * ((su + ox) - 0.0f)) / 1.0f;
    float qy = -1.0f + (2.0f * ((sv + oy) - 0.0f)) / 1.0f;
    qx = qx * 1.0f * u.spawnSize[0];
    qy = qy * -1.0f * u.spawnSize[1];
    const float *m = u.spawnMatrix;
    px = m[0] * qx + m[3] * qy + m[6] * 1.0f;
    py = m[1] * qx + m[4] * qy + m[7] * 1.0f;
}

// src/spawn/pixels/index.frag = frag/direct-main.frag:10-21: uv = (gl_FragCoord.xy/dataRes)*(geomRes/dataRes),
// geomRes = [w, 2h] (src/index.js:195-197)
__global__ __launch_bounds__(256) void spawn_direct_kernel(const SpawnSampleParams p)
{
    const th_spawn_sample_uniforms &u = p.u;
    const float dwf = (float)p.dw, dhf = (float)p.dh, dwm1 = (float)(p.dw - 1), dhm1 = (float)(p.dh - 1);
    unsigned long long took = 0;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < p.count; idx += gridDim.x * 256u) {
        uint32_t y = idx / p.width, x = idx - y * p.width;
        float uvx = (((float)x + 0.5f) / p.wf) * (p.wf / p.wf);
        float uvy = (((float)(y + p.row0) + 0.5f) / p.hf) * ((2.0f * p.hf) / p.hf);
        float px, py;
        spawn_to_pos(u, uvx, uvy, u.time * 0.001f, px, py);
        float4 t = p.data[nearest_texel_f32(uvy, dhf, dhm1) * p.dw + nearest_texel_f32(uvx, dwf, dwm1)];
        float4 st = spawn_apply_color(t, spawn_vignette(uvx, uvy), u.time, px, py);
        p.out[idx] = make_float4(st.x, st.y, st.z * u.speed, st.w * u.speed);
        ++took;
    }
    took += __shfl_xor(took, 32); took += __shfl_xor(took, 16); took += __shfl_xor(took, 8);
    took += __shfl_xor(took, 4); took += __shfl_xor(took, 2); took += __shfl_xor(took, 1);
    if ((threadIdx.x & 63u) == 0 && took) atomicAdd(p.accepted, took);
}

void launch_spawn_direct(const SpawnSampleParams &p, hipStream_t s)
{
    if (p.count) hipLaunchKernelGGL(spawn_direct_kernel, dim3(grid_for(p.count, 8)), dim3(256), 0, s, p);
}

// src/spawn/pixels/frag/best-sample-main.frag:21-46 with head.frag:28-34
__global__ __launch_bounds__(256) void spawn_sample_kernel(const SpawnSampleParams p)
{
    const th_spawn_sample_uniforms &u = p.u;
    const float dwf = (float)p.dw, dhf = (float)p.dh, dwm1 = (float)(p.dw - 1), dhm1 = (float)(p.dh - 1);
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < p.count; idx += gridDim.x * 256u) {
        uint32_t y = idx / p.width, x = idx - y * p.width;
        float uvx = ((float)x + 0.5f) / p.wf, uvy = ((float)(y + p.row0) + 0.5f) / p.hf;
        float4 st = p.particles[idx];
        float tt = u.time * 0.001f;
        float add = 1.2345f + tt;
        float b0 = st.x + uvx + add, b1 = st.y + uvy + add, b2 = st.z + uvx + add, b3 = st.w + uvy + add;
        bool took = false;
        for (int n = 0; n < u.samples; ++n) {
            float fn = (float)n;
            float su = mod_glsl(random_glsl(b0 + fn, b1 + fn), 1.0f);
            float sv = mod_glsl(random_glsl(b2 + fn, b3 + fn), 1.0f);
            float px, py;
            spawn_to_pos(u, su, sv, tt, px, py);
            float4 t = p.data[nearest_texel_f32(sv, dhf, dhm1) * p.dw + nearest_texel_f32(su, dwf, dwm1)];
            float4 other;
            if (u.apply == 3) {            // bright-sample.frag -> apply/brightest.glsl:11-15 (GeometrySpawner)
                float sc = t.x * t.z + t.y * t.w;
                float ang = mod_glsl(random_glsl(su * sc, sv * sc), 1.0f) * 6.28318530717958647692f;
                float sn, cs;
                sincos_pinned(ang, sn, cs);
                float lum = (t.x * 0.299f + t.y * 0.587f) + t.z * 0.114f;
                other = make_float4(px, py, (cs * lum) * t.w, (sn * lum) * t.w);
            } else if (u.apply == 2) {     // best-sample.frag: colour apply over the vignette pass
                other = spawn_apply_color(t, spawn_vignette(su, sv), u.time, px, py);
            } else if (u.apply == 0) {     // apply/flow.glsl: vec4(pos, getFlow(pixel, time, decay))
                float k = __builtin_fmaxf(0.0f, 1.0f - ((u.time - t.z) * u.flowDecay));
                other = make_float4(px, py, t.x * k, t.y * k);
            } else {                       // data-sample: identity after the vignette pass
                float vg = spawn_vignette(su, sv);
                other = make_float4(t.x * vg, t.y * vg, t.z * vg, t.w * vg);
            }
            float4 cand = make_float4(other.x, other.y, other.z * u.speed, other.w * u.speed);
            float tc = st.z * st.z + st.w * st.w, tn = cand.z * cand.z + cand.w * cand.w;
            if (!(tc > u.bias * tn)) { st = cand; took = true; }
        }
        p.out[idx] = st;
        const unsigned long long wave = __ballot(took);                 // statistics only: one atomic per wave
        if (wave != 0ull && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(wave))
            atomicAdd(p.accepted, (unsigned long long)__builtin_popcountll(wave));
    }
}

void launch_spawn_sample(const SpawnSampleParams &p, hipStream_t s)
{
    if (p.count) hipLaunchKernelGGL(spawn_sample_kernel, dim3(grid_for(p.count, 8)), dim3(256), 0, s, p);
}

// ---------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_kernel(float4 *dst, float4 v, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

void launch_fill(float4 *dst, float4 value, size_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, dst, value, n);
}

// *flag |= 1 when any component of src is NaN or +-inf
__global__ __launch_bounds__(256) void finite_check_kernel(const float4 *src, size_t n, unsigned int *flag)
{
    size_t stride = (size_t)gridDim.x * 256;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float4 v = src[i];
        bad |= !(__builtin_fabsf(v.x) <= 3.4028234e38f) || !(__builtin_fabsf(v.y) <= 3.4028234e38f) ||
               !(__builtin_fabsf(v.z) <= 3.4028234e38f) || !(__builtin_fabsf(v.w) <= 3.4028234e38f);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

void launch_finite_check(const float4 *src, size_t n, unsigned int *flag, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(finite_check_kernel, dim3(grid_for(n, 4)), dim3(256), 0, s, src, n, flag);
}

// ---------------------------------------------------------------------------
// statistics of one state buffer: per-block partials (plain stores, fixed
// order => reproducible) then a one-block fold.  Not on the hot path.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_kernel(const float4 *st, size_t n, float limit, StatsPartial *part)
{
    __shared__ StatsPartial sh[4];
    size_t stride = (size_t)gridDim.x * 256;
    unsigned long long live = 0, nan = 0, capped = 0;
    double sum = 0.0, mx = 0.0;
    const float cap = limit * (1.0f - 9.5367431640625e-07f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float4 v = st[i];
        bool is_live = v.x != kInert || v.y != kInert;
        bool is_nan = (v.x != v.x) || (v.y != v.y) || (v.z != v.z) || (v.w != v.w);
        float sp = __builtin_sqrtf(v.z * v.z + v.w * v.w);
        bool fin = is_live && !is_nan && sp <= 3.4028234e38f;
        live += is_live; nan += is_nan;
        capped += fin && sp >= cap;
        if (fin) { sum += sp; mx = sp > mx ? sp : mx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        live += __shfl_xor(live, o); nan += __shfl_xor(nan, o); capped += __shfl_xor(capped, o);
        sum += __shfl_xor(sum, o);
        double om = __shfl_xor(mx, o); mx = om > mx ? om : mx;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = StatsPartial{live, nan, capped, sum, mx};
    __syncthreads();
    if (threadIdx.x == 0) {
        StatsPartial a = sh[0];
        for (int w = 1; w < 4; ++w) {
            a.live += sh[w].live; a.nan += sh[w].nan; a.capped += sh[w].capped;
            a.sum_speed += sh[w].sum_speed; a.max_speed = sh[w].max_speed > a.max_speed ? sh[w].max_speed : a.max_speed;
        }
        part[blockIdx.x] = a;
    }
}

__global__ void counter_add_kernel(unsigned long long *counter, unsigned long long n) { *counter += n; }

void launch_counter_add(unsigned long long *counter, unsigned long long n, hipStream_t s)
{
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, s, counter, n);
}

__global__ __launch_bounds__(256) void stats_fold_kernel(const StatsPartial *part, int nparts, size_t n,
                                                          const unsigned long long *respawned, th_counters *out)
{
    __shared__ StatsPartial sh[4];
    unsigned long long live = 0, nan = 0, capped = 0;
    double sum = 0.0, mx = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) {      // fixed assignment => reproducible sums
        live += part[k].live; nan += part[k].nan; capped += part[k].capped;
        sum += part[k].sum_speed;
        mx = part[k].max_speed > mx ? part[k].max_speed : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        live += __shfl_xor(live, o); nan += __shfl_xor(nan, o); capped += __shfl_xor(capped, o);
        sum += __shfl_xor(sum, o);
        double om = __shfl_xor(mx, o); mx = om > mx ? om : mx;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = StatsPartial{live, nan, capped, sum, mx};
    __syncthreads();
    if (threadIdx.x == 0) {
        th_counters c{};
        c.particles = n;
        c.respawned = *respawned;
        for (int w = 0; w < 4; ++w) {
            c.live += sh[w].live; c.nan += sh[w].nan; c.capped += sh[w].capped;
            c.sum_speed += sh[w].sum_speed;
            c.max_speed = sh[w].max_speed > c.max_speed ? sh[w].max_speed : c.max_speed;
        }
        *out = c;
    }
}

// 256 partials -> 1, per workgroup (the first level of launch_stats_fold).  (Round 6 folded in ONE launch - every workgroup its
// 256, the one that finishes last the rest, a counter deciding which: at C3 the counter's 1024 device-scope adds on one word, served
// one after the other, made the fold 29 us instead of 11; at a band of 2 M particles 10.1 against 9.0.  Two short launches it is.)
__global__ __launch_bounds__(256) void stats_fold_parts_kernel(const StatsPartial *part, uint32_t nparts, StatsPartial *out)
{
    __shared__ StatsPartial sh[4];
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    StatsPartial a = k < nparts ? part[k] : StatsPartial{0, 0, 0, 0.0, 0.0};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a.live += __shfl_xor(a.live, o); a.nan += __shfl_xor(a.nan, o); a.capped += __shfl_xor(a.capped, o);
        a.sum_speed += __shfl_xor(a.sum_speed, o);
        const double om = __shfl_xor(a.max_speed, o); a.max_speed = om > a.max_speed ? om : a.max_speed;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        StatsPartial t = sh[0];
        for (int w = 1; w < 4; ++w) {
            t.live += sh[w].live; t.nan += sh[w].nan; t.capped += sh[w].capped;
            t.sum_speed += sh[w].sum_speed; t.max_speed = sh[w].max_speed > t.max_speed ? sh[w].max_speed : t.max_speed;
        }
        out[blockIdx.x] = t;
    }
}

uint32_t fused_stats_parts(uint32_t count, bool sorted) { return 4u * (uint32_t)fused_grid(count, sorted); }       // one per wave

void launch_stats_fold(const StatsPartial *parts, uint32_t nparts, StatsPartial *scratch, size_t n, const unsigned long long *respawned,
                       th_counters *out, hipStream_t s)
{
    // (scratch holds ceil(nparts / 256) partials; a state of up to 2^31 texels leaves <= 2^23 + 8 partials: two levels, then the fold)
    while (nparts > 4096u) {
        const uint32_t blocks = (nparts + 255u) / 256u;
        hipLaunchKernelGGL(stats_fold_parts_kernel, dim3(blocks), dim3(256), 0, s, parts, nparts, scratch);
        parts = scratch; scratch += blocks; nparts = blocks;
    }
    hipLaunchKernelGGL(stats_fold_kernel, dim3(1), dim3(256), 0, s, parts, (int)nparts, n, respawned, out);
}

void launch_stats(const float4 *state, size_t n, float speed_limit, StatsPartial *partials,
                  const unsigned long long *respawned, th_counters *out, hipStream_t s)
{
    int grid = grid_for(n, 4);
    if (grid > kStatsBlocks) grid = kStatsBlocks;
    hipLaunchKernelGGL(stats_kernel, dim3(grid), dim3(256), 0, s, state, n, speed_limit, partials);
    hipLaunchKernelGGL(stats_fold_kernel, dim3(1), dim3(256), 0, s, partials, grid, n, respawned, out);
}

}  // namespace th

