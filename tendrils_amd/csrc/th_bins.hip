// th_bins.hip - the binned draw() pipeline: Tendrils.draw()'s particle lines (src/index.js:278-337) blended into the flow
// field and / or the view buffer from particles held in ANY slot order - in particular the tile-sorted order the
// integrator steps over (th_kernels.hip "Tile-sorted slot order"), so that the reference's frame loop - step(); draw()
// (src/demo.main.js:1082) - never has to return the state to texel order.
//
// Same lines, same rasteriser, same varyings, same blend arithmetic as th_deposit.hip (th_raster.hpp); what differs is how
// GL's primitive order is reconstructed.  The stream-ordered pipeline produces the fragments in stream order and sorts
// them stably by texel with three global radix passes + a gather.  Here the order is restored where the fragments
// meet - inside one 16 x 16-texel bin of the target, in LDS:
//   1. bins_raster_kernel: one thread per SLOT (coalesced state reads whatever the order): rasterise once, keep the
//      covered texels of a line of <= 8 fragments in its record, count the fragments of every bin (per workgroup in an
//      LDS table, one global atomic per workgroup and bin)
//   2. bins_scan_kernel: exclusive scan over the bins -> every bin's range of the fragment array
//   3. bins_emit_kernel: per slot, the varyings at the recorded texels; a workgroup reserves its share of every bin it
//      meets with one atomic and writes (texel, stream index) keys + varyings there - in whatever order: the
//      arrival order inside a bin is not defined
//   4. bins_blend_kernel: one workgroup per bin: its fragments grouped by texel (LDS counting sort), every texel's
//      run ordered by the stream index of its line (short runs: rank by counting; long ones: bitonic sort) and blended
//      in that order by the texel's thread - dst = src*a + dst*(1-a), fragment after fragment, GL's order and arithmetic.
//      Bins of more than kBinCap fragments go through LDS in groups of texels; a single texel of more than kBinCap
//      fragments in windows of its stream indices.
// The stream index of a line is a pure function of its particle id, so the result is the stream-ordered pipeline's, and
// the restatement's, bit for bit, whatever the slot order and whatever the atomics did.
#include "th_kernels.hpp"
#include "th_raster.hpp"

namespace th {
namespace {

constexpr uint32_t kBinSide = 1u << kBinShift, kBinTexels = kBinSide * kBinSide;      // 256 texels = one per thread
constexpr uint32_t kBinCap = 4096;           // fragments ordered in LDS at a time
constexpr uint32_t kRankMaxRun = 64;         // runs up to this length are ordered by counting, longer ones by the bitonic network
constexpr uint32_t kOwnRun = 64;             // runs up to this length are blended by their texel's thread alone

TH_D uint32_t bin_of(const DepositParams &p, uint32_t x, uint32_t y) { return (y >> kBinShift) * p.bins_x + (x >> kBinShift); }

// ---- a workgroup's table of the bins it meets (open addressing in LDS) -------------------------------------------
constexpr uint32_t kWgBinsLog2 = 9, kWgBins = 1u << kWgBinsLog2, kWgBinEmpty = 0xffffffffu, kWgProbes = 48;
struct WgBins { uint32_t key[kWgBins], count[kWgBins], base[kWgBins]; };

TH_D void wgbins_clear(WgBins &t)
{
    for (uint32_t k = threadIdx.x; k < kWgBins; k += 256u) { t.key[k] = kWgBinEmpty; t.count[k] = 0u; t.base[k] = 0u; }
}
// table slot of `bin` (claimed when absent); -1: no room near its hash (the caller goes to the global counter)
TH_D int wgbins_slot(WgBins &t, uint32_t bin)
{
    uint32_t h = (bin * 2654435761u) >> (32 - kWgBinsLog2);
    for (uint32_t n = 0; n < kWgProbes; ++n, h = (h + 1u) & (kWgBins - 1u)) {
        const uint32_t k = *(volatile uint32_t *)&t.key[h];
        if (k == bin) return (int)h;
        if (k == kWgBinEmpty) {
            const uint32_t old = atomicCAS(&t.key[h], kWgBinEmpty, bin);
            if (old == kWgBinEmpty || old == bin) return (int)h;
        }
    }
    return -1;
}
TH_D void wgbins_count(WgBins &t, const DepositParams &p, uint32_t bin)
{
    const int h = wgbins_slot(t, bin);
    if (h >= 0) atomicAdd(&t.count[h], 1u); else atomicAdd(&p.bin_hist[bin], 1u);
}
TH_D void wgbins_flush(const WgBins &t, const DepositParams &p)
{
    for (uint32_t k = threadIdx.x; k < kWgBins; k += 256u)
        if (t.key[k] != kWgBinEmpty && t.count[k]) atomicAdd(&p.bin_hist[t.key[k]], t.count[k]);
}

// the line of slot s: particle id, its texel (col, row), the line itself
TH_D void slot_line(const DepositParams &p, uint32_t s, uint32_t &col, uint32_t &row, DepositLine &L)
{
    const uint32_t pid = p.perm ? p.perm[s] : s;
    row = pid / p.W; col = pid - row * p.W;
    dep_setup(p, col, p.row0 + row, L, s);
}

// a line's fragments counted from its record (lines of up to kRecordTexels fragments)
TH_D void count_record(WgBins &t, const DepositParams &p, const LineRecord &r)
{
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k)
        if (k < r.n) wgbins_count(t, p, bin_of(p, r.r[k] & 0xffffu, r.r[k] >> 16));
}

// pass 1: every slot's line rasterised once (the common case: a small hexagon inside the view, all in registers; the
// rest through the slow list).  Lines of more than kRecordTexels fragments are counted by bins_count_long_kernel.
__global__ __launch_bounds__(256) void bins_raster_kernel(const DepositParams p)
{
    __shared__ WgBins bins;
    wgbins_clear(bins);
    __syncthreads();
    const uint32_t s = blockIdx.x * 256u + threadIdx.x, slots = p.W * p.rows;
    LineRecord r{};
    bool slow = false;
    if (s < slots) {
        uint32_t col, row;
        DepositLine L;
        slot_line(p, s, col, row, L);
        if (L.draws) {
            float cx[6], cy[6];
            const int where = dep_hexagon(p, L, cx, cy);
            if (where == kHexInside) {
                int PX[6], PY[6], ymin, ymax;
                dep_snap_hexagon(p, cx, cy, PX, PY);
                if (dep_hexagon_is_small(PX, PY, ymin, ymax))
                    dep_raster_small_hexagon(p, PX, PY, ymin, ymax, [&](int x, int y) { rec_add(r, x, y); });
                else slow = true;
            } else if (where == kHexClip) slow = true;
        }
        p.count[s] = slow ? kNeedsSlow : r.n;
        if (r.n) rec_store(p, s, r);
        if (r.n <= kRecordTexels) count_record(bins, p, r);
    }
    dep_list_append(p, kListSlow, blockIdx.x, slow, s);
    dep_list_append(p, kListLong, blockIdx.x, r.n > kRecordTexels, s);
    __syncthreads();
    wgbins_flush(bins, p);
}

__global__ __launch_bounds__(256) void bins_raster_slow_kernel(const DepositParams p)
{
    dep_list_work(p, kListSlow, [&](bool have, uint32_t s, uint32_t seg) {
        LineRecord r{};
        if (have) {
            uint32_t col, row;
            DepositLine L;
            slot_line(p, s, col, row, L);
            dep_raster_line(p, L, [&](int x, int y) { rec_add(r, x, y); });
            p.count[s] = r.n;
            if (r.n) rec_store(p, s, r);
            if (r.n <= kRecordTexels) {
#pragma unroll
                for (uint32_t k = 0; k < kRecordTexels; ++k)
                    if (k < r.n) atomicAdd(&p.bin_hist[bin_of(p, r.r[k] & 0xffffu, r.r[k] >> 16)], 1u);
            }
        }
        dep_list_append(p, kListLong, seg, r.n > kRecordTexels, s);
    });
}

// ... and the fragments of the long lines (more than a record holds), rasterised again
__global__ __launch_bounds__(256) void bins_count_long_kernel(const DepositParams p)
{
    dep_list_work(p, kListLong, [&](bool have, uint32_t s, uint32_t) {
        if (!have) return;
        uint32_t col, row;
        DepositLine L;
        slot_line(p, s, col, row, L);
        dep_raster_line(p, L, [&](int x, int y) { atomicAdd(&p.bin_hist[bin_of(p, (uint32_t)x, (uint32_t)y)], 1u); });
    });
}

// pass 2: one workgroup: exclusive scan of the bins' fragment counts (64-bit sums, saturated: a total beyond 2^32 must
// be seen) -> bin_start (nbins + 1), the fill cursors, totals[0] = fragments, totals[2] = the largest bin
__global__ __launch_bounds__(1024) void bins_scan_kernel(const DepositParams p, uint32_t *totals)
{
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t most[1024];
    const uint32_t per = (p.nbins + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per < p.nbins ? threadIdx.x * per : p.nbins, hi = lo + per < p.nbins ? lo + per : p.nbins;
    unsigned long long n = 0;
    uint32_t m = 0;
    for (uint32_t b = lo; b < hi; ++b) { const uint32_t h = p.bin_hist[b]; n += h; m = h > m ? h : m; }
    part[threadIdx.x] = n; most[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        unsigned long long a = 0;
        uint32_t am = 0;
        if (threadIdx.x >= off) { a = part[threadIdx.x - off]; am = most[threadIdx.x - off]; }
        __syncthreads();
        part[threadIdx.x] += a; most[threadIdx.x] = most[threadIdx.x] > am ? most[threadIdx.x] : am;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - n;
    for (uint32_t b = lo; b < hi; ++b) {
        const uint32_t at = (uint32_t)(run > 0xffffffffull ? 0xffffffffull : run);
        p.bin_start[b] = at; p.bin_cursor[b] = at;
        run += p.bin_hist[b];
    }
    if (threadIdx.x == 1023u) {
        const uint32_t total = (uint32_t)(part[1023] > 0xffffffffull ? 0xffffffffull : part[1023]);
        p.bin_start[p.nbins] = total;
        totals[0] = total; totals[2] = most[1023];
    }
}

// one fragment into slot `at` of the bin-major fragment array
TH_D void bins_put(const DepositParams &p, const DepositLine &L, uint32_t id, uint32_t at, int x, int y)
{
    p.frag_keys[at] = ((unsigned long long)(((uint32_t)y << 12) | (uint32_t)x) << 32) | id;
    float t = 0.0f;
    const bool along = dep_param(L, x, y, t);
    if (p.mode == 2) {
        p.colors[2u * (size_t)at] = dep_mix(L.a.c, L.b.c, along, t);
        p.colors[2u * (size_t)at + 1u] = dep_mix(L.a.c2, L.b.c2, along, t);
    } else p.colors[at] = dep_mix(L.a.c, L.b.c, along, t);
}

// pass 3: the fragments of the lines of up to kRecordTexels fragments, from their records, into their bins
__global__ __launch_bounds__(256) void bins_emit_kernel(const DepositParams p)
{
    __shared__ WgBins bins;
    wgbins_clear(bins);
    __syncthreads();
    const uint32_t s = blockIdx.x * 256u + threadIdx.x, slots = p.W * p.rows;
    uint32_t n = s < slots ? p.count[s] : 0u;
    if (n > kRecordTexels) n = 0u;
    uint4 ra = make_uint4(0u, 0u, 0u, 0u), rb = ra;
    if (n) ra = p.record[2u * s];
    if (n > 4u) rb = p.record[2u * s + 1u];
    const uint32_t xy[kRecordTexels] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    uint32_t where[kRecordTexels];          // table slot << 16 | rank inside the workgroup's share of the bin
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) {
        where[k] = 0xffffffffu;
        if (k < n) {
            const int h = wgbins_slot(bins, bin_of(p, xy[k] & 0xffffu, xy[k] >> 16));
            if (h >= 0) where[k] = ((uint32_t)h << 16) | atomicAdd(&bins.count[h], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kWgBins; k += 256u)
        if (bins.key[k] != kWgBinEmpty && bins.count[k]) bins.base[k] = atomicAdd(&p.bin_cursor[bins.key[k]], bins.count[k]);
    __syncthreads();
    if (!n) return;
    uint32_t col, row;
    DepositLine L;
    slot_line(p, s, col, row, L);
    const uint32_t id = col * p.H + p.row0 + row;
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k)
        if (k < n) {
            const uint32_t x = xy[k] & 0xffffu, y = xy[k] >> 16;
            const uint32_t at = where[k] != 0xffffffffu ? bins.base[where[k] >> 16] + (where[k] & 0xffffu)
                                                        : atomicAdd(&p.bin_cursor[bin_of(p, x, y)], 1u);
            bins_put(p, L, id, at, (int)x, (int)y);
        }
}

__global__ __launch_bounds__(256) void bins_emit_long_kernel(const DepositParams p)
{
    dep_list_work(p, kListLong, [&](bool have, uint32_t s, uint32_t) {
        if (!have) return;
        uint32_t col, row;
        DepositLine L;
        slot_line(p, s, col, row, L);
        const uint32_t id = col * p.H + p.row0 + row;
        dep_raster_line(p, L, [&](int x, int y) {
            bins_put(p, L, id, atomicAdd(&p.bin_cursor[bin_of(p, (uint32_t)x, (uint32_t)y)], 1u), x, y);
        });
    });
}

// ---- pass 4: one workgroup per bin -----------------------------------------------------------------------------
// sort key of a fragment while its bin is ordered: texel inside the bin (8 bits) | stream index (32) | position in the bin's
// range of the fragment array (24 bits: where its varying lies)
TH_D uint32_t key_local(unsigned long long k) { return (uint32_t)((k >> 40) & 0xf0u) | (uint32_t)((k >> 32) & 0xfu); }   // (y & 15) << 4 | (x & 15)
TH_D unsigned long long sort_key(unsigned long long k, uint32_t f) { return ((unsigned long long)key_local(k) << 56) | ((k & 0xffffffffull) << 24) | f; }

// the destination texel(s) of one thread, in registers while the runs are blended.  MODE 0: the flow texture, 1: the view
// buffer, 2: both (the fragments carry two varyings side by side)
template <int MODE>
struct BinTexel {
    float4 f;
    uchar4 v;
    TH_D void load(const DepositParams &p, uint32_t texel)
    {
        if constexpr (MODE != 1) f = p.flow[texel];
        if constexpr (MODE != 0) v = p.view[texel];
    }
    TH_D void store(const DepositParams &p, uint32_t texel) const
    {
        if constexpr (MODE != 1) p.flow[texel] = f;
        if constexpr (MODE != 0) p.view[texel] = v;
    }
};
template <int MODE> struct BinSources { BlendSource a, b; };      // (b: the view pass's, MODE 2 only)

template <int MODE>
TH_D void fetch_colors(const DepositParams &p, size_t frag, float4 &c0, float4 &c1)
{
    if constexpr (MODE == 2) { c0 = p.colors[2u * frag]; c1 = p.colors[2u * frag + 1u]; }
    else { c0 = p.colors[frag]; c1 = c0; }
}
template <int MODE>
TH_D void apply_colors(BinTexel<MODE> &d, float4 c0, float4 c1)
{
    if constexpr (MODE == 0) FlowTarget::apply(d.f, FlowTarget::source(c0));
    else if constexpr (MODE == 1) ViewTarget::apply(d.v, ViewTarget::source(c0));
    else { FlowTarget::apply(d.f, FlowTarget::source(c0)); ViewTarget::apply(d.v, ViewTarget::source(c1)); }
}

template <int MODE>
struct BinShared {
    unsigned long long skey[kBinCap];        // the batch being ordered: sort keys, grouped by texel
    uint16_t order[kBinCap];                 // order[j] = index into skey of the j-th fragment in blend order
    uint32_t cnt[kBinTexels], first[kBinTexels + 1u], fill[kBinTexels];
    BlendSource stage_a[256], stage_b[MODE == 2 ? 256 : 1];  // a long run's sources, 256 at a time (b: the view pass's beside the flow pass's)
    uint32_t misc[8];
};

// bitonic network over s.skey[0, P) (P a power of two >= m, the tail padded with ~0); then order = identity
template <int MODE>
TH_D void bin_bitonic(BinShared<MODE> &s, uint32_t m)
{
    uint32_t P = 64;
    while (P < m) P <<= 1;
    for (uint32_t i = m + threadIdx.x; i < P; i += 256u) s.skey[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < (P >> 1); i += 256u) {
                const uint32_t lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;
                const unsigned long long a = s.skey[lo], b = s.skey[hi];
                const bool up = (lo & k) == 0u;
                if ((a > b) == up) { s.skey[lo] = b; s.skey[hi] = a; }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < m; i += 256u) s.order[i] = (uint16_t)i;
    __syncthreads();
}

// a long run - positions [at, at + len) of the blend order - by the whole workgroup: every thread turns one fragment's
// varying into its side of the blend (256 coalesced-as-they-come loads in flight), the texel's thread applies them in order
template <int MODE>
TH_D void bin_blend_long(BinShared<MODE> &s, const DepositParams &p, uint32_t begin, uint32_t at, uint32_t len, uint32_t owner, BinTexel<MODE> &d)
{
    for (uint32_t j0 = 0; j0 < len; j0 += 256u) {
        const uint32_t j = j0 + threadIdx.x;
        if (j < len) {
            const uint32_t src = (uint32_t)(s.skey[s.order[at + j]] & 0xffffffull);
            float4 c0, c1;
            fetch_colors<MODE>(p, (size_t)begin + src, c0, c1);
            if constexpr (MODE == 1) s.stage_a[threadIdx.x] = ViewTarget::source(c0);
            else s.stage_a[threadIdx.x] = FlowTarget::source(c0);
            if constexpr (MODE == 2) s.stage_b[threadIdx.x] = ViewTarget::source(c1);
        }
        __syncthreads();
        if (threadIdx.x == owner) {
            const uint32_t n = len - j0 < 256u ? len - j0 : 256u;
            for (uint32_t q = 0; q < n; ++q) {
                if constexpr (MODE == 1) ViewTarget::apply(d.v, s.stage_a[q]);
                else FlowTarget::apply(d.f, s.stage_a[q]);
                if constexpr (MODE == 2) ViewTarget::apply(d.v, s.stage_b[q]);
            }
        }
        __syncthreads();
    }
}

// order the batch s.skey[0, m) - grouped by texel, run of texel t at [first[t] - first[t0], ...) - and blend it.
// `longest`: the longest run of the batch.
template <int MODE>
TH_D void bin_order_and_blend(BinShared<MODE> &s, const DepositParams &p, uint32_t begin, uint32_t m, uint32_t t0, uint32_t t1, uint32_t longest,
                              BinTexel<MODE> &d, bool &touched)
{
    const uint32_t t = threadIdx.x, base = s.first[t0];
    if (longest <= kRankMaxRun) {
        // rank by counting: a fragment's place in its run = the fragments of the run with a smaller key
        for (uint32_t q = t; q < m; q += 256u) {
            const unsigned long long k = s.skey[q];
            const uint32_t lt = (uint32_t)(k >> 56), r0 = s.first[lt] - base, r1 = s.first[lt + 1u] - base;
            uint32_t rank = 0;
            for (uint32_t j = r0; j < r1; ++j) rank += s.skey[j] < k ? 1u : 0u;
            s.order[r0 + rank] = (uint16_t)q;
        }
        __syncthreads();
    } else bin_bitonic(s, m);
    // every texel's run, by its own thread while the run is short
    const bool mine = t >= t0 && t < t1;
    const uint32_t r0 = mine ? s.first[t] - base : 0u, len = mine ? s.first[t + 1u] - s.first[t] : 0u;
    if (len && len <= kOwnRun) {
        touched = true;
        for (uint32_t j0 = 0; j0 < len; j0 += 4u) {
            float4 c0[4], c1[4];
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                const uint32_t j = j0 + q < len ? j0 + q : len - 1u;
                const uint32_t src = (uint32_t)(s.skey[s.order[r0 + j]] & 0xffffffull);
                fetch_colors<MODE>(p, (size_t)begin + src, c0[q], c1[q]);
            }
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) if (j0 + q < len) apply_colors<MODE>(d, c0[q], c1[q]);
        }
    }
    if (longest > kOwnRun) {
        // the long runs one after the other, by everybody (found by all threads in the same order: workgroup-uniform)
        for (uint32_t lt = t0; lt < t1; ++lt) {
            const uint32_t l = s.first[lt + 1u] - s.first[lt];
            if (l <= kOwnRun) continue;
            if (t == lt) touched = true;
            bin_blend_long<MODE>(s, p, begin, s.first[lt] - base, l, lt, d);
        }
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void bins_blend_kernel(const DepositParams p)
{
    __shared__ BinShared<MODE> s;
    const uint32_t b = blockIdx.x, begin = p.bin_start[b], n = p.bin_start[b + 1u] - begin;
    if (n == 0u) return;
    const uint32_t t = threadIdx.x;
    const unsigned long long *keys = p.frag_keys + begin;
    const uint32_t by = b / p.bins_x, bx = b - by * p.bins_x;
    const uint32_t x = (bx << kBinShift) + (t & (kBinSide - 1u)), y = (by << kBinShift) + (t >> kBinShift);
    const bool inside = x < (uint32_t)p.fw && y < (uint32_t)p.fh;
    const uint32_t texel = inside ? y * (uint32_t)p.fw + x : 0u;
    BinTexel<MODE> d{};
    if (inside) d.load(p, texel);
    bool touched = false;

    // fragments per texel of the bin, and every texel's first position among all n
    s.cnt[t] = 0u;
    __syncthreads();
    for (uint32_t f = t; f < n; f += 256u) atomicAdd(&s.cnt[key_local(keys[f])], 1u);
    __syncthreads();
    {
        const uint32_t mine = s.cnt[t], lane = t & 63u, wave = t >> 6;
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)lane >= o) incl += up; }
        if (lane == 63u) s.misc[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += s.misc[w];
        s.first[t] = before + incl - mine;
        if (t == 255u) s.first[256] = before + incl;
        __syncthreads();
    }

    // batches: as many whole texels as fit kBinCap fragments; a texel that does not fit alone goes in windows of ids
    uint32_t t0 = 0;
    while (t0 < kBinTexels) {
        // (every thread walks the same table: workgroup-uniform without another barrier)
        uint32_t t1 = t0, longest = 0;
        while (t1 < kBinTexels && s.first[t1 + 1u] - s.first[t0] <= kBinCap) { const uint32_t l = s.first[t1 + 1u] - s.first[t1]; longest = l > longest ? l : longest; ++t1; }
        if (t1 > t0) {
            const uint32_t m = s.first[t1] - s.first[t0];
            if (m) {
                if (t >= t0 && t < t1) s.fill[t] = 0u;
                __syncthreads();
                if (t0 == 0u && t1 == kBinTexels) {            // the whole bin at once: no texel test
                    for (uint32_t f = t; f < n; f += 256u) {
                        const unsigned long long k = keys[f];
                        const uint32_t lt = key_local(k);
                        s.skey[s.first[lt] + atomicAdd(&s.fill[lt], 1u)] = sort_key(k, f);
                    }
                } else {
                    for (uint32_t f = t; f < n; f += 256u) {
                        const unsigned long long k = keys[f];
                        const uint32_t lt = key_local(k);
                        if (lt >= t0 && lt < t1) s.skey[s.first[lt] - s.first[t0] + atomicAdd(&s.fill[lt], 1u)] = sort_key(k, f);
                    }
                }
                __syncthreads();
                bin_order_and_blend<MODE>(s, p, begin, m, t0, t1, longest, d, touched);
            }
            t0 = t1;
            continue;
        }
        // texel t0 alone holds more than kBinCap fragments: windows [lo, hi) of its stream indices (distinct inside one
        // texel: a line covers a texel at most once), each small enough for LDS, in rising order
        {
            uint32_t *hist = reinterpret_cast<uint32_t *>(s.order);         // 1024 buckets (order[] is rebuilt per window)
            unsigned long long lo = 0ull;
            uint32_t shift = p.id_bits > 10u ? p.id_bits - 10u : 0u;
            while (true) {
                for (uint32_t k = t; k < 1024u; k += 256u) hist[k] = 0u;
                if (t == 0u) s.misc[4] = 0u;
                __syncthreads();
                for (uint32_t f = t; f < n; f += 256u) {
                    const unsigned long long k = keys[f];
                    const unsigned long long id = k & 0xffffffffull;
                    if (key_local(k) == t0 && id >= lo) {
                        const unsigned long long bkt = (id - lo) >> shift;
                        atomicAdd(&hist[bkt < 1023ull ? (uint32_t)bkt : 1023u], 1u);
                    }
                }
                __syncthreads();
                if (t == 0u) {
                    uint32_t acc = 0, k = 0, rest = 0;
                    while (k < 1024u && acc + hist[k] <= kBinCap) { acc += hist[k]; ++k; }
                    for (uint32_t q = k; q < 1024u; ++q) rest += hist[q];
                    s.misc[5] = k; s.misc[6] = acc; s.misc[7] = rest;
                }
                __syncthreads();
                const uint32_t nb = s.misc[5], m = s.misc[6], rest = s.misc[7];
                __syncthreads();
                if (nb == 0u) { shift = shift > 4u ? shift - 4u : 0u; continue; }     // the first bucket alone is too large: finer buckets
                // (bucket 1023 also holds everything beyond it: taken only together with all the others = the rest of the run)
                const unsigned long long hi = nb == 1024u ? 0x100000000ull : lo + ((unsigned long long)nb << shift);
                for (uint32_t f = t; f < n; f += 256u) {
                    const unsigned long long k = keys[f];
                    const unsigned long long id = k & 0xffffffffull;
                    if (key_local(k) == t0 && id >= lo && id < hi) s.skey[atomicAdd(&s.misc[4], 1u)] = sort_key(k, f);
                }
                __syncthreads();
                bin_bitonic(s, m);
                if (t == t0) touched = true;
                bin_blend_long<MODE>(s, p, begin, 0u, m, t0, d);
                if (rest == 0u) break;
                lo = hi;
            }
        }
        ++t0;
    }
    if (inside && touched) d.store(p, texel);
}

}  // namespace

static uint32_t slot_blocks(const DepositParams &p) { const uint32_t n = (p.W * p.rows + 255u) / 256u; return n ? n : 1u; }

void launch_bins_raster(const DepositParams &p, hipStream_t s)
{
    (void)hipMemsetAsync(p.list_n, 0, (size_t)2 * kDepLists * kDepListStride * sizeof(uint32_t), s);
    (void)hipMemsetAsync(p.bin_hist, 0, (size_t)p.nbins * sizeof(uint32_t), s);
    hipLaunchKernelGGL(bins_raster_kernel, dim3(slot_blocks(p)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_raster_slow_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_count_long_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
}

void launch_bins_scan(const DepositParams &p, uint32_t *totals, hipStream_t s)
{
    hipLaunchKernelGGL(bins_scan_kernel, dim3(1), dim3(1024), 0, s, p, totals);
}

void launch_bins_emit(const DepositParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(bins_emit_kernel, dim3(slot_blocks(p)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_emit_long_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
}

void launch_bins_blend(const DepositParams &p, hipStream_t s)
{
    if (p.mode == 0) hipLaunchKernelGGL(bins_blend_kernel<0>, dim3(p.nbins), dim3(256), 0, s, p);
    else if (p.mode == 1) hipLaunchKernelGGL(bins_blend_kernel<1>, dim3(p.nbins), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(bins_blend_kernel<2>, dim3(p.nbins), dim3(256), 0, s, p);
}

}  // namespace th
