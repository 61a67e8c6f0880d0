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
constexpr uint32_t kRankMaxRun = 256;        // runs up to this length are ordered by counting, longer ones by the bitonic network
constexpr uint32_t kOwnRun = 64;             // runs up to this length are blended by their texel's thread alone

TH_D uint32_t bin_of(const DepositParams &p, uint32_t x, uint32_t y) { return (y >> kBinShift) * p.bins_x + (x >> kBinShift); }

// ---- wave-wide bin arithmetic -------------------------------------------------------------------------------------
// In the tile-sorted order the lines of a wave fall into the same few bins.  The lanes of every distinct bin are found with
// one ballot round per bin, their fragments counted and numbered with ballots and population counts alone - no LDS, no
// barrier - and the leader lane of the bin goes to the global counter once for all of them.
constexpr uint32_t kNoBin = 0xffffffffu;

// The bins' global counters are kept in kBinReplicas copies p.bin_stride words apart, and a slot always uses the copy of
// its 64-slot group: a crowded bin is met by thousands of waves, and device-wide atomics on ONE word are served one after
// the other at the memory side (every XCD has its own L2) - a thousand of them set the time of a whole pass.  The scan
// (bins_replica_kernel) lays the copies of a bin out one after the other inside the bin's range.
TH_D uint32_t *rep_word(uint32_t *base, const DepositParams &p, uint32_t slot, uint32_t bin)
{
    return base + (size_t)((slot >> 6) & (kBinReplicas - 1u)) * p.bin_stride + bin;
}

// raster: the lanes' `c` fragments (<= kRecordTexels each) of `bin` (kNoBin: none) added to the bins' counts
TH_D void wave_bin_count(const DepositParams &p, uint32_t s, uint32_t bin, uint32_t c)
{
    const uint32_t lane = __lane_id();
    unsigned long long todo = __ballot(bin != kNoBin);
    while (todo != 0ull) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bin, (int)leader);
        const bool mine = bin == lb;
        uint32_t total = 0;
#pragma unroll
        for (uint32_t j = 0; j < kRecordTexels; ++j) total += (uint32_t)__builtin_popcountll(__ballot(mine && c > j));
        if (lane == leader) atomicAdd(rep_word(p.rep_hist, p, s, lb), total);
        todo &= ~__ballot(mine);
    }
}

// emit: slot[j] = place in the bin's range of the lane's j-th fragment of `bin` (j < c).  The wave's fragments of one bin
// form ONE contiguous run, numbered j-major: the j-th fragments of all lanes follow each other, so a store of "fragment j"
// writes consecutive places (the arrival order inside a bin is free).  One atomic per bin, all bins' atomics in flight together.
TH_D void wave_bin_slots(const DepositParams &p, uint32_t s, uint32_t bin, uint32_t c, uint32_t (&slot)[kRecordTexels])
{
    const uint32_t lane = __lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(bin != kNoBin);
    uint32_t total = 0, my_leader = lane;
#pragma unroll
    for (uint32_t j = 0; j < kRecordTexels; ++j) slot[j] = 0u;
    while (todo != 0ull) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bin, (int)leader);
        const bool mine = bin == lb;
        uint32_t off = 0;
#pragma unroll
        for (uint32_t j = 0; j < kRecordTexels; ++j) {
            const unsigned long long m = __ballot(mine && c > j);
            if (mine) slot[j] = off + (uint32_t)__builtin_popcountll(m & below);
            off += (uint32_t)__builtin_popcountll(m);
        }
        if (mine) { total = off; my_leader = leader; }
        todo &= ~__ballot(mine);
    }
    uint32_t base = 0;
    if (bin != kNoBin && lane == my_leader) base = atomicAdd(rep_word(p.rep_cursor, p, s, bin), total);
    base = (uint32_t)__shfl((int)base, (int)my_leader);
#pragma unroll
    for (uint32_t j = 0; j < kRecordTexels; ++j) slot[j] += base;
}

// slot s: its particle's texel (col, row) and whether draw() can make a line of it at all (th_api.hip: line_rows)
TH_D bool slot_particle(const DepositParams &p, uint32_t s, uint32_t &col, uint32_t &row)
{
    const uint32_t pid = p.perm ? p.perm[s] : s;
    if ((p.W & (p.W - 1u)) == 0u) { row = pid >> (31 - __builtin_clz(p.W)); col = pid & (p.W - 1u); }       // (uniform branch)
    else { row = pid / p.W; col = pid - row * p.W; }
    const uint32_t g = p.row0 + row;
    return (p.row_draws[g >> 5] >> (g & 31u)) & 1u;
}
// the line of slot s
TH_D void slot_line(const DepositParams &p, uint32_t s, uint32_t &col, uint32_t &row, DepositLine &L)
{
    slot_particle(p, s, col, row);
    dep_setup(p, col, p.row0 + row, L, s);
}
// The slots of a workgroup of the two big passes: in the tile-sorted order the particles whose lines can draw lie apart from
// the others inside every tile (th_kernels.hip: tile_key), so most workgroups meet only one kind: whole waves of lines that
// exist - or nothing to do at all.  Returns false when no slot of the workgroup draws.
// (`which`: a kernel's calls use different words - a second call must not reset what a straggler of the first still reads)
TH_D bool workgroup_draws(bool mine, int which)
{
    __shared__ uint32_t any[2];
    if (threadIdx.x == 0u) any[which] = 0u;
    __syncthreads();
    if (mine && (__lane_id() == (uint32_t)__builtin_ctzll(__ballot(mine)))) any[which] = 1u;
    __syncthreads();
    return any[which] != 0u;
}

// The fragments of a line of up to kRecordTexels fragments, by bin.  A line is about a texel long: nearly always all of its
// fragments fall into one bin (b0), sometimes into two (b1); the table is asked once per bin, not once per fragment.
// Fragments of a third bin (`others`) go to the global counters one by one.
struct LineBins { uint32_t bin[kRecordTexels], b0, b1, c0, c1, others; };
TH_D LineBins line_bins(const DepositParams &p, const uint32_t (&xy)[kRecordTexels], uint32_t n)
{
    LineBins q;
    q.b0 = q.b1 = kNoBin; q.c0 = q.c1 = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) q.bin[k] = k < n ? bin_of(p, xy[k] & 0xffffu, xy[k] >> 16) : kNoBin;
    q.b0 = q.bin[0];
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) {
        const bool in0 = k < n && q.bin[k] == q.b0;
        q.c0 += in0 ? 1u : 0u;
        if (k < n && !in0 && q.b1 == kNoBin) q.b1 = q.bin[k];
    }
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) q.c1 += (k < n && q.bin[k] == q.b1 && q.b1 != kNoBin) ? 1u : 0u;
    q.others = n - q.c0 - q.c1;
    return q;
}

// (called by whole waves; n = 0 for lanes without a line to count)
TH_D void count_record(const DepositParams &p, uint32_t s, const uint32_t (&xy)[kRecordTexels], uint32_t n)
{
    const LineBins q = line_bins(p, xy, n);
    wave_bin_count(p, s, n ? q.b0 : kNoBin, q.c0);
    wave_bin_count(p, s, q.c1 ? q.b1 : kNoBin, q.c1);
    if (q.others) {
#pragma unroll
        for (uint32_t k = 0; k < kRecordTexels; ++k)
            if (k < n && q.bin[k] != q.b0 && q.bin[k] != q.b1) atomicAdd(rep_word(p.rep_hist, p, s, q.bin[k]), 1u);
    }
}

// pass 1: every slot's line rasterised once (the common case: a small hexagon inside the view, all in registers; the
// rest through the slow list).  Lines of more than kRecordTexels fragments are counted by bins_count_long_kernel.
__global__ __launch_bounds__(256) void bins_raster_kernel(const DepositParams p)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x, slots = p.W * p.rows;
    uint32_t col = 0, row = 0;
    const bool can = s < slots && slot_particle(p, s, col, row);
    if (!workgroup_draws(can, 0)) {
        if (s < slots) p.count[s] = 0u;
        if (threadIdx.x == 0u) p.block_flags[blockIdx.x] = 0u;
        return;
    }
    float4 own[2];                                      // (both ends of the line, before anything else)
    if (can) { own[0] = p.cur[s]; own[1] = p.prev[s]; }
    LineRecord r{};
    bool slow = false;
    if (can) {
        DepositLine L;
        dep_setup(p, col, p.row0 + row, L, s, own);
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
        if (r.n) rec_store(p, s, r);
    }
    if (s < slots) p.count[s] = slow ? kNeedsSlow : r.n;        // (one store per wave: whole lines)
    count_record(p, s, r.r, r.n <= kRecordTexels ? r.n : 0u);
    dep_list_append(p, kListSlow, blockIdx.x, slow, s);
    dep_list_append(p, kListLong, blockIdx.x, r.n > kRecordTexels, s);
    const bool busy = workgroup_draws(slow || r.n != 0u, 1);
    if (threadIdx.x == 0u) p.block_flags[blockIdx.x] = busy ? 1u : 0u;     // the emitting pass skips the blocks without fragments
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
                    if (k < r.n) atomicAdd(rep_word(p.rep_hist, p, s, bin_of(p, r.r[k] & 0xffffu, r.r[k] >> 16)), 1u);
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
        dep_raster_line(p, L, [&](int x, int y) { atomicAdd(rep_word(p.rep_hist, p, s, bin_of(p, (uint32_t)x, (uint32_t)y)), 1u); });
    });
}

// pass 2a: every bin's copies added up (-> bin_hist) and laid out one after the other (rep_hist becomes the first place of
// every copy inside its bin's range)
__global__ __launch_bounds__(256) void bins_replica_kernel(const DepositParams p)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= p.nbins) return;
    uint32_t h[kBinReplicas], run = 0;
#pragma unroll
    for (uint32_t r = 0; r < kBinReplicas; ++r) h[r] = p.rep_hist[(size_t)r * p.bin_stride + b];
#pragma unroll
    for (uint32_t r = 0; r < kBinReplicas; ++r) {
        p.rep_hist[(size_t)r * p.bin_stride + b] = run;
        run = run + h[r] < run ? 0xffffffffu : run + h[r];         // (saturating: a count beyond 2^32 must be seen)
    }
    p.bin_hist[b] = run;
}
// ... and after the scan: every copy's fill cursor = first fragment of the bin + first place of the copy
__global__ __launch_bounds__(256) void bins_cursor_kernel(const DepositParams p)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= p.nbins) return;
    const uint32_t start = p.bin_start[b];
#pragma unroll
    for (uint32_t r = 0; r < kBinReplicas; ++r) p.rep_cursor[(size_t)r * p.bin_stride + b] = start + p.rep_hist[(size_t)r * p.bin_stride + b];
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
        const uint32_t at = (uint32_t)(run > 0xffffffffull ? 0xffffffffull : run), h = p.bin_hist[b];
        p.bin_start[b] = at;
        run += h;
        if (h > kBinCap) p.large_bins[atomicAdd(&totals[3], 1u)] = b;       // (in whatever order)
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

// pass 3: the fragments of the lines of up to kRecordTexels fragments, from their records, into their bins.  Wave by
// wave, no LDS, no barrier: the fragments of a wave in one bin are written as one contiguous run (wave_bin_slots); the
// few fragments of a line beyond its first bin (a line crossing a bin's edge) take their places one by one.
__global__ __launch_bounds__(256) void bins_emit_kernel(const DepositParams p)
{
    if (p.block_flags[blockIdx.x] == 0u) return;         // (uniform: no fragments in this block's lines)
    const uint32_t s = blockIdx.x * 256u + threadIdx.x, slots = p.W * p.rows, at = s < slots ? s : slots - 1u;
    // everything a line needs, in flight together (unconditional, clamped)
    uint32_t n = p.count[at];
    const uint4 ra = p.record[2u * at], rb = p.record[2u * at + 1u];
    float4 own[2] = {p.cur[at], p.prev[at]};
    uint32_t col = 0, row = 0;
    slot_particle(p, at, col, row);
    if (s >= slots || n > kRecordTexels) n = 0u;
    const uint32_t xy[kRecordTexels] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    const LineBins q = line_bins(p, xy, n);
    uint32_t slot[kRecordTexels];
    wave_bin_slots(p, s, n ? q.b0 : kNoBin, q.c0, slot);
    if (!n) return;
    DepositLine L;
    dep_setup(p, col, p.row0 + row, L, s, own);
    const uint32_t id = col * p.H + p.row0 + row;
    uint32_t j = 0;
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k)
        if (k < n) {
            const uint32_t x = xy[k] & 0xffffu, y = xy[k] >> 16;
            uint32_t to;
            if (q.bin[k] == q.b0) {
                to = slot[0];
#pragma unroll
                for (uint32_t e = 1; e < kRecordTexels; ++e) to = j == e ? slot[e] : to;      // (selects: the places stay in registers)
                ++j;
            } else to = atomicAdd(rep_word(p.rep_cursor, p, s, q.bin[k]), 1u);
            bins_put(p, L, id, to, (int)x, (int)y);
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
            bins_put(p, L, id, atomicAdd(rep_word(p.rep_cursor, p, s, bin_of(p, (uint32_t)x, (uint32_t)y)), 1u), x, y);
        });
    });
}

// ---- pass 4: one workgroup per bin -----------------------------------------------------------------------------
TH_D uint32_t key_local(unsigned long long k) { return (uint32_t)((k >> 40) & 0xf0u) | (uint32_t)((k >> 32) & 0xfu); }   // (y & 15) << 4 | (x & 15)
// sort key of a fragment while a crowded bin is ordered: texel inside the bin (8 bits) | stream index (32) | position in the
// bin's range of the fragment array (24 bits: where its varying lies)
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

// LDS of a bin's workgroup.  `pool` is used two ways:
//   the common case (a bin of <= kBinCap fragments, no run longer than kRankMaxRun): sid = stream index of every fragment,
//     grouped by texel | osrc = the fragments' positions (where their varyings lie) in blend order        (2 x kBinCap words)
//   crowded bins: skey = 64-bit sort keys of a batch of <= kCrowdCap fragments | order = blend order as indices into skey |
//     hist = the id histogram of a texel too crowded for one batch
constexpr uint32_t kCrowdCap = 2048;
template <int MODE>
struct BinShared {
    uint32_t pool[2u * kBinCap];
    uint32_t cnt[kBinTexels], first[kBinTexels + 1u];        // fragments per texel (then: fill cursors); first fragment of every texel
    BlendSource stage_a[256], stage_b[MODE == 2 ? 256 : 1];  // a long run's sources, 256 at a time (b: the view pass's beside the flow pass's)
    uint32_t misc[8];
    TH_D uint32_t *sid() { return pool; }
    TH_D uint32_t *osrc() { return pool + kBinCap; }
    TH_D unsigned long long *skey() { return reinterpret_cast<unsigned long long *>(pool); }     // kCrowdCap keys = 4096 words
    TH_D uint16_t *order() { return reinterpret_cast<uint16_t *>(pool + 2u * kCrowdCap); }         // kCrowdCap indices = 1024 words
    TH_D uint32_t *hist() { return pool + 2u * kCrowdCap + kCrowdCap; }                             // 1024 buckets
};

// all threads: exclusive scan of s.cnt into s.first (first[256] = total); returns the longest run
template <int MODE>
TH_D uint32_t bin_scan_counts(BinShared<MODE> &s)
{
    const uint32_t t = threadIdx.x, mine = s.cnt[t], lane = t & 63u, wave = t >> 6;
    uint32_t incl = mine, most = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o) incl += up;
        const uint32_t other = __shfl_xor(most, o);
        most = other > most ? other : most;
    }
    if (lane == 63u) s.misc[wave] = incl;
    if (lane == 0u) s.misc[4u + wave] = most;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += s.misc[w];
    s.first[t] = before + incl - mine;
    if (t == 255u) s.first[256] = before + incl;
    uint32_t longest = s.misc[4];
    for (uint32_t w = 1; w < 4u; ++w) longest = s.misc[4u + w] > longest ? s.misc[4u + w] : longest;
    __syncthreads();
    return longest;
}

// a texel's run blended by its own thread: `len` fragments whose positions src_at(0..len-1) gives in blend order; the
// varyings are fetched eight ahead of the dependent blends
template <int MODE, typename SrcAt>
TH_D void bin_blend_own(const DepositParams &p, uint32_t begin, uint32_t len, BinTexel<MODE> &d, SrcAt src_at)
{
    constexpr uint32_t kAhead = MODE == 2 ? 4u : 8u;
    for (uint32_t j0 = 0; j0 < len; j0 += kAhead) {
        uint32_t src[kAhead];
        float4 c0[kAhead], c1[kAhead];
#pragma unroll
        for (uint32_t q = 0; q < kAhead; ++q) src[q] = src_at(j0 + q < len ? j0 + q : len - 1u);
#pragma unroll
        for (uint32_t q = 0; q < kAhead; ++q) fetch_colors<MODE>(p, (size_t)begin + src[q], c0[q], c1[q]);
#pragma unroll
        for (uint32_t q = 0; q < kAhead; ++q) if (j0 + q < len) apply_colors<MODE>(d, c0[q], c1[q]);
    }
}

// a long run by the whole workgroup: every thread turns one fragment's varying into its side of the blend (256 loads in
// flight), the texel's thread applies them in order
template <int MODE, typename SrcAt>
TH_D void bin_blend_long(BinShared<MODE> &s, const DepositParams &p, uint32_t begin, uint32_t len, uint32_t owner, BinTexel<MODE> &d, SrcAt src_at)
{
    for (uint32_t j0 = 0; j0 < len; j0 += 256u) {
        const uint32_t j = j0 + threadIdx.x;
        if (j < len) {
            float4 c0, c1;
            fetch_colors<MODE>(p, (size_t)begin + src_at(j), c0, c1);
            if constexpr (MODE == 1) s.stage_a[threadIdx.x] = ViewTarget::source(c0);
            else s.stage_a[threadIdx.x] = FlowTarget::source(c0);
            if constexpr (MODE == 2) s.stage_b[threadIdx.x] = ViewTarget::source(c1);
        }
        __syncthreads();
        if (threadIdx.x == owner) {
            // (sources read eight ahead of the dependent blends: the chain left is the blend's own multiply and add)
            const uint32_t n = len - j0 < 256u ? len - j0 : 256u;
            uint32_t q = 0;
            for (; q + 8u <= n; q += 8u) {
                BlendSource a[8], b[8];
#pragma unroll
                for (uint32_t e = 0; e < 8u; ++e) { a[e] = s.stage_a[q + e]; if constexpr (MODE == 2) b[e] = s.stage_b[q + e]; }
#pragma unroll
                for (uint32_t e = 0; e < 8u; ++e) {
                    if constexpr (MODE == 1) ViewTarget::apply(d.v, a[e]);
                    else FlowTarget::apply(d.f, a[e]);
                    if constexpr (MODE == 2) ViewTarget::apply(d.v, b[e]);
                }
            }
            for (; q < n; ++q) {
                if constexpr (MODE == 1) ViewTarget::apply(d.v, s.stage_a[q]);
                else FlowTarget::apply(d.f, s.stage_a[q]);
                if constexpr (MODE == 2) ViewTarget::apply(d.v, s.stage_b[q]);
            }
        }
        __syncthreads();
    }
}

// bitonic network over skey[0, P) (P a power of two >= m, the tail padded with ~0); then order = identity
template <int MODE>
TH_D void bin_bitonic(BinShared<MODE> &s, uint32_t m)
{
    unsigned long long *skey = s.skey();
    uint32_t P = 64;
    while (P < m) P <<= 1;
    for (uint32_t i = m + threadIdx.x; i < P; i += 256u) skey[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < (P >> 1); i += 256u) {
                const uint32_t lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;
                const unsigned long long a = skey[lo], b = skey[hi];
                const bool up = (lo & k) == 0u;
                if ((a > b) == up) { skey[lo] = b; skey[hi] = a; }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < m; i += 256u) s.order()[i] = (uint16_t)i;
    __syncthreads();
}

// every key of the bin: body(key, position); a thread's loads of one round of 16 are issued together (unconditional, clamped)
template <typename Body>
TH_D void for_keys(const unsigned long long *keys, uint32_t n, Body body)
{
    constexpr uint32_t kPer = 8;
    for (uint32_t c0 = 0; c0 < n; c0 += kPer * 256u) {
        unsigned long long k[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = c0 + q * 256u + threadIdx.x; k[q] = keys[f < n ? f : n - 1u]; }
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = c0 + q * 256u + threadIdx.x; if (f < n) body(k[q], f); }
    }
}

// One texel's fragments when they do not fit LDS at once: windows [lo, hi) of its stream indices (distinct inside one
// texel: a line covers a texel at most once), each small enough to be sorted in LDS, in rising order; the texel's thread
// (`owner`) carries the destination from window to window.  for_texel(body) hands every fragment's sort key to body -
// stream index in bits 24..55, position of its varying in bits 0..23.
template <int MODE, typename ForTexel>
TH_D void blend_texel_windows(BinShared<MODE> &s, const DepositParams &p, uint32_t begin, uint32_t owner, BinTexel<MODE> &d, ForTexel for_texel)
{
    const uint32_t t = threadIdx.x;
    unsigned long long *skey = s.skey();
    uint16_t *order = s.order();
    uint32_t *hist = s.hist();
    unsigned long long lo = 0ull;
    uint32_t shift = p.id_bits > 10u ? p.id_bits - 10u : 0u;
    while (true) {
        for (uint32_t k = t; k < 1024u; k += 256u) hist[k] = 0u;
        if (t == 0u) s.misc[0] = 0u;
        __syncthreads();
        for_texel([&](unsigned long long k) {
            const unsigned long long id = (k >> 24) & 0xffffffffull;
            if (id >= lo) {
                const unsigned long long bkt = (id - lo) >> shift;
                atomicAdd(&hist[bkt < 1023ull ? (uint32_t)bkt : 1023u], 1u);
            }
        });
        __syncthreads();
        if (t == 0u) {
            uint32_t acc = 0, k = 0, rest = 0;
            while (k < 1024u && acc + hist[k] <= kCrowdCap) { acc += hist[k]; ++k; }
            for (uint32_t q = k; q < 1024u; ++q) rest += hist[q];
            s.misc[1] = k; s.misc[2] = acc; s.misc[3] = rest;
        }
        __syncthreads();
        const uint32_t nb = s.misc[1], m = s.misc[2], rest = s.misc[3];
        __syncthreads();
        if (nb == 0u) { shift = shift > 4u ? shift - 4u : 0u; continue; }     // the first bucket alone is too large: finer buckets
        // (bucket 1023 also holds everything beyond it: taken only together with all the others = the rest of the run)
        const unsigned long long hi = nb == 1024u ? 0x100000000ull : lo + ((unsigned long long)nb << shift);
        for_texel([&](unsigned long long k) {
            const unsigned long long id = (k >> 24) & 0xffffffffull;
            if (id >= lo && id < hi) skey[atomicAdd(&s.misc[0], 1u)] = k;
        });
        __syncthreads();
        bin_bitonic(s, m);
        bin_blend_long<MODE>(s, p, begin, m, owner, d, [&](uint32_t j) { return (uint32_t)(skey[order[j]] & 0xffffffull); });
        if (rest == 0u) break;
        lo = hi;
    }
}

// A bin of up to kBinCap fragments with a run longer than kRankMaxRun (bins of MORE fragments are spread over many
// workgroups: crowd_*_kernel below): batches of whole texels that fit kCrowdCap sort keys, each ordered in LDS; a texel
// that does not fit alone goes in windows of its stream indices.
template <int MODE>
TH_D void bin_crowded(BinShared<MODE> &s, const DepositParams &p, const unsigned long long *keys, uint32_t begin, uint32_t n,
                      BinTexel<MODE> &d, bool &touched)
{
    const uint32_t t = threadIdx.x;
    unsigned long long *skey = s.skey();
    uint16_t *order = s.order();
    auto src_at = [&](uint32_t base) { return [&, base](uint32_t j) { return (uint32_t)(skey[order[base + j]] & 0xffffffull); }; };
    s.cnt[t] = 0u;
    __syncthreads();
    for_keys(keys, n, [&](unsigned long long k, uint32_t) { atomicAdd(&s.cnt[key_local(k)], 1u); });
    __syncthreads();
    bin_scan_counts(s);
    uint32_t t0 = 0;
    while (t0 < kBinTexels) {
        // (every thread walks the same table: workgroup-uniform)
        uint32_t t1 = t0, longest = 0;
        while (t1 < kBinTexels && s.first[t1 + 1u] - s.first[t0] <= kCrowdCap) { const uint32_t l = s.first[t1 + 1u] - s.first[t1]; longest = l > longest ? l : longest; ++t1; }
        if (t1 > t0) {
            const uint32_t m = s.first[t1] - s.first[t0], base = s.first[t0];
            if (m) {
                if (t >= t0 && t < t1) s.cnt[t] = 0u;
                __syncthreads();
                for_keys(keys, n, [&](unsigned long long k, uint32_t f) {
                    const uint32_t lt = key_local(k);
                    if (lt >= t0 && lt < t1) skey[s.first[lt] - base + atomicAdd(&s.cnt[lt], 1u)] = sort_key(k, f);
                });
                __syncthreads();
                if (longest <= kRankMaxRun) {
                    // rank by counting: a fragment's place in its run = the fragments of the run with a smaller key
                    for (uint32_t q = t; q < m; q += 256u) {
                        const unsigned long long k = skey[q];
                        const uint32_t lt = (uint32_t)(k >> 56), r0 = s.first[lt] - base, r1 = s.first[lt + 1u] - base;
                        uint32_t rank = 0;
                        for (uint32_t j = r0; j < r1; ++j) rank += skey[j] < k ? 1u : 0u;
                        order[r0 + rank] = (uint16_t)q;
                    }
                    __syncthreads();
                } else bin_bitonic(s, m);
                const bool mine = t >= t0 && t < t1;
                const uint32_t r0 = mine ? s.first[t] - base : 0u, len = mine ? s.first[t + 1u] - s.first[t] : 0u;
                if (len && len <= kOwnRun) { touched = true; bin_blend_own<MODE>(p, begin, len, d, src_at(r0)); }
                if (longest > kOwnRun)
                    for (uint32_t lt = t0; lt < t1; ++lt) {        // the long runs one after the other, by everybody
                        const uint32_t l = s.first[lt + 1u] - s.first[lt];
                        if (l <= kOwnRun) continue;
                        if (t == lt) touched = true;
                        bin_blend_long<MODE>(s, p, begin, l, lt, d, src_at(s.first[lt] - base));
                    }
                __syncthreads();
            }
            t0 = t1;
            continue;
        }
        // texel t0 alone holds more than kCrowdCap fragments
        if (t == t0) touched = true;
        blend_texel_windows<MODE>(s, p, begin, t0, d, [&](auto body) {
            for_keys(keys, n, [&](unsigned long long k, uint32_t f) { if (key_local(k) == t0) body(sort_key(k, f)); });
        });
        ++t0;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void bins_blend_kernel(const DepositParams p)
{
    __shared__ BinShared<MODE> s;
    const uint32_t b = blockIdx.x, begin = p.bin_start[b], n = p.bin_start[b + 1u] - begin;
    if (n == 0u || n > kBinCap) return;              // (bins of more than kBinCap fragments: crowd_*_kernel)
    const uint32_t t = threadIdx.x;
    const unsigned long long *keys = p.frag_keys + begin;
    const uint32_t by = b / p.bins_x, bx = b - by * p.bins_x;
    const uint32_t x = (bx << kBinShift) + (t & (kBinSide - 1u)), y = (by << kBinShift) + (t >> kBinShift);
    const bool inside = x < (uint32_t)p.fw && y < (uint32_t)p.fh;
    const uint32_t texel = inside ? y * (uint32_t)p.fw + x : 0u;
    BinTexel<MODE> d{};
    if (inside) d.load(p, texel);
    bool touched = false;

    bool crowded = false;
    {
        // The common case.  A thread's <= 16 keys stay in registers from the count to the ranking: counted per texel, the
        // runs laid out by the scan, every fragment's stream index dropped into its texel's run (in whatever order the LDS
        // atomics hand out), then every fragment ranks itself inside its run by counting the smaller indices - its place
        // in GL's order - and leaves its position there for the texel's thread.
        constexpr uint32_t kPer = kBinCap / 256u;
        unsigned long long k[kPer];
        uint32_t at[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = q * 256u + t; k[q] = keys[f < n ? f : n - 1u]; }
        s.cnt[t] = 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) if (q * 256u + t < n) atomicAdd(&s.cnt[key_local(k[q])], 1u);
        __syncthreads();
        const uint32_t mine = s.cnt[t];
        const uint32_t longest = bin_scan_counts(s);
        crowded = longest > kRankMaxRun;
        if (!crowded) {
            s.cnt[t] = 0u;
            __syncthreads();
            uint32_t *sid = s.sid(), *osrc = s.osrc();
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q)
                if (q * 256u + t < n) {
                    const uint32_t lt = key_local(k[q]);
                    at[q] = s.first[lt] + atomicAdd(&s.cnt[lt], 1u);
                    sid[at[q]] = (uint32_t)k[q];
                }
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q)
                if (q * 256u + t < n) {
                    const uint32_t lt = key_local(k[q]), id = (uint32_t)k[q], r0 = s.first[lt], r1 = s.first[lt + 1u];
                    uint32_t rank = 0;
                    for (uint32_t j = r0; j < r1; ++j) rank += sid[j] < id ? 1u : 0u;
                    osrc[r0 + rank] = q * 256u + t;
                }
            __syncthreads();
            const uint32_t r0 = s.first[t];
            auto src_at = [&](uint32_t base) { return [osrc, base](uint32_t j) { return osrc[base + j]; }; };
            if (mine && mine <= kOwnRun) { touched = true; bin_blend_own<MODE>(p, begin, mine, d, src_at(r0)); }
            if (longest > kOwnRun)
                for (uint32_t lt = 0; lt < kBinTexels; ++lt) {
                    const uint32_t l = s.first[lt + 1u] - s.first[lt];
                    if (l <= kOwnRun) continue;
                    if (t == lt) touched = true;
                    bin_blend_long<MODE>(s, p, begin, l, lt, d, src_at(s.first[lt]));
                }
        }
    }
    if (crowded) bin_crowded<MODE>(s, p, keys, begin, n, d, touched);
    if (inside && touched) d.store(p, texel);
}


// ---- bins of more than kBinCap fragments: many workgroups per bin ------------------------------------------------------
// The wake makes particles converge: after a few dozen frames a few hundred bins hold a third of all fragments (tens of
// thousands each, a thousand in single texels).  Those bins get one more level of the same scheme: their fragments are
// regrouped by TEXEL - a counting sort over the 256 texels of the bin, kCrowdBlock fragments per workgroup, exact ranges
// from a per-bin scan - and then every texel's run is ordered by stream index and blended by a workgroup of its own.
//   crowd_plan_kernel     workgroup blocks of the large bins: first block of every large bin (prefix over the list)
//   crowd_hist_kernel     fragments per texel of every large bin
//   crowd_scan_kernel     every texel's range inside its bin
//   crowd_scatter_kernel  (stream index << 24 | position of the varying) of every fragment into its texel's range
//   crowd_blend_kernel    one workgroup per texel of a large bin: order the run in LDS (windows of stream indices when it
//                         does not fit), blend
constexpr uint32_t kCrowdBlock = 4096;

__global__ __launch_bounds__(1024) void crowd_plan_kernel(const DepositParams p, uint32_t *totals)
{
    __shared__ uint32_t part[1024];
    const uint32_t nlarge = totals[3], per = (nlarge + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per < nlarge ? threadIdx.x * per : nlarge, hi = lo + per < nlarge ? lo + per : nlarge;
    uint32_t n = 0;
    for (uint32_t i = lo; i < hi; ++i) n += (p.bin_hist[p.large_bins[i]] + kCrowdBlock - 1u) / kCrowdBlock;
    part[threadIdx.x] = n;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t a = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += a;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - n;
    for (uint32_t i = lo; i < hi; ++i) { p.large_block0[i] = run; run += (p.bin_hist[p.large_bins[i]] + kCrowdBlock - 1u) / kCrowdBlock; }
    if (threadIdx.x == 1023u) { p.large_block0[nlarge] = part[1023]; totals[4] = part[1023]; }
}

// block -> (large bin i, first fragment of the block inside the bin)
TH_D void crowd_block(const DepositParams &p, uint32_t block, uint32_t &i, uint32_t &first)
{
    uint32_t lo = 0, hi = p.nlarge;                 // the last i with large_block0[i] <= block
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (p.large_block0[mid] <= block) lo = mid; else hi = mid; }
    i = lo;
    first = (block - p.large_block0[lo]) * kCrowdBlock;
}

__global__ __launch_bounds__(256) void crowd_hist_kernel(const DepositParams p)
{
    __shared__ uint32_t hist[kBinTexels];
    uint32_t i, first;
    crowd_block(p, blockIdx.x, i, first);
    const uint32_t b = p.large_bins[i], begin = p.bin_start[b], n = p.bin_start[b + 1u] - begin;
    const uint32_t m = n - first < kCrowdBlock ? n - first : kCrowdBlock;
    hist[threadIdx.x] = 0u;
    __syncthreads();
    for_keys(p.frag_keys + begin + first, m, [&](unsigned long long k, uint32_t) { atomicAdd(&hist[key_local(k)], 1u); });
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&p.crowd_count[(size_t)i * kBinTexels + threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void crowd_scan_kernel(const DepositParams p)
{
    __shared__ uint32_t wave_total[4];
    const uint32_t i = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t mine = p.crowd_count[(size_t)i * kBinTexels + t];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)lane >= o) incl += up; }
    if (lane == 63u) wave_total[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_total[w];
    const uint32_t start = before + incl - mine;
    p.crowd_start[(size_t)i * (kBinTexels + 1u) + t] = start;
    p.crowd_cursor[(size_t)i * kBinTexels + t] = start;
    if (t == 255u) p.crowd_start[(size_t)i * (kBinTexels + 1u) + 256u] = before + incl;
}

__global__ __launch_bounds__(256) void crowd_scatter_kernel(const DepositParams p)
{
    __shared__ uint32_t hist[kBinTexels], base[kBinTexels];
    uint32_t i, first;
    crowd_block(p, blockIdx.x, i, first);
    const uint32_t b = p.large_bins[i], begin = p.bin_start[b], n = p.bin_start[b + 1u] - begin;
    const uint32_t m = n - first < kCrowdBlock ? n - first : kCrowdBlock, t = threadIdx.x;
    constexpr uint32_t kPer = kCrowdBlock / 256u;
    unsigned long long k[kPer];
    const unsigned long long *keys = p.frag_keys + begin + first;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = q * 256u + t; k[q] = keys[f < m ? f : m - 1u]; }
    hist[t] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) if (q * 256u + t < m) atomicAdd(&hist[key_local(k[q])], 1u);
    __syncthreads();
    base[t] = hist[t] ? atomicAdd(&p.crowd_cursor[(size_t)i * kBinTexels + t], hist[t]) : 0u;
    hist[t] = 0u;
    __syncthreads();
    unsigned long long *out = p.crowd_keys + begin;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q)
        if (q * 256u + t < m) {
            const uint32_t lt = key_local(k[q]);
            out[base[lt] + atomicAdd(&hist[lt], 1u)] = ((k[q] & 0xffffffffull) << 24) | (first + q * 256u + t);
        }
}

template <int MODE>
__global__ __launch_bounds__(256) void crowd_blend_kernel(const DepositParams p)
{
    __shared__ BinShared<MODE> s;
    const uint32_t i = blockIdx.x >> 8, lt = blockIdx.x & 255u, t = threadIdx.x;
    const uint32_t b = p.large_bins[i], begin = p.bin_start[b];
    const uint32_t r0 = p.crowd_start[(size_t)i * (kBinTexels + 1u) + lt], len = p.crowd_start[(size_t)i * (kBinTexels + 1u) + lt + 1u] - r0;
    if (len == 0u) return;
    const uint32_t by = b / p.bins_x, bx = b - by * p.bins_x;
    const uint32_t x = (bx << kBinShift) + (lt & (kBinSide - 1u)), y = (by << kBinShift) + (lt >> kBinShift);
    const uint32_t texel = y * (uint32_t)p.fw + x;            // (a texel with fragments lies inside the target)
    const unsigned long long *run = p.crowd_keys + begin + r0;
    BinTexel<MODE> d{};
    if (t == 0u) d.load(p, texel);
    unsigned long long *skey = s.skey();
    uint16_t *order = s.order();
    if (len <= kCrowdCap) {
        for_keys(run, len, [&](unsigned long long k, uint32_t f) { skey[f] = k; });
        __syncthreads();
        if (len <= kRankMaxRun) {
            if (t < len) {
                const unsigned long long k = skey[t];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < len; ++j) rank += skey[j] < k ? 1u : 0u;
                order[rank] = (uint16_t)t;
            }
            __syncthreads();
        } else bin_bitonic(s, len);
        bin_blend_long<MODE>(s, p, begin, len, 0u, d, [&](uint32_t j) { return (uint32_t)(skey[order[j]] & 0xffffffull); });
    } else {
        blend_texel_windows<MODE>(s, p, begin, 0u, d, [&](auto body) { for_keys(run, len, [&](unsigned long long k, uint32_t) { body(k); }); });
    }
    if (t == 0u) d.store(p, texel);
}

}  // namespace

static uint32_t slot_blocks(const DepositParams &p) { const uint32_t n = (p.W * p.rows + 255u) / 256u; return n ? n : 1u; }

void launch_bins_raster(const DepositParams &p, hipStream_t s)
{
    (void)hipMemsetAsync(p.list_n, 0, (size_t)2 * kDepLists * kDepListStride * sizeof(uint32_t), s);
    (void)hipMemsetAsync(p.rep_hist, 0, (size_t)kBinReplicas * p.bin_stride * sizeof(uint32_t), s);
    hipLaunchKernelGGL(bins_raster_kernel, dim3(slot_blocks(p)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_raster_slow_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_count_long_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
}

void launch_bins_scan(const DepositParams &p, uint32_t *totals, hipStream_t s)
{
    hipLaunchKernelGGL(bins_replica_kernel, dim3((p.nbins + 255u) / 256u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_scan_kernel, dim3(1), dim3(1024), 0, s, p, totals);
    hipLaunchKernelGGL(crowd_plan_kernel, dim3(1), dim3(1024), 0, s, p, totals);
}

void launch_bins_emit(const DepositParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(bins_cursor_kernel, dim3((p.nbins + 255u) / 256u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_emit_kernel, dim3(slot_blocks(p)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_emit_long_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
}

// p.nlarge / nblocks: the large bins and their kCrowdBlock-fragment blocks, as the scan counted them (totals[3], totals[4])
void launch_bins_blend(const DepositParams &p, uint32_t nblocks, hipStream_t s)
{
    if (p.nlarge) {         // regroup the large bins by texel first: the two blend kernels then overlap at the tail
        (void)hipMemsetAsync(p.crowd_count, 0, (size_t)p.nlarge * kBinTexels * sizeof(uint32_t), s);
        hipLaunchKernelGGL(crowd_hist_kernel, dim3(nblocks), dim3(256), 0, s, p);
        hipLaunchKernelGGL(crowd_scan_kernel, dim3(p.nlarge), dim3(256), 0, s, p);
        hipLaunchKernelGGL(crowd_scatter_kernel, dim3(nblocks), dim3(256), 0, s, p);
    }
#define TH_GO(M) do { if (p.nlarge) hipLaunchKernelGGL(crowd_blend_kernel<M>, dim3(p.nlarge * kBinTexels), dim3(256), 0, s, p); \
                      hipLaunchKernelGGL(bins_blend_kernel<M>, dim3(p.nbins), dim3(256), 0, s, p); } while (0)
    if (p.mode == 0) TH_GO(0); else if (p.mode == 1) TH_GO(1); else TH_GO(2);
#undef TH_GO
}
size_t crowd_words_per_bin() { return 3u * kBinTexels + 1u; }

}  // namespace th
