// th_bins.hip - the binned draw() pipeline: Tendrils.draw()'s particle lines (src/index.js:278-337) blended into the flow
// field and / or the view buffer from particles held in ANY slot order - in particular the tile-sorted order the
// integrator steps over (th_kernels.hip "Tile-sorted slot order"), so that the reference's frame loop - step(); draw()
// (src/demo.main.js:1082) - never has to return the state to texel order.
//
// Same lines, same rasteriser, same varyings, same blend arithmetic as th_deposit.hip (th_raster.hpp); what differs is how
// GL's primitive order is reconstructed.  The stream-ordered pipeline produces the fragments in stream order and sorts
// them stably by texel with three global radix passes + a gather.  Here the order is restored where the fragments
// meet - inside one 16 x 16-texel bin of the target, in LDS - and a line is touched ONCE:
//   1. bins_fused_kernel: one thread per SLOT (coalesced state reads whatever the order): the line is rasterised and
//      every covered texel's fragment - (texel, stream index) key + varying(s) - goes straight into the bin it falls
//      into.  No counting pass: a bin is kBinReplicas LISTS of PAGES of kBinPage places (page 0 of list r of bin b is
//      page b * kBinReplicas + r - a bin's first pages lie side by side - further pages come from a pool as a list
//      grows); a line is rasterised in registers first (a record of <= kRecordTexels texels) and reserves EXACTLY its
//      fragments per bin - the reservations of a workgroup's lines are added up in an LDS table, ONE global atomic
//      per (workgroup, bin) moves the cursor of the list the workgroup uses (a crowded bin is met by thousands of
//      workgroups, and device-wide atomics on one cache line are served one after the other at the memory side - a
//      thousand of them set the time of a whole pass - hence the lists, their cursors kept bin_stride words apart), and the
//      workgroup whose reservation crosses into a new page takes that page from the pool and publishes it.  Every place
//      handed out is written (a pass that ran out of pages is flagged and repeated before anything reads it).
//      bins_listed_kernel: the few lines that cross the view's edge (clipped) and the lines of more fragments than a record
//      holds, one place at a time.  bins_span_kernel: the lines that SPAN the view - the drifting rows / columns of some shapes
//      join unrelated particles (th_order.hip: line_rows; th::LineSources is how their vertices are found in a slot order) - a
//      wave each: rows to lanes, texels dealt evenly, one cursor atomic per bin a batch of 64 fragments meets.
//   2. bins_plan_kernel / crowd_plan_kernel: the bins of more than kBinCap places ("large").
//   3. bins_blend_kernel: one workgroup per bin of up to kBinCap places: its fragments grouped by texel (LDS counting sort), every
//      texel's run ordered by the stream index of its line (short runs: rank by counting; long ones: bitonic sort) and
//      blended in that order by the texel's thread - dst = src*a + dst*(1-a), fragment after fragment, GL's order and
//      arithmetic.
//      crowd_*_kernel: the large bins get one more level of the same scheme - regrouped by texel, then by the length of a
//      texel's run: up to 256 fragments ordered by a wave and walked by four lanes, up to 1024 ordered by a wave
//      (long_sort_kernel), beyond that parted by stream index and ordered window by window (giant_*_kernel), both walked by
//      a wave per run and target (run_walk_kernel).
// The stream index of a line is a pure function of its particle id, so the result is the stream-ordered pipeline's, and
// the restatement's, bit for bit, whatever the slot order and whatever the atomics did.
#include "th_kernels.hpp"
#include "th_raster.hpp"

namespace th {
namespace {

constexpr uint32_t kBinSide = 1u << kBinShift, kBinTexels = kBinSide * kBinSide;      // 256 texels = one per thread
constexpr uint32_t kPageShift = 8;
static_assert(kBinPage == 1u << kPageShift && kBinPage * kBinReplicas == kBinCap, "page size");
static_assert(sizeof(BlendSource) == 5 * sizeof(float), "bin_blend_long reads a source's components as five floats side by side");
constexpr uint32_t kRankMaxRun = 256;        // runs up to this length are ordered by counting, longer ones by the bitonic network
constexpr uint32_t kOwnRun = 256;            // runs up to this length are blended by their texel's thread alone (64 texels of a wave side by side: a chain per lane, nothing redundant)
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kNoPlace = 0xffffffffu;
constexpr uint32_t kPageSpins = 1u << 19;        // tries of page_of<WAIT> (each a sleep of ~128 cycles and a load: ~50 ms in all) before it gives up

TH_D uint32_t bin_of(const DepositParams &p, uint32_t x, uint32_t y) { return (y >> kBinShift) * p.bins_x + (x >> kBinShift); }
TH_D void bins_flag(const DepositParams &p, uint32_t what) { atomicOr(&p.totals[kTotFlags], what); }
TH_D void wave_sync_lds() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// ---- pages ---------------------------------------------------------------------------------------------------------------
// Place `v` (a virtual index handed out by the list's cursor) of list `list` = bin * kBinReplicas + r -> position in the
// key / varying arrays.  The page a place lies in was taken from the pool by whoever's reservation contained the page's
// FIRST place - a thread that had moved the cursor before this one and publishes the page right after, without waiting for
// anybody: the wait below always ends.  Should that ever not hold (a reservation lost, a table entry overwritten), the wait
// gives up after kPageSpins tries - tens of milliseconds, a thousand times the longest wait a pass has been seen to make -:
// the place is nowhere, the pass is flagged (kBinsWaitBroken) and the host repeats the draw in stream order like any pass
// it cannot trust, instead of a launch that never ends.
// (WAIT: inside the pass that hands the pages out - a relaxed device-scope load per try: only the entry's own value is
// needed, nothing else is published with it, and an acquire would empty the caches at every fragment; readers of later
// launches load plainly)
template <bool WAIT>
TH_D uint32_t page_of(const DepositParams &p, uint32_t list, uint32_t pn)
{
    if (pn == 0u) return list;
    if (pn >= p.max_pages) return kNoPlace;
    uint32_t *slot = &p.page_table[(size_t)list * p.max_pages + pn];
    if constexpr (!WAIT) return *slot;
    uint32_t id = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t spins = 0; id == 0u; ++spins) {
        if (spins >= kPageSpins) { bins_flag(p, kBinsWaitBroken); return kNoPlace; }
        __builtin_amdgcn_s_sleep(2);
        id = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return id;                                      // (kNoPlace: the pool was exhausted - flagged by the allocator)
}
template <bool WAIT = false>
TH_D uint32_t place_of(const DepositParams &p, uint32_t list, uint32_t v)
{
    const uint32_t id = page_of<WAIT>(p, list, v >> kPageShift);
    return id == kNoPlace ? kNoPlace : (id << kPageShift) | (v & (kBinPage - 1u));
}
// the pages that start inside the reservation [base, base + n) of `list` (n >= 1): taken from the pool and published
TH_D void pages_open(const DepositParams &p, uint32_t list, uint32_t base, uint32_t n)
{
    uint32_t pn = (base + kBinPage - 1u) >> kPageShift;
    if (pn == 0u) pn = 1u;
    const uint32_t last = (base + n - 1u) >> kPageShift;
    for (; pn <= last; ++pn) {
        if (pn >= p.max_pages) { bins_flag(p, kBinsBinFull); break; }
        const uint32_t k = atomicAdd(&p.totals[kTotPool], 1u);
        uint32_t id = p.nbins * kBinReplicas + k;
        if (k >= p.pool_pages) { id = kNoPlace; bins_flag(p, kBinsPoolExhausted); }
        __hip_atomic_store(&p.page_table[(size_t)list * p.max_pages + pn], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// a reader is done with list `list` of `n` places: its table entries are left empty for the next pass
TH_D void pages_forget(const DepositParams &p, uint32_t list, uint32_t n, uint32_t tid, uint32_t threads)
{
    const uint32_t pages = (n + kBinPage - 1u) >> kPageShift;
    for (uint32_t pn = 1u + tid; pn < pages && pn < p.max_pages; pn += threads) p.page_table[(size_t)list * p.max_pages + pn] = 0u;
}
// (the cursors of one list index r, transposed in 32 columns: neighbouring bins' cursors 1 KB apart, not in one 128-byte line -
// the lines that take their places one by one, bins_listed_kernel, meet neighbouring bins at the same time: 50 -> 44 us)
TH_D uint32_t *list_cursor(const DepositParams &p, uint32_t bin, uint32_t r)
{
    return p.bin_cursor + (size_t)r * p.bin_stride + (bin & 31u) * ((p.nbins + 31u) >> 5) + (bin >> 5);
}

// one fragment into place `at` (kNoPlace: nowhere - the pass is flagged and repeated)
TH_D void bins_put(const DepositParams &p, const DepositLine &L, uint32_t id, uint32_t at, int x, int y)
{
    if (at == kNoPlace) return;
    p.frag_keys[at] = ((unsigned long long)(((uint32_t)y << 12) | (uint32_t)x) << 32) | id;
    float t = 0.0f;
    const bool along = dep_param(L, x, y, t);
    if (p.mode == 2) {
        p.colors[2u * (size_t)at] = dep_mix(L.a.c, L.b.c, along, t);
        p.colors[2u * (size_t)at + 1u] = dep_mix(L.a.c2, L.b.c2, along, t);
    } else p.colors[at] = dep_mix(L.a.c, L.b.c, along, t);
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

// ---- the reservations of a workgroup's lines, by bin ----------------------------------------------------------------------
template <uint32_t N>                       // (a power of two)
struct Reservations {
    uint32_t tag[N];                         // bin + 1 (0: free)
    uint32_t sum[N];                         // places reserved by the workgroup's lines; after the flush: the first of them
};
// n places of `bin` for the calling line -> entry << 20 | offset inside the workgroup's share
template <uint32_t N>
TH_D uint32_t resv_take(Reservations<N> &t, uint32_t bin, uint32_t n)
{
    uint32_t h = ((bin * 2654435761u) >> 8) & (N - 1u);
    for (;;) {
        const uint32_t old = atomicCAS(&t.tag[h], 0u, bin + 1u);
        if (old == 0u || old == bin + 1u) break;
        h = (h + 1u) & (N - 1u);
    }
    return (h << 20) | atomicAdd(&t.sum[h], n);
}

// The fragments of a line of up to kRecordTexels fragments, by bin.  A line is about a texel long: nearly always all of its
// fragments fall into one bin (b0), sometimes into two (b1); fragments of a third bin (`others`) take their places one by one.
struct LineBins { uint32_t bin[kRecordTexels], b0, b1, c0, c1, others; };
TH_D LineBins line_bins(const DepositParams &p, const uint32_t (&xy)[kRecordTexels], uint32_t n)
{
    LineBins q;
    q.b0 = q.b1 = kNoPlace; q.c0 = q.c1 = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) q.bin[k] = k < n ? bin_of(p, xy[k] & 0xffffu, xy[k] >> 16) : kNoPlace;
    q.b0 = q.bin[0];
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) {
        const bool in0 = k < n && q.bin[k] == q.b0;
        q.c0 += in0 ? 1u : 0u;
        if (k < n && !in0 && q.b1 == kNoPlace) q.b1 = q.bin[k];
    }
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) q.c1 += (k < n && q.bin[k] == q.b1 && q.b1 != kNoPlace) ? 1u : 0u;
    q.others = n - q.c0 - q.c1;
    return q;
}

// one place of list `list`, straight from its cursor (the odd fragment: slow lines, a line's third bin)
TH_D uint32_t place_single(const DepositParams &p, uint32_t bin, uint32_t rep)
{
    const uint32_t v = atomicAdd(list_cursor(p, bin, rep), 1u);
    if (v == 0xffffffffu) { bins_flag(p, kBinsBinFull); return kNoPlace; }
    if (v && (v & (kBinPage - 1u)) == 0u) pages_open(p, bin * kBinReplicas + rep, v, 1u);
    return place_of<true>(p, bin * kBinReplicas + rep, v);
}

// all threads of a 1024-thread workgroup: `mine` -> its exclusive prefix over the workgroup; `total` (same on every thread)
TH_D unsigned long long block_scan_1024(unsigned long long *lds, unsigned long long mine, unsigned long long &total)
{
    const uint32_t t = threadIdx.x;
    __syncthreads();
    lds[t] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const unsigned long long a = t >= off ? lds[t - off] : 0ull;
        __syncthreads();
        lds[t] += a;
        __syncthreads();
    }
    total = lds[1023];
    return lds[t] - mine;
}

// The blocks of 256 slots in which some slot's line can draw at all (a property of the slot order and the shape alone: found
// once per order), in rising order.
// (the same walk over a new slot order notes where the texels lie that other lines' vertices read - LineSources, shapes whose
// lookup drifts off the line's own texel: src_slots)
__global__ __launch_bounds__(256) void bins_block_flags_kernel(const DepositParams p, uint8_t *flags, uint32_t *src_slots)
{
    __shared__ uint32_t any;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    uint32_t col = 0, row = 0;
    const bool can = s < p.W * p.rows && slot_particle(p, s, col, row);
    if (src_slots && s < p.W * p.rows) {
        const uint32_t ri = p.src.row_index[row], ci = p.src.col_index[col];
        if (ri != 0xffffu) src_slots[(size_t)ri * p.W + col] = s;
        if (ci != 0xffffu) src_slots[(size_t)p.src.nrows * p.W + (size_t)ci * p.rows + row] = s;
    }
    if (threadIdx.x == 0u) any = 0u;
    __syncthreads();
    if (can && (__lane_id() == (uint32_t)__builtin_ctzll(__ballot(can)))) any = 1u;
    __syncthreads();
    if (threadIdx.x == 0u) flags[blockIdx.x] = (uint8_t)any;
}
__global__ __launch_bounds__(1024) void bins_block_list_kernel(const uint8_t *flags, uint32_t blocks, uint32_t *list, uint32_t *count)
{
    __shared__ unsigned long long lds[1024];
    const uint32_t per = (blocks + 1023u) / 1024u, lo = threadIdx.x * per < blocks ? threadIdx.x * per : blocks, hi = lo + per < blocks ? lo + per : blocks;
    unsigned long long n = 0, total = 0;
    for (uint32_t b = lo; b < hi; ++b) n += flags[b];
    uint32_t at = (uint32_t)block_scan_1024(lds, n, total);
    for (uint32_t b = lo; b < hi; ++b) if (flags[b]) list[at++] = b;
    if (threadIdx.x == 0u) *count = (uint32_t)total;
}

__global__ __launch_bounds__(256) void bins_zero_kernel(uint32_t *a, uint32_t na, uint32_t *b, uint32_t nb, uint32_t *c, uint32_t nc)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < na + nb + nc; i += gridDim.x * 256u) {
        if (i < na) a[i] = 0u;
        else if (i < na + nb) b[i - na] = 0u;
        else c[i - na - nb] = 0u;
    }
}

// pass 1.  Per slot: the line set up, classified, and - the common case: a small hexagon inside the view, all in registers -
// rasterised into a record of <= kRecordTexels texels; its places reserved per bin, exactly; its fragments written.
// a line's hexagon and its record while the wave's rows are dealt to its lanes (LDS, one per line of the workgroup)
constexpr uint32_t kMaxRowsDealt = 12;       // lines of more rows (none of ordinary length: a line is <= 10 texels long) walk their own rows
struct LineStage {
    int bx, by;                              // the hexagon's least x and y (sixteenths of a texel) ...
    uint32_t pxy[6];                         // ... and its corners from there: x | y << 16 (a small hexagon is less than 4096 x 1024 sixteenths)
    uint32_t first;                          // first of the line's (line, row) pairs among the wave's
    uint32_t n;                              // fragments so far
    uint32_t rec[kRecordTexels];
    uint32_t pad;                            // (19 words: an odd stride - a wave's lines in different banks)
};

// (six workgroups per CU: 26 KB of LDS each - a table of 2 entries per line, rows dealt up to 12 per line, the hexagon packed - and
// at most 80 VGPRs; four workgroups of 37 KB and 118 VGPRs: 580 us, five of 31 KB and 81: 492-512, profiles/r5_h)
// PLAIN: f32 texels and every vertex of every line its own particle (the shapes of the frame loop this kernel was tuned on: the
// own texels go through the kernel by value); otherwise - a packed ring, or a shape whose vertex lookup drifts off the line's own
// texel (LineSources) - the vertices are fetched by dep_fetch, again when the varyings are made.
template <uint32_t BS, bool DEAL, bool PLAIN>
__global__ __launch_bounds__(BS, 6) void bins_fused_kernel(const DepositParams p)
{
    static_assert(DEAL, "the last phase reads every line's record from the stage (every lane walking its own line's rows was measured and dropped: 0.65 against 0.58 ms)");
    constexpr uint32_t kTab = BS * 2u;          // table entries: <= 2 bins per line reserve here (a third bin goes to its cursor directly): never more bins than entries
    __shared__ Reservations<kTab> t;
    __shared__ LineStage stage[DEAL ? BS : 1u];
    __shared__ uint8_t owner[DEAL ? BS / 64u : 1u][DEAL ? 64u * kMaxRowsDealt : 1u];     // per wave: the line (lane) of every dealt row
    const uint32_t slots = p.W * p.rows;
    // In the tile-sorted order the particles whose lines can draw lie apart from the others inside every tile
    // (th_kernels.hip: tile_key): most blocks of 256 slots meet only one kind - whole waves of lines that exist, or nothing to
    // do.  The blocks with something to do are listed once per slot order (bins_block_list_kernel): half of all blocks never
    // start.
    // (ONE listed block per workgroup, no loop: what the body computes from the pass's parameters alone - a dozen integer-to-float
    // conversions - is then computed where it is used; hoisted out of a loop over blocks it was held in registers across the
    // whole body, and with 96 of them spilled to scratch memory)
    if (blockIdx.x >= p.draw_nblocks) return;
    const uint32_t block = p.draw_blocks[blockIdx.x];
    // (a frame loop: the step that wrote these slots saw every line of the block end up beyond one edge of the view - nothing
    // of it would be rasterised, listed or counted: LogicParams::seen)
    if (p.block_seen && p.block_seen[block] == 0u) return;
    const uint32_t s = block * BS + threadIdx.x;
    uint32_t col = 0, row = 0;
    const bool can = s < slots && slot_particle(p, s, col, row);
    for (uint32_t e = threadIdx.x; e < kTab; e += BS) { t.tag[e] = 0u; t.sum[e] = 0u; }
    __syncthreads();

    OwnTexels own;                                      // (both ends of the line, before anything else)
    if constexpr (PLAIN) { if (can) { own.have = true; own.cur = p.cur[s]; own.prev = p.prev[s]; } }
    else if (can) { own.have = true; own.cur = dep_state(p, p.cur, s); own.prev = dep_state(p, p.prev, s); }
    DepositLine L;
    L.draws = false;
    LineRecord r{};
    bool slow = false;
    uint32_t dealt = 0;                                 // rows of this line handed to the wave's lanes
    if (can) {
        dep_setup<PLAIN>(p, col, p.row0 + row, L, s, own, false);          // (the varyings: once the line is known to cover a texel)
        if (L.draws) {
            float cx[6], cy[6];
            const int where = dep_hexagon(p, L, cx, cy);
            if (where == kHexInside) {
                int PX[6], PY[6], ymin, ymax;
                dep_snap_hexagon(p, cx, cy, PX, PY);
                if (dep_hexagon_is_small(PX, PY, ymin, ymax)) {
                    int r0 = (ymin + 15) >> 4, r1 = (ymax + 15) >> 4;
                    r0 = r0 < 0 ? 0 : r0; r1 = r1 > p.fh ? p.fh : r1;
                    if (DEAL && r1 - r0 <= (int)kMaxRowsDealt) {
                        dealt = r1 > r0 ? (uint32_t)(r1 - r0) : 0u;
                        LineStage &g = stage[threadIdx.x];
                        int xmin = PX[0];
#pragma unroll
                        for (int k = 1; k < 6; ++k) xmin = PX[k] < xmin ? PX[k] : xmin;
                        g.bx = xmin; g.by = ymin;
#pragma unroll
                        for (int k = 0; k < 6; ++k) g.pxy[k] = (uint32_t)(PX[k] - xmin) | ((uint32_t)(PY[k] - ymin) << 16);
                    } else dep_raster_small_hexagon2(p, PX, PY, ymin, ymax, [&](int x, int y) { rec_add(r, x, y); });
                } else slow = true;
            } else if (where == kHexClip) slow = true;
        }
    }
    if constexpr (DEAL) {
        // The rows of the wave's lines, dealt evenly to its lanes: a lane rasterises ONE row of some line per round (two
        // divisions), whatever the lengths of the lines - walking its own line's rows, a wave runs as many rounds as its
        // longest line has rows with most lanes idle.  Wave-private LDS, no workgroup barrier.
        const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
        uint32_t incl = dealt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)lane >= o) incl += up; }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63), first = incl - dealt;
        stage[threadIdx.x].first = first; stage[threadIdx.x].n = 0u;
        for (uint32_t k = 0; k < dealt; ++k) owner[wave][first + k] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
            const uint32_t g = g0 + lane;
            if (g < total) {
                LineStage &q = stage[(wave << 6) + owner[wave][g]];
                int QX[6], QY[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) { const uint32_t w = q.pxy[k]; QX[k] = q.bx + (int)(w & 0xffffu); QY[k] = q.by + (int)(w >> 16); }
                const int top = (q.by + 15) >> 4;                      // (the line's first row, as its thread clamped it)
                const int y = (top < 0 ? 0 : top) + (int)(g - q.first);
                int left, right;
                dep_hexagon_row_span(p, QX, QY, y, left, right);
                if (right > left) {
                    uint32_t at = atomicAdd(&q.n, (uint32_t)(right - left));
                    for (int x = left; x < right && at < kRecordTexels; ++x, ++at) q.rec[at] = (uint32_t)x | ((uint32_t)y << 16);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (dealt) {
            const LineStage &q = stage[threadIdx.x];
            r.n = q.n;
#pragma unroll
            for (uint32_t k = 0; k < kRecordTexels; ++k) r.r[k] = q.rec[k];
        } else {                                            // (the last phase reads every line's record from there)
            LineStage &q = stage[threadIdx.x];
#pragma unroll
            for (uint32_t k = 0; k < kRecordTexels; ++k) q.rec[k] = r.r[k];
        }
    }
    const bool lengthy = r.n > kRecordTexels;                // more fragments than a record holds: the long list
    if (lengthy) r.n = 0u;
    const uint32_t n = r.n;
    const LineBins q = line_bins(p, r.r, n);
    uint32_t took0 = 0, took1 = 0;
    if (n) took0 = resv_take(t, q.b0, q.c0);
    if (q.c1) took1 = resv_take(t, q.b1, q.c1);
    dep_list_append(p, kListSlow, block, slow, s);
    dep_list_append(p, kListLong, block, lengthy, s);
    __syncthreads();
    // the block's share of every bin it met: one atomic each on the cursor of the block's list of that bin; the
    // pages that start inside it are taken from the pool
    const uint32_t rep = block & (kBinReplicas - 1u);
    {
        // (a thread's atomics go out together: one round trip to the memory side, not one per entry)
        constexpr uint32_t kPer = kTab / BS;
        uint32_t tag[kPer], m[kPer], base[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) { tag[q] = t.tag[q * BS + threadIdx.x]; m[q] = t.sum[q * BS + threadIdx.x]; }
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) base[q] = tag[q] ? atomicAdd(list_cursor(p, tag[q] - 1u, rep), m[q]) : 0u;
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            if (tag[q] == 0u) continue;
            t.sum[q * BS + threadIdx.x] = base[q];
            if (base[q] + m[q] < base[q]) bins_flag(p, kBinsBinFull);
            else if (((base[q] + m[q] - 1u) >> kPageShift) != (base[q] >> kPageShift) || (base[q] & (kBinPage - 1u)) == 0u)
                pages_open(p, (tag[q] - 1u) * kBinReplicas + rep, base[q], m[q]);
        }
    }
    __syncthreads();
    if (n) {
        // a line's <= 8 places of a bin lie in one page or two: both looked up once, all lookups in flight together
        const uint32_t v0 = t.sum[took0 >> 20] + (took0 & 0xfffffu), v1 = q.c1 ? t.sum[took1 >> 20] + (took1 & 0xfffffu) : 0u;
        const uint32_t l0 = q.b0 * kBinReplicas + rep, l1 = (q.c1 ? q.b1 : q.b0) * kBinReplicas + rep;
        const uint32_t pa0 = v0 >> kPageShift, pb0 = (v0 + q.c0 - 1u) >> kPageShift, pa1 = v1 >> kPageShift, pb1 = (v1 + (q.c1 ? q.c1 - 1u : 0u)) >> kPageShift;
        const uint32_t ga0 = page_of<true>(p, l0, pa0), gb0 = pb0 != pa0 ? page_of<true>(p, l0, pb0) : ga0;
        const uint32_t ga1 = q.c1 ? page_of<true>(p, l1, pa1) : 0u, gb1 = (q.c1 && pb1 != pa1) ? page_of<true>(p, l1, pb1) : ga1;
        const uint32_t id = col * p.H + p.row0 + row;
        // (the line's texels read again - from the caches - rather than kept through the reservations: the kernel is short of
        // registers, and kept as an array they had gone to scratch memory)
        if constexpr (PLAIN) {
            const float4 again_cur = p.cur[s], again_prev = p.prev[s];
            auto texel = [&](bool c) { return make_float4(c ? again_cur.x : again_prev.x, c ? again_cur.y : again_prev.y, c ? again_cur.z : again_prev.z, c ? again_cur.w : again_prev.w); };
            dep_vertex_colors(p, texel(L.a.from_cur), L.a);
            dep_vertex_colors(p, texel(L.b.from_cur), L.b);
        } else {            // (whichever particles the vertices are: fetched as the set-up fetched them, with their varyings this time)
            L.a = dep_fetch(p, col, 2u * (p.row0 + row), row, s);
            L.b = dep_fetch(p, col, 2u * (p.row0 + row) + 1u, row, s);
        }
        // (one fragment at a time, the record read back from the line's own words of the stage: eight fragments' varyings side
        // by side were the kernel's register peak)
        uint32_t i0 = 0, i1 = 0;
        const uint32_t *rec = stage[threadIdx.x].rec;
#pragma unroll 1
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t xy = rec[k], x = xy & 0xffffu, y = xy >> 16, b = bin_of(p, x, y);
            uint32_t at;
            if (b == q.b0) { const uint32_t v = v0 + i0++, g = (v >> kPageShift) == pa0 ? ga0 : gb0; at = g == kNoPlace ? kNoPlace : (g << kPageShift) | (v & (kBinPage - 1u)); }
            else if (b == q.b1) { const uint32_t v = v1 + i1++, g = (v >> kPageShift) == pa1 ? ga1 : gb1; at = g == kNoPlace ? kNoPlace : (g << kPageShift) | (v & (kBinPage - 1u)); }
            else at = place_single(p, b, rep);
            bins_put(p, L, id, at, (int)x, (int)y);
        }
    }
}

// ... and the lines of the slow list (hexagons that cross the view's edge or need 64-bit edges: clipped, scan-converted with
// run-time indexed edges): one place at a time from the lists' cursors.  (Counting a line's fragments per bin first and
// reserving them together - two rasterisations, one round trip - was slower with the polygon in scratch memory, 115 against
// 92 us, and is no faster with it in LDS, 58 against 54: the pass is bound by the general rasteriser, not by its atomics.)
TH_D void bins_slow_lines(const DepositParams &p, uint32_t block, uint32_t blocks, float *polygons)
{
    LdsWords<256> words{polygons + threadIdx.x};          // (the clipped polygon: indexed at run time - in LDS, not in scratch memory)
    dep_list_work(p, kListSlow, [&](bool have, uint32_t s, uint32_t seg) {
        const uint32_t rep = seg & (kBinReplicas - 1u);
        // a slow line that SPANS the view - tens, hundreds of rows or columns - is a wave's work, not a lane's: passed on to
        // bins_span_kernel (asked here, of the few slow lines, not on the other sixteen million lines' path)
        bool spans = false;
        DepositLine L;
        uint32_t id = 0;
        if (have) {
            uint32_t col, row;
            slot_particle(p, s, col, row);
            dep_setup(p, col, p.row0 + row, L, s);
            id = col * p.H + p.row0 + row;
            if (L.draws) { float cx[6], cy[6]; dep_hexagon(p, L, cx, cy); spans = dep_hexagon_spans(p, cx, cy); }
        }
        dep_list_append(p, kListSpan, seg, spans, s);
        if (have && !spans)
            dep_raster_line(p, L, [&](int x, int y) {
                bins_put(p, L, id, place_single(p, bin_of(p, (uint32_t)x, (uint32_t)y), rep), x, y);
            }, words);
    }, block, blocks);
}

// ... and the lines of the long list: small hexagons inside the view like the rest, only with more fragments than a record
// holds - the same register-resident rasteriser, every fragment straight to a place of its own
TH_D void bins_long_lines(const DepositParams &p, uint32_t block, uint32_t blocks)
{
    dep_list_work(p, kListLong, [&](bool have, uint32_t s, uint32_t seg) {
        const uint32_t rep = seg & (kBinReplicas - 1u);
        if (have) {
            uint32_t col, row;
            slot_particle(p, s, col, row);
            DepositLine L;
            dep_setup(p, col, p.row0 + row, L, s);
            const uint32_t id = col * p.H + p.row0 + row;
            float cx[6], cy[6];
            int PX[6], PY[6], ymin, ymax;
            dep_hexagon(p, L, cx, cy);                       // (inside and small: the fused pass said so)
            dep_snap_hexagon(p, cx, cy, PX, PY);
            dep_hexagon_is_small(PX, PY, ymin, ymax);
            // Rasterised twice - the register rasteriser is cheap, a round trip to a list's cursor is not, and a fragment at a
            // time a line of twenty fragments made twenty of them one after the other (this kernel stands between the fused
            // pass and the plan, alone on the chip): first its fragments counted per bin (up to four bins - a line of ten
            // texels meets no more; further ones take their places one by one), their places reserved together, then written.
            uint32_t bin[4] = {kNoPlace, kNoPlace, kNoPlace, kNoPlace}, cnt[4] = {0u, 0u, 0u, 0u};
            dep_raster_small_hexagon2(p, PX, PY, ymin, ymax, [&](int x, int y) {
                const uint32_t b = bin_of(p, (uint32_t)x, (uint32_t)y);
                bool placed = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!placed && (bin[j] == b || bin[j] == kNoPlace)) { bin[j] = b; ++cnt[j]; placed = true; }
                }
            });
            uint32_t base[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) base[j] = cnt[j] ? atomicAdd(list_cursor(p, bin[j], rep), cnt[j]) : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (cnt[j] == 0u) continue;
                if (base[j] + cnt[j] < base[j]) { bins_flag(p, kBinsBinFull); cnt[j] = 0u; bin[j] = kNoPlace - 1u; }        // (nothing of it is written: the pass is repeated)
                else if (((base[j] + cnt[j] - 1u) >> kPageShift) != (base[j] >> kPageShift) || (base[j] & (kBinPage - 1u)) == 0u)
                    pages_open(p, bin[j] * kBinReplicas + rep, base[j], cnt[j]);
            }
            // (a bin's places of a line lie in one page or two: the page looked up when a place leaves the one before - on a crowded
            // target every place lies beyond its list's first page, and a device-scope load per fragment, twenty one after the
            // other, was the longest chain of the kernel)
            uint32_t seen_pn[4] = {kNoPlace, kNoPlace, kNoPlace, kNoPlace}, seen_page[4] = {0u, 0u, 0u, 0u};
            dep_raster_small_hexagon2(p, PX, PY, ymin, ymax, [&](int x, int y) {
                const uint32_t b = bin_of(p, (uint32_t)x, (uint32_t)y);
                uint32_t at = kNoPlace;
                bool placed = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!placed && bin[j] == b) {
                        const uint32_t v = base[j]++, pn = v >> kPageShift;
                        if (pn != seen_pn[j]) { seen_page[j] = page_of<true>(p, b * kBinReplicas + rep, pn); seen_pn[j] = pn; }
                        at = seen_page[j] == kNoPlace ? kNoPlace : (seen_page[j] << kPageShift) | (v & (kBinPage - 1u));
                        placed = true;
                    }
                }
                if (!placed) at = place_single(p, b, rep);
                bins_put(p, L, id, at, x, y);
            });
        }
    }, block, blocks);
}

// ... and the lines that SPAN the view (kListSpan): A WAVE PER LINE.  Where the vertex lookup of the texture's shape drifts
// off a line's own texel (th_order.hip: line_rows - every texture of 8192 and more, heights such as 100 and 1080) the lines of
// those rows and columns join two UNRELATED particles: tens of thousands of lines hundreds of texels long, as the reference's
// GL would draw them.  A lane walking such a line alone - the slow list's way - takes its thousand places one by one while the
// 63 other lanes of its wave walk theirs in lockstep, every one in its own branch: 15 ms per draw at 8192 x 8192 particles.
// Here the wave's lanes take the polygon's ROWS (64 at a time: the span of each, the same edge arithmetic in the same vertex
// order as dep_raster_poly), the spans' texels are dealt evenly to the lanes (a prefix sum over the rows, a search per lane),
// and the places of the lanes that meet in one bin are reserved with ONE atomic on that bin's cursor.
TH_D void bins_span_lines(const DepositParams &p, uint32_t block, uint32_t blocks, float *scratch)
{
    const uint32_t wave = threadIdx.x >> 6, lane = __lane_id();
    float *words = scratch + wave * 128u;                // per wave: the clipped polygon (48 words), then 64 row starts
    uint32_t *starts = reinterpret_cast<uint32_t *>(words + 48);
    LdsWords<1> w{words};                                // (every lane of the wave computes the same polygon into the same words)
    const uint32_t seg = block & (kDepLists - 1u);
    const uint32_t n = p.list_n[(kListSpan * kDepLists + seg) * kDepListStride];
    const uint32_t *list = p.lists + ((size_t)kListSpan * kDepLists + seg) * p.list_cap;
    const uint32_t rep = seg & (kBinReplicas - 1u);
    // the segment's lines are TAKEN, one at a time, by whichever wave of the segment's workgroups is free (the word beside the
    // segment's counter, zero when the pass starts): a line is a few dozen fragments or a few thousand, and dealt out in turn the
    // launch lasted as long as the wave with the longest lines (bins_span_kernel 803 -> us at 8192 x 8192 particles)
    uint32_t *next = &p.list_n[(kListSpan * kDepLists + seg) * kDepListStride + 1u];
    (void)blocks;
    for (;;) {
        uint32_t e = 0;
        if (lane == 0u) e = atomicAdd(next, 1u);
        e = (uint32_t)__shfl((int)e, 0);
        if (e >= n) break;
        const uint32_t s = list[e];
        uint32_t col, row;
        slot_particle(p, s, col, row);
        DepositLine L;
        dep_setup(p, col, p.row0 + row, L, s);
        const uint32_t id = col * p.H + p.row0 + row;
        if (!L.draws) continue;                          // (wave-uniform: every lane set the same line up)
        float cx[6], cy[6];
        const int where = dep_hexagon(p, L, cx, cy);
        int nv = 0;
        if (where == kHexInside) {
            int PX[6], PY[6];
            dep_snap_hexagon(p, cx, cy, PX, PY);
#pragma unroll
            for (int k = 0; k < 6; ++k) { w.i(24 + k) = PX[k]; w.i(36 + k) = PY[k]; }
            nv = 6;
        } else if (where == kHexClip) {
            dep_clip_hexagon(p, L, cx, cy, w);
            nv = L.draws ? L.n : 0;
        }
        wave_sync_lds();
        if (nv < 3) continue;
        int ymin = w.i(36), ymax = w.i(36);
        for (int k = 1; k < nv; ++k) { const int y = w.i(36 + k); ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax; }
        int r0 = (ymin + 15) >> 4, r1 = (ymax + 15) >> 4;
        r0 = r0 < 0 ? 0 : r0; r1 = r1 > p.fh ? p.fh : r1;
        for (int base = r0; base < r1; base += 64) {
            // this lane's row: its span, edge by edge in vertex order - a later edge over an earlier one, as dep_raster_poly has it
            const int y = base + (int)lane;
            int left = p.fw, right = 0;
            if (y < r1)
                for (int k = 0; k < nv; ++k) {
                    const int kn = k + 1 == nv ? 0 : k + 1;
                    const int Xa = w.i(24 + k), Ya = w.i(36 + k), Xb = w.i(24 + kn), Yb = w.i(36 + kn);
                    if (Ya == Yb) continue;
                    const bool swap = Yb < Ya;
                    const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
                    if (y < ((Y1 + 15) >> 4) || y >= ((Y2 + 15) >> 4)) continue;
                    const long long DX = X2 - X1, DY = Y2 - Y1;
                    // ceil(num / den), den > 0, |num| < 2^53: the quotient estimated in fp64 (within one of the floor), the remainder
                    // decides - dep_ceil_div's integer, without a 64-bit integer division per edge and row (software: hundreds of
                    // dependent instructions, most of this kernel's arithmetic)
                    const long long num = DX * (((long long)y << 4) - Y1) + (long long)X1 * DY, den = 16 * DY;
                    long long q = (long long)__builtin_floor((double)num / (double)den), rem = num - q * den;
                    if (rem < 0) { --q; rem += den; }
                    if (rem >= den) { ++q; rem -= den; }
                    long long x = rem > 0 ? q + 1 : q;
                    x = x < 0 ? 0 : (x > p.fw ? p.fw : x);
                    if (swap) right = (int)x; else left = (int)x;
                }
            const uint32_t len = right > left ? (uint32_t)(right - left) : 0u;
            uint32_t incl = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)lane >= o) incl += up; }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63);
            starts[lane] = incl - len;
            wave_sync_lds();
            for (uint32_t f0 = 0; f0 < total; f0 += 64u) {
                const uint32_t f = f0 + lane;
                const bool have = f < total;
                // the row fragment f lies in: the last lane whose rows start at or before it (a search over the wave's 64 starts)
                uint32_t j = 0;
#pragma unroll
                for (uint32_t step = 32u; step > 0u; step >>= 1) if (starts[j + step] <= (have ? f : 0u)) j += step;
                const int x = __shfl(left, (int)j) + (int)((have ? f : 0u) - starts[j]), yy = base + (int)j;
                const uint32_t b = have ? bin_of(p, (uint32_t)x, (uint32_t)yy) : kNoPlace;
                // one reservation per bin the 64 fragments meet (a line passes from bin to bin: a handful) - the lanes are grouped by
                // bin first (ballots only), then every group's first lane moves its bin's cursor: the groups' atomics go out TOGETHER,
                // one round trip to the memory side per 64 fragments instead of one per bin met
                uint32_t leader = lane, rank = 0, cnt = 0;
                unsigned long long left_over = __ballot(have);
                while (left_over) {
                    const int first_lane = __builtin_ctzll(left_over);
                    const uint32_t lb = (uint32_t)__shfl((int)b, first_lane);
                    const unsigned long long same = __ballot(have && b == lb) & left_over;
                    if (same >> lane & 1ull) {
                        leader = (uint32_t)first_lane;
                        rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
                        cnt = (uint32_t)__builtin_popcountll(same);
                    }
                    left_over &= ~same;
                }
                uint32_t first = 0;
                if (have && leader == lane) {
                    first = atomicAdd(list_cursor(p, b, rep), cnt);
                    if (first + cnt < first) { bins_flag(p, kBinsBinFull); first = 0xffffffffu - cnt; }
                    else if (((first + cnt - 1u) >> kPageShift) != (first >> kPageShift) || (first & (kBinPage - 1u)) == 0u)
                        pages_open(p, b * kBinReplicas + rep, first, cnt);
                }
                const uint32_t v = (uint32_t)__shfl((int)first, (int)leader) + rank;
                if (have) bins_put(p, L, id, place_of<true>(p, b * kBinReplicas + rep, v), x, yy);       // (a list that overflowed: no place, the pass is repeated)
            }
            wave_sync_lds();                             // (the starts are the next 64 rows' now)
        }
        wave_sync_lds();                                 // (... and the polygon's words the next line's)
    }
}

// both lists in one launch, half of the grid each: two small grids that wait on their loads and atomics, side by side
__global__ __launch_bounds__(256) void bins_listed_kernel(const DepositParams p)
{
    __shared__ float polygons[48 * 256];                  // (48 KB: three workgroups per CU - the grid is that large)
    const uint32_t half = gridDim.x >> 1;
    if (blockIdx.x < half) bins_long_lines(p, blockIdx.x, half);
    else bins_slow_lines(p, blockIdx.x - half, half, polygons);
}
// ... and behind it the spanning lines the slow list's pass found (none on an ordinary frame: the workgroups read a zero and
// leave - a launch of ~3 us in the draw's chain, where classifying every line in the emit cost that kernel 3 %: 473 -> 487 us)
__global__ __launch_bounds__(256) void bins_span_kernel(const DepositParams p)
{
    __shared__ float scratch[4 * 128];
    bins_span_lines(p, blockIdx.x, gridDim.x, scratch);
}

// the places handed out in bin b (all its lists; saturated)
TH_D uint32_t bin_places(const DepositParams &p, uint32_t b)
{
    unsigned long long n = 0;
#pragma unroll
    for (uint32_t r = 0; r < kBinReplicas; ++r) n += *list_cursor(p, b, r);
    return n > 0xffffffffull ? 0xffffffffu : (uint32_t)n;
}

// pass 2: the bins of more places than one workgroup orders in LDS; the fragments of the pass (every place handed out holds
// one: no hot counter in the pass itself)
__global__ __launch_bounds__(256) void bins_plan_kernel(const DepositParams p)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = b < p.nbins ? bin_places(p, b) : 0u;
    if (n > kBinCap) p.large_bins[atomicAdd(&p.totals[kTotLarge], 1u)] = b;       // (in whatever order)
    unsigned long long sum = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (__lane_id() == 0u && sum) {
        const unsigned long long cap = 0xffffffffull;
        const uint32_t old = atomicAdd(&p.totals[kTotFragments], (uint32_t)(sum > cap ? cap : sum));
        if ((unsigned long long)old + sum > cap) bins_flag(p, kBinsBinFull);
    }
}

// ---- pass 3: one workgroup per bin -----------------------------------------------------------------------------
TH_D uint32_t key_local(unsigned long long k) { return (uint32_t)((k >> 40) & 0xf0u) | (uint32_t)((k >> 32) & 0xfu); }   // (y & 15) << 4 | (x & 15)
// sort key of a fragment while a crowded bin is ordered: texel inside the bin (8 bits) | stream index (32) | position in the
// bin's range of the fragment array (24 bits: where its varying lies)
TH_D unsigned long long sort_key(unsigned long long k, uint32_t f) { return ((unsigned long long)key_local(k) << 56) | ((k & 0xffffffffull) << 24) | f; }

// the destination texel(s) of one thread, in registers while the runs are blended.  MODE 0: the flow texture, 1: the view
// buffer, 2: both (the fragments carry two varyings side by side)
template <int MODE>
struct BinTexel {
    float4 f;
    float4 v;           // (the RGBA8 texel unpacked: ViewTarget::apply_unpacked)
    TH_D void load(const DepositParams &p, uint32_t texel)
    {
        if constexpr (MODE != 1) f = p.flow[texel];
        if constexpr (MODE != 0) v = ViewTarget::unpack(p.view[texel]);
    }
    TH_D void store(const DepositParams &p, uint32_t texel) const
    {
        if constexpr (MODE != 1) p.flow[texel] = f;
        if constexpr (MODE != 0) p.view[texel] = ViewTarget::pack(v);
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
    else if constexpr (MODE == 1) ViewTarget::apply_unpacked(d.v, ViewTarget::source(c0));
    else { FlowTarget::apply(d.f, FlowTarget::source(c0)); ViewTarget::apply_unpacked(d.v, ViewTarget::source(c1)); }
}

// LDS of a bin's workgroup.  `pool` is used two ways:
//   the common case (a bin of <= kBinCap fragments, no run longer than kRankMaxRun): sid = stream index of every fragment,
//     grouped by texel, and then - once every fragment knows its rank - in the same words osrc = the fragments' positions
//     (where their varyings lie) in blend order                                                            (kBinCap words)
//   crowded bins: skey = 64-bit sort keys of a batch of <= kCrowdCap fragments | order = blend order as indices into skey |
//     hist = the id histogram of a texel too crowded for one batch                                  (3 x kCrowdCap words)
// (24 KB + the stages: five workgroups per CU for one target, four for both - the pass waits on its loads, not on arithmetic)
constexpr uint32_t kCrowdCap = 2048;
template <int MODE>
struct BinShared {
    uint32_t pool[3u * kCrowdCap];
    uint32_t cnt[kBinTexels], first[kBinTexels + 1u];        // fragments per texel (then: fill cursors); first fragment of every texel
    BlendSource stage_a[256], stage_b[MODE == 2 ? 256 : 1];  // a long run's sources, 256 at a time (b: the view pass's beside the flow pass's)
    uint32_t misc[8];
    float chan[8];                                           // a long run's destination, a channel per lane (bin_blend_long)
    uint32_t lists[kBinReplicas + 1u];                       // first place of every list of the bin when its lists are walked one after the other
    TH_D uint32_t *sid() { return pool; }
    TH_D uint32_t *osrc() { return pool; }
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
template <int MODE, typename SrcAt, uint32_t AHEAD = (MODE == 2 ? 4u : 8u)>
TH_D void bin_blend_own(const DepositParams &p, uint32_t begin, uint32_t len, BinTexel<MODE> &d, SrcAt src_at)
{
    constexpr uint32_t kAhead = AHEAD;
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

// A run of one texel walked by FOUR lanes, a channel each (of the flow texel and / or of the view texel; c = the lane's
// channel): the same operations per channel in the same order as a thread doing all channels, a quarter of the chain.
// The destination is read from and written to the target(s) directly.  While a batch is applied, the next batch's varyings
// and the positions of the batch after it are in flight.
template <int MODE, typename SrcAt>
TH_D void quad_walk(const DepositParams &p, uint32_t texel, uint32_t len, uint32_t c, SrcAt src_at)
{
    float f = 0.0f, v = 0.0f;                                 // this lane's channel of the two destinations
    if constexpr (MODE != 1) f = reinterpret_cast<const float *>(p.flow + texel)[c];
    if constexpr (MODE != 0) v = (float)reinterpret_cast<const unsigned char *>(p.view + texel)[c];
    const float *colors = reinterpret_cast<const float *>(p.colors);
    constexpr uint32_t kAhead = 8u, kFloats = MODE == 2 ? 8u : 4u;
    struct Piece { float c0, a0, c1, a1; };                   // the lane's component and the alpha of the fragment's varying(s)
    // Both varyings (32 B): ONE 8-byte load a lane - floats 2c and 2c + 1 - and the quad hands the components round (six moves
    // inside the quad and two selects) instead of four 4-byte loads a lane: a quarter of the load instructions of a kernel
    // that is all loads.
    struct Raw { float x, y, z, w; };
    auto fetch = [&](uint32_t place) {
        const float *at = colors + (size_t)place * kFloats;
        Raw q{};
        if constexpr (MODE == 2) { const float2 v = *reinterpret_cast<const float2 *>(at + 2u * c); q.x = v.x; q.y = v.y; }
        else { q.x = at[c]; q.y = at[3]; }
        return q;
    };
    auto quad = [](float v, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true)); };
    auto unpack = [&](const Raw &r) {
        Piece q{};
        if constexpr (MODE == 2) {
            const bool odd = (c & 1u) != 0u;
            const float lx = quad(r.x, std::integral_constant<int, 0x50>{}), ly = quad(r.y, std::integral_constant<int, 0x50>{});      // lanes 0 0 1 1: floats 0 1 | 2 3
            const float hx = quad(r.x, std::integral_constant<int, 0xfa>{}), hy = quad(r.y, std::integral_constant<int, 0xfa>{});      // lanes 2 2 3 3: floats 4 5 | 6 7
            q.c0 = odd ? ly : lx; q.c1 = odd ? hy : hx;
            q.a0 = quad(r.y, std::integral_constant<int, 0x55>{});             // lane 1's second float: 3
            q.a1 = quad(r.y, std::integral_constant<int, 0xff>{});             // lane 3's second float: 7
        } else { q.c0 = r.x; q.a0 = r.y; }
        return q;
    };
    auto apply = [&](const Raw &raw) {
        const Piece q = unpack(raw);
        if constexpr (MODE != 1) {                            // FlowTarget::source + apply, one channel
            const float sa = q.a0;
            FlowTarget::apply_channel(f, q.c0 * sa, 1.0f - sa);
        }
        if constexpr (MODE != 0) {                            // ViewTarget::source + apply_unpacked, one channel
            const float cc = MODE == 1 ? q.c0 : q.c1, ca = MODE == 1 ? q.a0 : q.a1;
            const float col = __builtin_fminf(__builtin_fmaxf(cc, 0.0f), 1.0f), sa = __builtin_fminf(__builtin_fmaxf(ca, 0.0f), 1.0f);
            ViewTarget::apply_channel(v, col * sa, 1.0f - sa);
        }
    };
    uint32_t src[kAhead];
    Raw cur[kAhead];
#pragma unroll
    for (uint32_t q = 0; q < kAhead; ++q) src[q] = src_at(q < len ? q : len - 1u);
#pragma unroll
    for (uint32_t q = 0; q < kAhead; ++q) cur[q] = fetch(src[q]);
#pragma unroll
    for (uint32_t q = 0; q < kAhead; ++q) src[q] = src_at(kAhead + q < len ? kAhead + q : len - 1u);
    for (uint32_t j0 = 0; j0 < len; j0 += kAhead) {
        Raw nxt[kAhead];
        uint32_t after[kAhead];
#pragma unroll
        for (uint32_t q = 0; q < kAhead; ++q) nxt[q] = fetch(src[q]);
#pragma unroll
        for (uint32_t q = 0; q < kAhead; ++q) { const uint32_t j = j0 + 2u * kAhead + q; after[q] = src_at(j < len ? j : len - 1u); }
#pragma unroll
        for (uint32_t q = 0; q < kAhead; ++q) if (j0 + q < len) apply(cur[q]);
#pragma unroll
        for (uint32_t q = 0; q < kAhead; ++q) { cur[q] = nxt[q]; src[q] = after[q]; }
    }
    if constexpr (MODE != 1) reinterpret_cast<float *>(p.flow + texel)[c] = f;
    if constexpr (MODE != 0) reinterpret_cast<unsigned char *>(p.view + texel)[c] = (unsigned char)v;
}

// The chain of one staged batch: n sources - {x, y, z, w, 1 - alpha} side by side from `from` on - applied to this lane's
// channel `ch` of the destination, one after the other (read eight ahead of the dependent blends).  The loop holds ONE target's
// arithmetic: with the flow texel's lanes and the view texel's lanes choosing theirs fragment by fragment, a wave spent more
// on its execution mask than on the blends (profiles/r4_g_giants.txt).
template <bool VIEW>
TH_D void apply_batch(float &comp, const BlendSource *from, uint32_t ch, uint32_t n)
{
    const float *mine = reinterpret_cast<const float *>(from) + ch, *das = reinterpret_cast<const float *>(from) + 4u;
    auto apply = [&](float src, float da) {
        if constexpr (VIEW) ViewTarget::apply_channel(comp, src, da);
        else FlowTarget::apply_channel(comp, src, da);
    };
    uint32_t q = 0;
    for (; q + 8u <= n; q += 8u) {
        float sv[8], da[8];
#pragma unroll
        for (uint32_t e = 0; e < 8u; ++e) { sv[e] = mine[(q + e) * 5u]; da[e] = das[(q + e) * 5u]; }
#pragma unroll
        for (uint32_t e = 0; e < 8u; ++e) apply(sv[e], da[e]);
    }
    for (; q < n; ++q) apply(mine[q * 5u], das[q * 5u]);
}

// a long run by the whole workgroup: every thread turns one fragment's varying into its side of the blend (256 loads in
// flight); then the destination's CHANNELS are applied side by side, a lane each - lanes 0-3 of the workgroup the flow
// texel's four floats, lanes 4-7 the view texel's four bytes (as integer-valued floats) - instead of one thread doing all
// eight one after the other: the same operations on every channel in the same order, a quarter of the instructions per
// fragment in the one wave that everything else waits for (the texel's thread hands its value over through LDS and takes
// it back at the end)
template <int MODE, typename SrcAt>
TH_D void bin_blend_long(BinShared<MODE> &s, const DepositParams &p, uint32_t begin, uint32_t len, uint32_t owner, BinTexel<MODE> &d, SrcAt src_at)
{
    const uint32_t t = threadIdx.x;
    const bool flow_lane = t < 4u;
    const bool channel = t < 8u && (MODE == 2 || (MODE == 0) == flow_lane);
    if (t == owner) {
        if constexpr (MODE != 1) { s.chan[0] = d.f.x; s.chan[1] = d.f.y; s.chan[2] = d.f.z; s.chan[3] = d.f.w; }
        if constexpr (MODE != 0) { s.chan[4] = d.v.x; s.chan[5] = d.v.y; s.chan[6] = d.v.z; s.chan[7] = d.v.w; }
    }
    float comp = 0.0f;
    for (uint32_t j0 = 0; j0 < len; j0 += 256u) {
        const uint32_t j = j0 + t;
        if (j < len) {
            float4 c0, c1;
            fetch_colors<MODE>(p, (size_t)begin + src_at(j), c0, c1);
            if constexpr (MODE == 1) s.stage_a[t] = ViewTarget::source(c0);
            else s.stage_a[t] = FlowTarget::source(c0);
            if constexpr (MODE == 2) s.stage_b[t] = ViewTarget::source(c1);
        }
        __syncthreads();
        if (channel) {
            if (j0 == 0u) comp = s.chan[t];
            // this lane's component of every source, and the source's 1 - alpha: {x, y, z, w, da} lie side by side
            const uint32_t n = len - j0 < 256u ? len - j0 : 256u;
            if (MODE != 1 && flow_lane) apply_batch<false>(comp, s.stage_a, t & 3u, n);
            else apply_batch<true>(comp, MODE == 2 ? s.stage_b : s.stage_a, t & 3u, n);
            if (j0 + 256u >= len) s.chan[t] = comp;
        }
        __syncthreads();
    }
    if (t == owner) {
        if constexpr (MODE != 1) d.f = make_float4(s.chan[0], s.chan[1], s.chan[2], s.chan[3]);
        if constexpr (MODE != 0) d.v = make_float4(s.chan[4], s.chan[5], s.chan[6], s.chan[7]);
    }
    __syncthreads();            // (the words are free for the next run's texel)
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

// every fragment among n places (place(f) = where place f of the walk lies): body(key, f) (empty places skipped); a
// thread's loads of one round are issued together (unconditional, clamped)
template <typename PlaceAt, typename Body>
TH_D void for_keys(const unsigned long long *keys, PlaceAt place, uint32_t n, Body body)
{
    constexpr uint32_t kPer = 8;
    for (uint32_t c0 = 0; c0 < n; c0 += kPer * 256u) {
        unsigned long long k[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = c0 + q * 256u + threadIdx.x; k[q] = keys[place(f < n ? f : n - 1u)]; }
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = c0 + q * 256u + threadIdx.x; if (f < n && k[q] != kEmptyKey) body(k[q], f); }
    }
}

// One texel's fragments when they do not fit LDS at once: windows [lo, hi) of its stream indices (distinct inside one
// texel: a line covers a texel at most once), each small enough to be sorted in LDS, in rising order; the texel's thread
// (`owner`) carries the destination from window to window.  for_texel(body) hands every fragment's sort key to body -
// stream index above the PB position bits, which place() turns into the place of the varying.
template <int MODE, int PB, typename PlaceAt, typename ForTexel>
TH_D void blend_texel_windows(BinShared<MODE> &s, const DepositParams &p, uint32_t owner, BinTexel<MODE> &d, PlaceAt place, ForTexel for_texel)
{
    const uint32_t t = threadIdx.x;
    unsigned long long *skey = s.skey();
    uint16_t *order = s.order();
    uint32_t *hist = s.hist();
    unsigned long long lo = 0ull;
    uint32_t id_bits = 1;                           // bits of a stream index: ceil(log2(W * H))
    while (id_bits < 32u && (1ull << id_bits) < (unsigned long long)p.W * p.H) ++id_bits;
    uint32_t shift = id_bits > 10u ? id_bits - 10u : 0u;
    while (true) {
        for (uint32_t k = t; k < 1024u; k += 256u) hist[k] = 0u;
        if (t < 3u) s.misc[t] = 0u;
        __syncthreads();
        for_texel([&](unsigned long long k) {
            const unsigned long long id = (k >> PB) & 0xffffffffull;
            if (id >= lo) {
                const unsigned long long bkt = (id - lo) >> shift;
                atomicAdd(&hist[bkt < 1023ull ? (uint32_t)bkt : 1023u], 1u);
            }
        });
        __syncthreads();
        {
            // the window: the leading buckets whose running total stays within kCrowdCap.  Every thread its four buckets,
            // a scan of the workgroup over them (one thread walking the 1024 words took ~30 us per window - of a chain
            // that is the draw's critical path)
            const uint32_t lane = t & 63u, wave = t >> 6;
            const uint32_t h0 = hist[4u * t], h1 = hist[4u * t + 1u], h2 = hist[4u * t + 2u], h3 = hist[4u * t + 3u];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)lane >= o) incl += up; }
            if (lane == 63u) s.misc[4u + wave] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (uint32_t w = 0; w < 4u; ++w) { const uint32_t v = s.misc[4u + w]; total += v; before += w < wave ? v : 0u; }
            const uint32_t a0 = before + incl - mine + h0, a1 = a0 + h1, a2 = a1 + h2, a3 = a2 + h3;
            const uint32_t within = (a0 <= kCrowdCap ? 1u : 0u) + (a1 <= kCrowdCap ? 1u : 0u) + (a2 <= kCrowdCap ? 1u : 0u) + (a3 <= kCrowdCap ? 1u : 0u);
            const uint32_t reach = a3 <= kCrowdCap ? a3 : (a2 <= kCrowdCap ? a2 : (a1 <= kCrowdCap ? a1 : (a0 <= kCrowdCap ? a0 : 0u)));
            if (within) { atomicAdd(&s.misc[1], within); atomicMax(&s.misc[2], reach); }       // (the running total never falls: the buckets within the cap are the leading ones)
            if (t == 0u) s.misc[3] = total;
        }
        __syncthreads();
        const uint32_t nb = s.misc[1], m = s.misc[2], rest = s.misc[3] - m;
        __syncthreads();
        if (nb == 0u) { shift = shift > 4u ? shift - 4u : 0u; continue; }     // the first bucket alone is too large: finer buckets
        // (bucket 1023 also holds everything beyond it: taken only together with all the others = the rest of the run)
        const unsigned long long hi = nb == 1024u ? 0x100000000ull : lo + ((unsigned long long)nb << shift);
        for_texel([&](unsigned long long k) {
            const unsigned long long id = (k >> PB) & 0xffffffffull;
            if (id >= lo && id < hi) skey[atomicAdd(&s.misc[0], 1u)] = k;
        });
        __syncthreads();
        bin_bitonic(s, m);
        bin_blend_long<MODE>(s, p, 0u, m, owner, d, [&](uint32_t j) { return place((uint32_t)(skey[order[j]] & ((1ull << PB) - 1ull))); });
        if (rest == 0u) break;
        lo = hi;
    }
}

// A bin of one chunk with a run longer than kRankMaxRun (bins of more chunks are spread over many workgroups:
// crowd_*_kernel below): batches of whole texels that fit kCrowdCap sort keys, each ordered in LDS; a texel
// that does not fit alone goes in windows of its stream indices.
template <int MODE, typename PlaceAt>
TH_D void bin_crowded(BinShared<MODE> &s, const DepositParams &p, const unsigned long long *keys, PlaceAt place, uint32_t n,
                      BinTexel<MODE> &d, bool &touched)
{
    const uint32_t t = threadIdx.x;
    constexpr uint32_t begin = 0u;                 // (the sources are places)
    unsigned long long *skey = s.skey();
    uint16_t *order = s.order();
    auto src_at = [&](uint32_t base) { return [&, base](uint32_t j) { return place((uint32_t)(skey[order[base + j]] & 0xffffffull)); }; };
    s.cnt[t] = 0u;
    __syncthreads();
    for_keys(keys, place, n, [&](unsigned long long k, uint32_t) { atomicAdd(&s.cnt[key_local(k)], 1u); });
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
                for_keys(keys, place, n, [&](unsigned long long k, uint32_t f) {
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
        blend_texel_windows<MODE, 24>(s, p, t0, d, place, [&](auto body) {
            for_keys(keys, place, n, [&](unsigned long long k, uint32_t f) { if (key_local(k) == t0) body(sort_key(k, f)); });
        });
        ++t0;
    }
}

#ifdef TH_BLEND_STAMPS
// (diagnostic builds only - tools/build_variant_libs.sh, tools/gpu_ab_prev.sh: where a bin's workgroup spends its life, cycles per phase summed over the launch)
__device__ unsigned long long g_blend_stamps[16];
#define TH_STAMP(k) do { if (threadIdx.x == 0u) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&g_blend_stamps[k], now_ - last_); last_ = now_; } } while (0)
#define TH_STAMP_BEGIN() unsigned long long last_ = __builtin_readcyclecounter()
#else
#define TH_STAMP(k) do {} while (0)
#define TH_STAMP_BEGIN() do {} while (0)
#endif

template <int MODE>
__global__ __launch_bounds__(256, MODE == 2 ? 4 : 5) void bins_blend_kernel(const DepositParams p)
{
    __shared__ BinShared<MODE> s;
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    TH_STAMP_BEGIN();
    // (the pass is launched before the host has seen the emitting pass's flags - its read-back hides under this kernel: a
    // pass that ran out of pages or of room in a bin is repeated before anything is blended)
    if (p.totals[kTotFlags] != 0u || p.totals[kTotOob] != 0u) return;
    if (t < 64u) {
        // the lists' cursors, a lane each, and their running sum across the lanes (one thread walking the sixteen: a tenth of the
        // workgroup's life went by before its first barrier)
        static_assert(kBinReplicas <= 64u, "a lane per list");
        unsigned long long incl = t < kBinReplicas ? *list_cursor(p, b, t) : 0u;
        const unsigned long long mine = incl;
#pragma unroll
        for (int o = 1; o < (int)kBinReplicas; o <<= 1) { const unsigned long long up = __shfl_up(incl, o); if ((int)t >= o) incl += up; }
        const unsigned long long cap = 0xffffffffull;
        if (t < kBinReplicas) s.lists[t] = (uint32_t)(incl - mine > cap ? cap : incl - mine);
        if (t == kBinReplicas - 1u) s.lists[kBinReplicas] = (uint32_t)(incl > cap ? cap : incl);
    }
    __syncthreads();
    const uint32_t n = s.lists[kBinReplicas];        // (places handed out: some may be empty)
    if (n == 0u || n > kBinCap) return;              // (bins of more places: crowd_*_kernel)
    TH_STAMP(0);
    // place f of the bin's lists walked one after the other
    auto place = [&](uint32_t f) {
        uint32_t r = 0;
#pragma unroll
        for (uint32_t q = 1; q < kBinReplicas; ++q) r += f >= s.lists[q] ? 1u : 0u;
        return place_of(p, b * kBinReplicas + r, f - s.lists[r]);
    };
    unsigned long long *keys = p.frag_keys;
    const uint32_t by = b / p.bins_x, bx = b - by * p.bins_x;
    const uint32_t x = (bx << kBinShift) + (t & (kBinSide - 1u)), y = (by << kBinShift) + (t >> kBinShift);
    const bool inside = x < (uint32_t)p.fw && y < (uint32_t)p.fh;
    const uint32_t texel = inside ? y * (uint32_t)p.fw + x : 0u;
    BinTexel<MODE> d{};
    if (inside) d.load(p, texel);
    bool touched = false;

    constexpr uint32_t kPer = kBinCap / 256u;
    uint32_t at[kPer];                               // where the thread's places lie
    bool crowded = false;
    {
        // The common case.  A thread's <= 16 keys stay in registers from the count to the ranking: counted per texel, the
        // runs laid out by the scan, every fragment's stream index dropped into its texel's run (in whatever order the LDS
        // atomics hand out), then every fragment ranks itself inside its run by counting the smaller indices - its place
        // in GL's order - and leaves the place of its varying there for the texel's thread.
        // (per fragment a thread keeps: the place of its varying, its stream index, its texel inside the bin - four to a word;
        // then its rank, likewise)
        uint32_t id[kPer], lts[kPer / 4u] = {}, ranks[kPer / 4u] = {}, have = 0u;
        {
            unsigned long long k[kPer];
            // (only the rounds the bin's places fill - seven of the sixteen on an ordinary first frame: a round beyond them made
            // its places up - fifteen comparisons against the lists' starts each - and loaded the last place's key again)
            const uint32_t rounds = (n + 255u) >> 8;
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = q * 256u + t; at[q] = q < rounds ? place(f < n ? f : n - 1u) : 0u; }
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q) k[q] = q < rounds ? keys[at[q]] : kEmptyKey;
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q) {
                id[q] = (uint32_t)k[q];
                lts[q >> 2] |= key_local(k[q]) << ((q & 3u) * 8u);
                have |= (q * 256u + t < n && k[q] != kEmptyKey) ? 1u << q : 0u;
            }
        }
        auto lt_of = [&](uint32_t q) { return (lts[q >> 2] >> ((q & 3u) * 8u)) & 0xffu; };
        s.cnt[t] = 0u;
        __syncthreads();
        TH_STAMP(1);
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) if (have >> q & 1u) atomicAdd(&s.cnt[lt_of(q)], 1u);
        __syncthreads();
        const uint32_t mine = s.cnt[t];
        const uint32_t longest = bin_scan_counts(s);
        crowded = longest > kRankMaxRun;
        TH_STAMP(2);
        if (!crowded) {
            s.cnt[t] = 0u;
            __syncthreads();
            uint32_t *sid = s.sid(), *osrc = s.osrc();
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q)
                if (have >> q & 1u) { const uint32_t lt = lt_of(q); sid[s.first[lt] + atomicAdd(&s.cnt[lt], 1u)] = id[q]; }
            __syncthreads();
            TH_STAMP(3);
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q)
                if (have >> q & 1u) {
                    const uint32_t lt = lt_of(q), r0 = s.first[lt], r1 = s.first[lt + 1u];
                    uint32_t rank = 0;                  // (< kRankMaxRun = 256: a byte)
                    // (four indices a trip from one 16-byte LDS read, the neighbours' masked out - what the crowded bins' sort gained
                    // from - was 50 us SLOWER here: runs of six, most of a read thrown away; profiles/r6_d_blend_experiments.txt)
                    for (uint32_t j = r0; j < r1; ++j) rank += sid[j] < id[q] ? 1u : 0u;
                    ranks[q >> 2] |= rank << ((q & 3u) * 8u);
                }
            __syncthreads();                            // (every stream index has been read: the words now take the places)
            TH_STAMP(4);
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q)
                if (have >> q & 1u) osrc[s.first[lt_of(q)] + ((ranks[q >> 2] >> ((q & 3u) * 8u)) & 0xffu)] = at[q];
            __syncthreads();
            TH_STAMP(5);
            const uint32_t r0 = s.first[t];
            auto src_at = [&](uint32_t base) { return [osrc, base](uint32_t j) { return osrc[base + j]; }; };
            // (eight fragments' varyings in flight per thread with both targets, sixteen with one: what the kernel's 122 / 96 registers
            // hold anyway - its peak is where the keys are read; profiles/r6_d_blend_experiments.txt)
            if (mine && mine <= kOwnRun) { touched = true; bin_blend_own<MODE, decltype(src_at(r0)), (MODE == 2 ? 8u : 16u)>(p, 0u, mine, d, src_at(r0)); }
#ifdef TH_BLEND_STAMPS
            TH_STAMP(6);                                 // (thread 0's own run)
            __syncthreads();
            TH_STAMP(7);                                 // (... and the wait for the workgroup's longest)
            if (threadIdx.x == 0u) { atomicAdd(&g_blend_stamps[12], 1ull); atomicAdd(&g_blend_stamps[13], (unsigned long long)n); atomicAdd(&g_blend_stamps[14], (unsigned long long)longest); }
#endif
            static_assert(kRankMaxRun <= kOwnRun, "a bin whose runs are ranked by counting has no run its texel's thread does not blend alone");
        }
    }
    if (crowded) { bin_crowded<MODE>(s, p, keys, place, n, d, touched); __syncthreads(); }
    if (inside && touched) d.store(p, texel);
    TH_STAMP(8);
    // the pages the bin's lists grew by are forgotten for the next pass
    for (uint32_t r = 0; r < kBinReplicas; ++r) if (s.lists[r + 1u] - s.lists[r] > kBinPage) pages_forget(p, b * kBinReplicas + r, s.lists[r + 1u] - s.lists[r], t, 256u);
}


// ---- bins of more than kBinCap places: many workgroups per bin ---------------------------------------------------------------
// The wake makes particles converge: after a few dozen frames a few hundred bins hold a third of all fragments (tens of
// thousands each, a thousand in single texels).  Those bins get one more level of the same scheme: their fragments are
// regrouped by TEXEL - a counting sort over the 256 texels of the bin, one workgroup per list of the bin, exact ranges from
// a per-bin scan - and then every texel's run is ordered by stream index and blended by a wave (or, the longest, a
// workgroup) of its own.
//   crowd_plan_kernel        regrouped-key ranges of the large bins (prefix over the list)
//   crowd_hist_kernel        fragments per texel of every large bin
//   crowd_scan_kernel        every texel's range inside its bin; the list of the long runs
//   crowd_scatter_kernel     (stream index << 32 | place of the varying) of every fragment into its texel's range; the
//                            places and the lists' pages are left empty
//   crowd_sort_kernel        one wave per texel: its run of up to kWaveRun fragments ordered (rank by counting)
//   crowd_walk_kernel        FOUR lanes per texel, a channel each: the ordered run blended, 16 texels of a wave side by side
//   crowd_blend_kernel       one workgroup per texel of the long list (windows of stream indices when a run does not fit LDS):
//                            sources staged by all threads, the destination's channels applied side by side, a lane each
constexpr uint32_t kWaveRun = 256;           // runs up to this length are ordered and blended by ONE wave
constexpr uint32_t kGiantRun = 1024;         // longer runs are listed apart (the giants): parted by stream index and ordered window by window

__global__ __launch_bounds__(1024) void crowd_plan_kernel(const DepositParams p)
{
    __shared__ unsigned long long kpart[1024];
    const uint32_t nlarge = p.totals[kTotLarge], per = (nlarge + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per < nlarge ? threadIdx.x * per : nlarge, hi = lo + per < nlarge ? lo + per : nlarge;
    unsigned long long kn = 0;
    for (uint32_t i = lo; i < hi; ++i) kn += bin_places(p, p.large_bins[i]);
    kpart[threadIdx.x] = kn;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const unsigned long long ka = threadIdx.x >= off ? kpart[threadIdx.x - off] : 0ull;
        __syncthreads();
        kpart[threadIdx.x] += ka;
        __syncthreads();
    }
    unsigned long long krun = kpart[threadIdx.x] - kn;
    for (uint32_t i = lo; i < hi; ++i) {
        p.large_key0[i] = (uint32_t)(krun > 0xffffffffull ? 0xffffffffull : krun);
        krun += bin_places(p, p.large_bins[i]);
    }
    if (threadIdx.x == 1023u) {
        const uint32_t keys = (uint32_t)(kpart[1023] > 0xffffffffull ? 0xffffffffull : kpart[1023]);      // (saturated: the host refuses it)
        p.totals[kTotCrowdKeys] = keys;
        // the pass's totals straight into the host's memory, a sequence number behind them: the host polls that word instead of
        // waiting for a copy on another stream to be scheduled, run and signalled (a draw's only round trip: 43 -> ~10 us)
        if (p.totals_host) {
            for (uint32_t w = 0; w < kTotWords; ++w) p.totals_host[w] = w == kTotCrowdKeys ? keys : __hip_atomic_load(&p.totals[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(&p.totals_host[kTotWords], p.totals_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// workgroup block -> (large bin i, one of its lists, the places handed out in it)
TH_D void crowd_block(const DepositParams &p, uint32_t block, uint32_t &i, uint32_t &list, uint32_t &n)
{
    i = block / kBinReplicas;
    const uint32_t r = block - i * kBinReplicas, b = p.large_bins[i];
    list = b * kBinReplicas + r;
    n = *list_cursor(p, b, r);
}
// the fragments of a list in rounds of kCrowdRound places: round(keys of the thread, their places, first index, places in the round)
// (2048: a list of a crowded bin holds 700 fragments on average - one round either way - and eight keys and places a thread
// instead of sixteen are 24 registers: crowd_scatter_kernel 92 -> 6x VGPRs)
constexpr uint32_t kCrowdRound = 2048;
template <typename Round>
TH_D void crowd_rounds(const DepositParams &p, uint32_t list, uint32_t n, Round round)
{
    constexpr uint32_t kPer = kCrowdRound / 256u;
    for (uint32_t f0 = 0; f0 < n; f0 += kCrowdRound) {
        const uint32_t m = n - f0 < kCrowdRound ? n - f0 : kCrowdRound;
        uint32_t at[kPer];
        unsigned long long k[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = q * 256u + threadIdx.x; at[q] = place_of(p, list, f0 + (f < m ? f : m - 1u)); }
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) k[q] = at[q] == kNoPlace ? kEmptyKey : p.frag_keys[at[q]];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) if (q * 256u + threadIdx.x >= m) k[q] = kEmptyKey;
        round(k, at, m);
    }
}

__global__ __launch_bounds__(256) void crowd_hist_kernel(const DepositParams p)
{
    __shared__ uint32_t hist[kBinTexels];
    uint32_t i, list, n;
    crowd_block(p, blockIdx.x, i, list, n);
    if (n == 0u) return;
    hist[threadIdx.x] = 0u;
    __syncthreads();
    crowd_rounds(p, list, n, [&](const unsigned long long (&k)[kCrowdRound / 256u], const uint32_t (&)[kCrowdRound / 256u], uint32_t) {
#pragma unroll
        for (uint32_t q = 0; q < kCrowdRound / 256u; ++q) if (k[q] != kEmptyKey) atomicAdd(&hist[key_local(k[q])], 1u);
    });
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&p.crowd_count[(size_t)i * kBinTexels + threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void crowd_scan_kernel(const DepositParams p)
{
    __shared__ uint32_t wave_total[4];
    const uint32_t i = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t mine = p.crowd_count[(size_t)i * kBinTexels + t];
    p.crowd_count[(size_t)i * kBinTexels + t] = 0u;         // (left empty for the next pass: no memset between a pass's plan and its regroup)
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
    // the lists of the runs a wave does not order, in whatever order: one atomic per wave and list (a crowded target has
    // thousands of such texels - one atomic each on the same two words was most of this kernel's time)
    auto listed = [&](bool is, uint32_t *list, uint32_t *counter) {
        const unsigned long long who = __ballot(is);
        if (who == 0ull) return;
        uint32_t first = 0;
        if (lane == (uint32_t)__builtin_ctzll(who)) first = atomicAdd(counter, (uint32_t)__builtin_popcountll(who));
        first = (uint32_t)__shfl((int)first, __builtin_ctzll(who));
        if (is) list[first + (uint32_t)__builtin_popcountll(who & ((1ull << lane) - 1ull))] = (i << 8) | t;
    };
    listed(mine > kGiantRun, p.crowd_giant, &p.totals[kTotGiant]);
    listed(mine > kWaveRun && mine <= kGiantRun, p.crowd_long, &p.totals[kTotLong]);
}

__global__ __launch_bounds__(256) void crowd_scatter_kernel(const DepositParams p)
{
    __shared__ uint32_t hist[kBinTexels], base[kBinTexels];
    uint32_t i, list, n;
    crowd_block(p, blockIdx.x, i, list, n);
    if (n == 0u) return;
    const uint32_t t = threadIdx.x;
    unsigned long long *out = p.crowd_keys + p.large_key0[i];
    crowd_rounds(p, list, n, [&](const unsigned long long (&k)[kCrowdRound / 256u], const uint32_t (&at)[kCrowdRound / 256u], uint32_t m) {
        hist[t] = 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < kCrowdRound / 256u; ++q) if (k[q] != kEmptyKey) atomicAdd(&hist[key_local(k[q])], 1u);
        __syncthreads();
        base[t] = hist[t] ? atomicAdd(&p.crowd_cursor[(size_t)i * kBinTexels + t], hist[t]) : 0u;
        hist[t] = 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < kCrowdRound / 256u; ++q) {
            if (k[q] != kEmptyKey) {
                const uint32_t lt = key_local(k[q]);
                out[base[lt] + atomicAdd(&hist[lt], 1u)] = ((k[q] & 0xffffffffull) << 32) | at[q];
            }
        }
        __syncthreads();
    });
    pages_forget(p, list, n, t, 256u);
}

// the giants that giant_part_kernel (below) left alone - more of one bucket of stream indices than a window holds - a
// workgroup each: ordered in LDS at once (up to kCrowdCap fragments) or in windows of stream indices narrowed as it goes
template <int MODE>
__global__ __launch_bounds__(256) void crowd_blend_kernel(const DepositParams p, const uint32_t *list, const uint32_t *count)
{
    __shared__ BinShared<MODE> s;
    const uint32_t t = threadIdx.x, nlong = *count;
    for (uint32_t e = blockIdx.x; e < nlong; e += gridDim.x) {
        if (p.crowd_giant_win[2u * e] != 0xffffffffu) continue;    // (parted, ordered window by window and walked: giant_*_kernel)
        const uint32_t entry = list[e], i = entry >> 8, lt = entry & 255u;
        const uint32_t b = p.large_bins[i];
        const uint32_t r0 = p.crowd_start[(size_t)i * (kBinTexels + 1u) + lt], len = p.crowd_start[(size_t)i * (kBinTexels + 1u) + lt + 1u] - r0;
        const uint32_t by = b / p.bins_x, bx = b - by * p.bins_x;
        const uint32_t x = (bx << kBinShift) + (lt & (kBinSide - 1u)), y = (by << kBinShift) + (lt >> kBinShift);
        const uint32_t texel = y * (uint32_t)p.fw + x;            // (a texel with fragments lies inside the target)
        const unsigned long long *run = p.crowd_keys + p.large_key0[i] + r0;
        BinTexel<MODE> d{};
        if (t == 0u) d.load(p, texel);
        unsigned long long *skey = s.skey();
        uint16_t *order = s.order();
        if (len <= kCrowdCap) {
            for (uint32_t f = t; f < len; f += 256u) skey[f] = run[f];
            __syncthreads();
            bin_bitonic(s, len);
            bin_blend_long<MODE>(s, p, 0u, len, 0u, d, [&](uint32_t j) { return (uint32_t)(skey[order[j]] & 0xffffffffull); });
        } else {
            blend_texel_windows<MODE, 32>(s, p, 0u, d, [](uint32_t at) { return at; }, [&](auto body) { for (uint32_t f = t; f < len; f += 256u) body(run[f]); });
        }
        if (t == 0u) d.store(p, texel);
        __syncthreads();
    }
}

TH_D void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// One texel's run of `len` fragments walked by ONE wave: place_at(j) = where the varying(s) of the j-th fragment in blend order
// lie (asked for j < len, by lane j & 63 of a batch).  64 sources staged at a time in the wave's own LDS words (the next
// batches' varyings in flight meanwhile, the places of the batch after those too) and the destination's channels applied by
// lanes 0-7: lanes 0-3 the flow texel's floats, 4-7 the view texel's bytes - the same operations on every channel in the same
// order as one thread doing all eight.  PART (a store of MODE 2 - two varyings per fragment): -1 both targets by this wave,
// 0 / 1 the flow / the view target alone - two waves then walk the run side by side, each its own chain.
template <int MODE, int PART = -1, typename PlaceAt>
TH_D void wave_walk(const DepositParams &p, uint32_t texel, uint32_t len, BlendSource (*stage_a)[64], BlendSource (*stage_b)[(MODE == 2 && PART < 0) ? 64 : 1], PlaceAt place_at)
{
    static_assert(PART < 0 || MODE == 2, "parts are halves of a store with two varyings per fragment");
    constexpr bool kFlow = PART == 0 || (PART < 0 && MODE != 1), kView = PART == 1 || (PART < 0 && MODE != 0), kBoth = kFlow && kView;
    const uint32_t lane = threadIdx.x & 63u;
    const bool flow_lane = kFlow && lane < 4u, view_lane = kView && lane >= 4u && lane < 8u;
    float comp = 0.0f;
    if (flow_lane) comp = reinterpret_cast<const float *>(p.flow + texel)[lane & 3u];
    if (view_lane) comp = (float)reinterpret_cast<const unsigned char *>(p.view + texel)[lane & 3u];
    // kAhead batches of varyings in flight (and the places of the batch after those): a batch is applied in about a
    // microsecond, a scattered read out of a store of a gigabyte can take longer than that
    constexpr uint32_t kAhead = 3;
    auto place_of_batch = [&](uint32_t j0) { const uint32_t j = j0 + lane; return place_at(j < len ? j : len - 1u); };
    auto fetch = [&](uint32_t place, float4 &c0, float4 &c1) {
        if constexpr (PART < 0) fetch_colors<MODE>(p, (size_t)place, c0, c1);
        else { c0 = p.colors[2u * (size_t)place + (uint32_t)PART]; c1 = c0; }
    };
    float4 c0[kAhead], c1[kAhead];
#pragma unroll
    for (uint32_t k = 0; k < kAhead; ++k) if (k * 64u < len) fetch(place_of_batch(k * 64u), c0[k], c1[k]);
    uint32_t at_ahead = place_of_batch(kAhead * 64u < len ? kAhead * 64u : 0u);
    uint32_t buf = 0;
    for (uint32_t j00 = 0; j00 < len; j00 += kAhead * 64u) {
#pragma unroll
        for (uint32_t k = 0; k < kAhead; ++k) {
            const uint32_t j0 = j00 + k * 64u;
            if (j0 >= len) break;
            if constexpr (kFlow) stage_a[buf][lane] = FlowTarget::source(c0[k]);
            else stage_a[buf][lane] = ViewTarget::source(c0[k]);
            if constexpr (kBoth) stage_b[buf][lane] = ViewTarget::source(c1[k]);
            if (j0 + kAhead * 64u < len) {                                // this slot's next batch
                fetch(at_ahead, c0[k], c1[k]);
                if (j0 + (kAhead + 1u) * 64u < len) at_ahead = place_of_batch(j0 + (kAhead + 1u) * 64u);
            }
            wave_sync();
            const uint32_t n = len - j0 < 64u ? len - j0 : 64u;
            if (flow_lane) apply_batch<false>(comp, stage_a[buf], lane & 3u, n);
            if (view_lane) apply_batch<true>(comp, kBoth ? stage_b[kBoth ? buf : 0u] : stage_a[buf], lane & 3u, n);
            buf ^= 1u;
        }
    }
    if (flow_lane) reinterpret_cast<float *>(p.flow + texel)[lane & 3u] = comp;
    if (view_lane) reinterpret_cast<unsigned char *>(p.view + texel)[lane & 3u] = (unsigned char)comp;
}

// ---- the giants (runs of more than kGiantRun fragments) --------------------------------------------------------------------
// Once the wake of a long-running loop has drawn the particles together, single texels receive thousands and tens of
// thousands of fragments per draw, and a thousand such texels at once.  Their runs are put in order in three steps, none of
// which reads a key more than a fixed number of times:
//   giant_part_kernel   a workgroup per run: its keys parted ONCE by the leading bits of their stream indices (1024 buckets
//                       over the range of indices the run holds: histogram, scan, scatter into p.crowd_parted) and the
//                       buckets grouped into WINDOWS of up to kGiantWindow keys - stream indices are distinct inside a
//                       texel; spread evenly, a bucket of a run of 20 000 holds ~20.  (A run with a bucket larger than a
//                       window all the same - dense clusters of indices far apart - is left to crowd_blend_kernel, which
//                       narrows its windows as it goes.)
//   giant_sort_kernel   a workgroup per WINDOW, all windows of all runs side by side: ordered in LDS, the places of the
//                       varyings written in blend order (p.crowd_sorted)
//   run_walk_kernel     a wave per run and target: wave_walk over those places - what is left on the draw's critical path is the chain
//                       of the longest run, one fragment after the other, and nothing else.
// (crowd_blend_kernel did all of this inside one workgroup per run, window after window - and read the whole run twice per
// window: a run of 17 000 fragments took 1.5 ms, profiles/r4_g_giants.txt.)
constexpr uint32_t kGiantWindow = 2048;
constexpr uint32_t kGiantFallback = 0xffffffffu;
struct GiantRun { uint32_t i, texel, r0, len; };
TH_D GiantRun giant_run(const DepositParams &p, uint32_t entry)
{
    GiantRun g;
    g.i = entry >> 8;
    const uint32_t lt = entry & 255u, b = p.large_bins[g.i];
    g.r0 = p.crowd_start[(size_t)g.i * (kBinTexels + 1u) + lt];
    g.len = p.crowd_start[(size_t)g.i * (kBinTexels + 1u) + lt + 1u] - g.r0;
    const uint32_t by = b / p.bins_x, bx = b - by * p.bins_x;
    const uint32_t x = (bx << kBinShift) + (lt & (kBinSide - 1u)), y = (by << kBinShift) + (lt >> kBinShift);
    g.texel = y * (uint32_t)p.fw + x;               // (a texel with fragments lies inside the target)
    return g;
}

__global__ __launch_bounds__(256) void giant_part_kernel(const DepositParams p)
{
    __shared__ uint32_t hist[1024], start[1025], wave_total[4], win[2], span[2];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, ngiant = p.totals[kTotGiant];
    constexpr uint32_t kPer = 8;
    for (uint32_t e = blockIdx.x; e < ngiant; e += gridDim.x) {
        const GiantRun g = giant_run(p, p.crowd_giant[e]);
        const unsigned long long *run = p.crowd_keys + p.large_key0[g.i] + g.r0;
        unsigned long long *out = p.crowd_parted + p.large_key0[g.i] + g.r0;
        for (uint32_t k = t; k < 1024u; k += 256u) hist[k] = 0u;
        if (t == 0u) { span[0] = 0xffffffffu; span[1] = 0u; }
        __syncthreads();
        auto for_run = [&](auto body) {              // (a thread's loads of one round go out together)
            for (uint32_t f0 = 0; f0 < g.len; f0 += kPer * 256u) {
                unsigned long long k[kPer];
#pragma unroll
                for (uint32_t q = 0; q < kPer; ++q) { const uint32_t f = f0 + q * 256u + t; k[q] = run[f < g.len ? f : g.len - 1u]; }
#pragma unroll
                for (uint32_t q = 0; q < kPer; ++q) if (f0 + q * 256u + t < g.len) body(k[q]);
            }
        };
        // the buckets cover the stream indices the run HOLDS, not all there are: the particles of a spawn - neighbours in the
        // state texture - stay together for a long time, and a thousand buckets over sixteen million indices put all of them
        // into one (such runs are a tenth of a crowded frame's giants: they went the slow way, 30-40 us at the end of the
        // giants' stream)
        {
            uint32_t lo = 0xffffffffu, hi = 0u;
            for_run([&](unsigned long long k) { const uint32_t id = (uint32_t)(k >> 32); lo = id < lo ? id : lo; hi = id > hi ? id : hi; });
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, o), h2 = (uint32_t)__shfl_xor((int)hi, o);
                lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
            }
            if (lane == 0u) { atomicMin(&span[0], lo); atomicMax(&span[1], hi); }
        }
        __syncthreads();
        const uint32_t id0 = span[0], width = span[1] - span[0];
        uint32_t shift = 0;
        while (shift < 32u && (width >> shift) >= 1024u) ++shift;
        auto bucket = [&](unsigned long long k) { return ((uint32_t)(k >> 32) - id0) >> shift; };       // (< 1024)
        for_run([&](unsigned long long k) { atomicAdd(&hist[bucket(k)], 1u); });
        __syncthreads();
        // exclusive scan of the buckets (every thread its four), the largest bucket
        const uint32_t h0 = hist[4u * t], h1 = hist[4u * t + 1u], h2 = hist[4u * t + 2u], h3 = hist[4u * t + 3u], mine = h0 + h1 + h2 + h3;
        uint32_t incl = mine, most = h0 > h1 ? h0 : h1;
        most = most > h2 ? most : h2; most = most > h3 ? most : h3;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o) incl += up;
            const uint32_t other = __shfl_xor(most, o);
            most = other > most ? other : most;
        }
        if (lane == 63u) wave_total[wave] = incl;
        if (t == 0u) win[0] = 0u;
        __syncthreads();
        if (lane == 0u && most > kGiantWindow) win[0] = 1u;
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += wave_total[w];
        const uint32_t s0 = before + incl - mine;
        start[4u * t] = s0; start[4u * t + 1u] = s0 + h0; start[4u * t + 2u] = s0 + h0 + h1; start[4u * t + 3u] = s0 + h0 + h1 + h2;
        if (t == 255u) start[1024] = s0 + mine;
        hist[4u * t] = s0; hist[4u * t + 1u] = s0 + h0; hist[4u * t + 2u] = s0 + h0 + h1; hist[4u * t + 3u] = s0 + h0 + h1 + h2;      // (fill cursors)
        __syncthreads();
        if (win[0]) {                                // a bucket no window holds: crowd_blend_kernel takes the run
            if (t == 0u) { p.crowd_giant_win[2u * e] = kGiantFallback; p.crowd_giant_win[2u * e + 1u] = 0u; }
            __syncthreads();
            continue;
        }
        for_run([&](unsigned long long k) { out[atomicAdd(&hist[bucket(k)], 1u)] = k; });
        // the windows: as many leading buckets as kGiantWindow keys allow, again and again (one wave; every window's end is
        // found by bisection over the buckets' starts)
        if (wave == 0u) {
            // (all lanes alike; counted first - the list wants a run's windows side by side - then written.  A run whose
            // windows the list has no room for - its size bounds them: more than kGiantWindow keys in any two neighbours - goes
            // the other way too)
            uint32_t n = 0, first = 0;
            bool fits = true;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {
                    if (lane == 0u) first = atomicAdd(&p.totals[kTotWindows], n);
                    first = (uint32_t)__shfl((int)first, 0);
                    fits = (unsigned long long)first + n <= p.crowd_windows_cap;
                    if (lane == 0u) { p.crowd_giant_win[2u * e] = fits ? first : kGiantFallback; p.crowd_giant_win[2u * e + 1u] = n; }
                }
                uint32_t b0 = 0;
                n = 0;
                while (b0 < 1024u && start[b0] < g.len) {
                    uint32_t lo = b0 + 1u, hi = 1024u;             // the largest b1 in (b0, 1024] with start[b1] - start[b0] <= kGiantWindow
                    while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (start[mid] - start[b0] <= kGiantWindow) lo = mid; else hi = mid - 1u; }
                    if (pass == 1 && lane == 0u && (unsigned long long)first + n < p.crowd_windows_cap) {
                        uint32_t *w = p.crowd_windows + 3u * (size_t)(first + n);
                        w[0] = e; w[1] = start[b0]; w[2] = fits ? start[lo] - start[b0] : 0u;
                    }
                    ++n;
                    b0 = lo;
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void giant_sort_kernel(const DepositParams p)
{
    __shared__ unsigned long long skey[kGiantWindow];
    const uint32_t t = threadIdx.x, total = p.totals[kTotWindows], nwin = total < p.crowd_windows_cap ? total : p.crowd_windows_cap;
    for (uint32_t w = blockIdx.x; w < nwin; w += gridDim.x) {
        const uint32_t *rec = p.crowd_windows + 3u * (size_t)w;
        const uint32_t e = rec[0], first = rec[1], m = rec[2];
        if (m == 0u) continue;                      // (a run the list had no room for)
        const GiantRun g = giant_run(p, p.crowd_giant[e]);
        const size_t base = (size_t)p.large_key0[g.i] + g.r0 + first;
        uint32_t P = 64u;
        while (P < m) P <<= 1;
        for (uint32_t f = t; f < P; f += 256u) skey[f] = f < m ? p.crowd_parted[base + f] : ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t q = t; q < (P >> 1); q += 256u) {
                    const uint32_t lo = ((q & ~(j - 1u)) << 1) | (q & (j - 1u)), hi = lo | j;
                    const unsigned long long a = skey[lo], c = skey[hi];
                    const bool up = (lo & k) == 0u;
                    if ((a > c) == up) { skey[lo] = c; skey[hi] = a; }
                }
                __syncthreads();
            }
        for (uint32_t f = t; f < m; f += 256u) p.crowd_sorted[base + f] = (uint32_t)(skey[f] & 0xffffffffull);
        __syncthreads();
    }
}

// The long list (kWaveRun < fragments <= kGiantRun): ONE WAVE per run orders it, four runs per workgroup side by side - a
// bitonic network in the wave's own LDS words, no workgroup barrier between its steps (a workgroup per run spent most of its
// time in the 55 barriers of a thousand-key sort) - and leaves the places of the varyings in blend order (p.crowd_sorted)
// for run_walk_kernel, as the giants' windows do.
__global__ __launch_bounds__(256) void long_sort_kernel(const DepositParams p)
{
    __shared__ unsigned long long skeys[4][kGiantRun];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nlong = p.totals[kTotLong];
    unsigned long long *skey = skeys[wave];
    for (uint32_t e = blockIdx.x * 4u + wave; e < nlong; e += gridDim.x * 4u) {
        const GiantRun g = giant_run(p, p.crowd_long[e]);
        const size_t base = (size_t)p.large_key0[g.i] + g.r0;
        // the keys (stream index << 32 | place of the varying), padded to a power of two, ordered
        uint32_t P = 64u;
        while (P < g.len) P <<= 1;
        for (uint32_t f = lane; f < P; f += 64u) skey[f] = f < g.len ? p.crowd_keys[base + f] : ~0ull;
        wave_sync();
        for (uint32_t k = 2; k <= P; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t q = lane; q < (P >> 1); q += 64u) {
                    const uint32_t lo = ((q & ~(j - 1u)) << 1) | (q & (j - 1u)), hi = lo | j;
                    const unsigned long long a = skey[lo], c = skey[hi];
                    const bool up = (lo & k) == 0u;
                    if ((a > c) == up) { skey[lo] = c; skey[hi] = a; }
                }
                wave_sync();
            }
        for (uint32_t f = lane; f < g.len; f += 64u) p.crowd_sorted[base + f] = (uint32_t)(skey[f] & 0xffffffffull);
        wave_sync();                                              // (the words are free for the wave's next run)
    }
}

// The runs of a list (GIANTS: p.crowd_giant, else p.crowd_long) walked over their places in blend order, a wave per run and
// target: with both targets in one store two waves walk a run side by side, each its own chain.
template <int MODE, bool GIANTS>
__global__ __launch_bounds__(256) void run_walk_kernel(const DepositParams p)
{
    __shared__ BlendSource stage_a[4][2][64], none[1][1];
    constexpr uint32_t kParts = MODE == 2 ? 2u : 1u;
    const uint32_t wave = threadIdx.x >> 6, ntask = p.totals[GIANTS ? kTotGiant : kTotLong] * kParts;
    const uint32_t *list = GIANTS ? p.crowd_giant : p.crowd_long;
    for (uint32_t task = blockIdx.x * 4u + wave; task < ntask; task += gridDim.x * 4u) {
        const uint32_t e = task / kParts;
        if (GIANTS && p.crowd_giant_win[2u * e] == kGiantFallback) continue;
        const GiantRun g = giant_run(p, list[e]);
        const uint32_t *sorted = p.crowd_sorted + p.large_key0[g.i] + g.r0;
        auto place_at = [&](uint32_t j) { return sorted[j]; };
        if constexpr (MODE == 2) {
            if (task & 1u) wave_walk<2, 1>(p, g.texel, g.len, stage_a[wave], none, place_at);
            else wave_walk<2, 0>(p, g.texel, g.len, stage_a[wave], none, place_at);
        } else wave_walk<MODE>(p, g.texel, g.len, stage_a[wave], none, place_at);
        wave_sync();
    }
}

// The runs of up to kWaveRun fragments in two steps.  crowd_sort_kernel: one WAVE per texel orders its run (rank by counting,
// every lane its <= 4 keys against all of the run's, read from LDS as broadcasts) and leaves the places of the varyings in blend order - tens of thousands of independent waves.
// crowd_walk_kernel: 16 texels per wave, FOUR lanes per texel - a channel of the destination(s) each -, every quad walking the
// run of its own texel: 16 chains side by side, each a quarter as long as one thread's doing all channels.  With one wave per
// texel for the whole job every lane applies every fragment to its own copy of the destination: 64 times the chain's
// arithmetic, and the chain (the view target's clamp, convert and round per channel on top of the flow target's multiply
// and add) is most of that kernel once runs are a hundred fragments long; a lane per texel for the whole job orders 64 runs
// one after the other per wave and waits for each.
// a run of up to 64 W fragments ordered by the LANES lanes [first, first + LANES) of a wave (W keys per lane... K keys each):
// rank by counting over the run's stream indices in `ids` (LDS words of those lanes alone)
template <uint32_t LANES, uint32_t K>
TH_D void lanes_rank_sort(const unsigned long long *run, uint32_t *sorted, uint32_t len, uint32_t *ids, uint32_t sl, uint32_t trips)
{
    unsigned long long mine[K];
    uint32_t id[K], rank[K];
#pragma unroll
    for (uint32_t q = 0; q < K; ++q) {
        const uint32_t f = q * LANES + sl;
        mine[q] = len ? run[f < len ? f : len - 1u] : 0ull;
        id[q] = (uint32_t)(mine[q] >> 32);            // (the stream indices alone order a texel's run: a line covers a texel at most once)
        rank[q] = 0u;
        ids[f] = f < len ? id[q] : 0xffffffffu;          // (past the run's end: an index no fragment's is greater than - it counts nothing)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (`trips`: the longest run of the wave's groups - every lane goes round as often, a group past its own run counts nothing)
    // Four stream indices a trip, one 16-byte LDS read (the groups' words start 256 B apart; every word a trip can read was
    // written above - trips <= LANES * K): a trip per index waited for its own read - crowd_sort_kernel 208 -> 180 us; a compare
    // and an add with carry per key and index, nothing about the run's end: ->
    const uint4 *ids4 = reinterpret_cast<const uint4 *>(ids);
    for (uint32_t j = 0; j < trips; j += 4u) {
        const uint4 k4 = ids4[j >> 2];
        const uint32_t k[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
        for (uint32_t e = 0; e < 4u; ++e) {
#pragma unroll
            for (uint32_t q = 0; q < K; ++q) rank[q] += k[e] < id[q] ? 1u : 0u;
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < K; ++q) { const uint32_t f = q * LANES + sl; if (f < len) sorted[rank[q]] = (uint32_t)(mine[q] & 0xffffffffull); }
}

// Four texels per wave.  Most runs of a crowded bin are a few dozen fragments long: a wave per texel spent its time waiting
// for three dependent loads to order thirty keys.  When all four runs have at most 64 fragments, 16 lanes take each (4 keys a
// lane); otherwise the wave takes them one after the other (4 keys a lane, up to kWaveRun).
__global__ __launch_bounds__(256) void crowd_sort_kernel(const DepositParams p)
{
    static_assert(kWaveRun == 256u, "four keys per lane");
    __shared__ __align__(16) uint32_t ids[4][kWaveRun];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, sub = lane >> 4, sl = lane & 15u;
    const uint32_t g = blockIdx.x * 4u + wave, i = g >> 6, lt0 = (g & 63u) << 2;          // texels lt0 .. lt0 + 3 of large bin i
    if (i >= p.nlarge) return;
    const uint32_t *start = p.crowd_start + (size_t)i * (kBinTexels + 1u) + lt0;
    const uint32_t r0 = start[sub];
    uint32_t len = start[sub + 1u] - r0;
    if (len > kWaveRun) len = 0u;                                      // (longer ones: the long list's kernels)
    uint32_t longest = len;
    longest = max(longest, (uint32_t)__shfl_xor((int)longest, 16));
    longest = max(longest, (uint32_t)__shfl_xor((int)longest, 32));
    if (longest == 0u) return;
    const size_t key0 = p.large_key0[i];
    // (as many keys a lane as the longest run asks for: most runs of a crowded bin are a few dozen fragments long, and every key
    // a lane holds is compared on every trip - four where one would do was three quarters of this kernel's arithmetic)
    if (longest <= 64u) {
        const unsigned long long *run = p.crowd_keys + key0 + r0;
        uint32_t *sorted = p.crowd_sorted + key0 + r0, *words = ids[wave] + sub * 64u;
        if (longest <= 16u) lanes_rank_sort<16u, 1u>(run, sorted, len, words, sl, longest);
        else if (longest <= 32u) lanes_rank_sort<16u, 2u>(run, sorted, len, words, sl, longest);
        else lanes_rank_sort<16u, 4u>(run, sorted, len, words, sl, longest);
        return;
    }
    for (uint32_t q = 0; q < 4u; ++q) {
        const uint32_t qr0 = (uint32_t)__shfl((int)r0, (int)(q << 4)), qlen = (uint32_t)__shfl((int)len, (int)(q << 4));
        if (qlen == 0u) continue;
        const unsigned long long *run = p.crowd_keys + key0 + qr0;
        uint32_t *sorted = p.crowd_sorted + key0 + qr0;
        if (qlen <= 64u) lanes_rank_sort<64u, 1u>(run, sorted, qlen, ids[wave], lane, qlen);
        else if (qlen <= 128u) lanes_rank_sort<64u, 2u>(run, sorted, qlen, ids[wave], lane, qlen);
        else lanes_rank_sort<64u, 4u>(run, sorted, qlen, ids[wave], lane, qlen);
        __builtin_amdgcn_wave_barrier();                              // (everybody is done with the words before the next run's)
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void crowd_walk_kernel(const DepositParams p)
{
    // four lanes per texel, a channel each (of the flow texel and / or of the view texel): 16 runs side by side per wave, every
    // run's chain a quarter as long as with a lane per texel doing all channels (the pass is bound by its longest chains)
    // The 16 waves of a bin on ONE XCD (workgroups go to the eight XCDs in turn: blockIdx.x % 8): the varyings of a bin's
    // fragments lie in the order they arrived, four to a 128-byte line, and its texels' runs pick them out in blend order -
    // with the bin's waves spread over all XCDs every L2 fetched most of the bin's lines for itself (3.4 x the bytes the
    // fragments hold, and the kernel runs at the memory's rate: profiles/r4_g_giants.txt)
    const uint32_t lane = threadIdx.x, c = lane & 3u, xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
    const uint32_t i = ((k >> 4) << 3) + xcd, lt = ((k & 15u) << 4) + (lane >> 2);
    if (i >= p.nlarge) return;
    const uint32_t r0 = p.crowd_start[(size_t)i * (kBinTexels + 1u) + lt], len = p.crowd_start[(size_t)i * (kBinTexels + 1u) + lt + 1u] - r0;
    if (len == 0u || len > kWaveRun) return;                  // (longer runs: crowd_blend_kernel, sources by the workgroup)
    const uint32_t b = p.large_bins[i];
    const uint32_t by = b / p.bins_x, bx = b - by * p.bins_x;
    const uint32_t x = (bx << kBinShift) + (lt & (kBinSide - 1u)), y = (by << kBinShift) + (lt >> kBinShift);
    const uint32_t texel = y * (uint32_t)p.fw + x;
    const uint32_t *sorted = p.crowd_sorted + p.large_key0[i] + r0;
    quad_walk<MODE>(p, texel, len, c, [sorted](uint32_t j) { return sorted[j]; });
}

// ---- row-band shards: the bins travel to the ranks that own them -----------------------------------------------------------
// A sharded job draws with the binned pipeline like a single context does - every rank rasterises its band's lines into
// ITS page store (bins_fused_kernel, untouched) - and then the bins change hands: a rank owns whole bin rows of the target
// (OwnerParams::bin_lo: contiguous ranges, hence contiguous texel ranges for the all-gather that follows), and what a rank
// emitted into the bins of ANOTHER owner r is compacted into one contiguous part per owner (keys + varyings, bin by bin, with
// the per-bin counts beside it) and exchanged; what it emitted into its OWN bins stays where it is, and what the other ranks
// send for those bins is appended to their lists exactly as if this rank had emitted it too: cursors moved on, pages from the
// pool, page table - so that the plan and every blend kernel run unchanged.  The order inside a bin is of no consequence (the
// blend restores GL's order from the keys), so nothing is sorted anywhere on the way; a world of one moves nothing at all.
//   owner_places_kernel, owner_counts_kernel   places per bin; their exclusive scan (= where each bin goes in the outgoing
//                          arrays) and the owners' bounds
//   owner_extract_kernel   a workgroup per bin: its lists walked, fragments copied out; the lists' pages forgotten
//   owner_prefix_kernel    (owner) per source: where each of my bins starts inside that source's part
//   owner_totals_kernel, owner_pages_kernel, owner_layout_kernel   (owner) per bin: places arriving; the pool pages they open
//                          (a scan); page table and cursors
//   owner_insert_kernel    (owner) a workgroup per bin: every source's fragments of the bin copied to their places

TH_D bool owner_mine(const OwnerParams &o, uint32_t b) { return b >= o.bin_lo[o.rank] && b < o.bin_lo[o.rank + 1u]; }

// what leaves: the places of every bin of another owner (my own bins stay in my store)
__global__ __launch_bounds__(256) void owner_places_kernel(const DepositParams p, const OwnerParams o)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b < p.nbins) o.counts[b] = owner_mine(o, b) ? 0u : bin_places(p, b);
}

__global__ __launch_bounds__(1024) void owner_counts_kernel(const DepositParams p, const OwnerParams o)
{
    __shared__ unsigned long long lds[1024];
    const uint32_t per = (p.nbins + 1023u) / 1024u, lo = threadIdx.x * per < p.nbins ? threadIdx.x * per : p.nbins, hi = lo + per < p.nbins ? lo + per : p.nbins;
    unsigned long long n = 0, total = 0;
    for (uint32_t b = lo; b < hi; ++b) n += o.counts[b];
    unsigned long long run = block_scan_1024(lds, n, total);
    for (uint32_t b = lo; b < hi; ++b) {
        for (uint32_t r = 0; r < o.world; ++r) if (o.bin_lo[r] == b) o.owner_bounds[r] = run;
        o.offsets[b] = run;
        run += o.counts[b];
    }
    if (threadIdx.x == 0u) {
        o.offsets[p.nbins] = total;
        for (uint32_t r = 0; r <= o.world; ++r) if (o.bin_lo[r] >= p.nbins) o.owner_bounds[r] = total;
    }
}

// the places of bin b walked list after list: place f of the walk (lists[] = first place of every list, lists[kBinReplicas] = all)
TH_D uint32_t walk_place(const DepositParams &p, uint32_t b, const uint32_t *lists, uint32_t f)
{
    uint32_t r = 0;
#pragma unroll
    for (uint32_t q = 1; q < kBinReplicas; ++q) r += f >= lists[q] ? 1u : 0u;
    return place_of(p, b * kBinReplicas + r, f - lists[r]);
}

template <bool PAIRS>
__global__ __launch_bounds__(256) void owner_extract_kernel(const DepositParams p, const OwnerParams o)
{
    __shared__ uint32_t lists[kBinReplicas + 1u];
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    if (owner_mine(o, b)) return;                       // (stays: the others' fragments of this bin will join it)
    if (t == 0u) {
        uint32_t run = 0;
        for (uint32_t r = 0; r < kBinReplicas; ++r) { lists[r] = run; run += *list_cursor(p, b, r); }
        lists[kBinReplicas] = run;
    }
    __syncthreads();
    const uint32_t n = lists[kBinReplicas];
    const unsigned long long base = o.offsets[b];
    for (uint32_t f = t; f < n; f += 256u) {
        const uint32_t at = walk_place(p, b, lists, f);
        o.out_keys[base + f] = p.frag_keys[at];
        if constexpr (PAIRS) { o.out_colors[2u * (base + f)] = p.colors[2u * (size_t)at]; o.out_colors[2u * (base + f) + 1u] = p.colors[2u * (size_t)at + 1u]; }
        else o.out_colors[base + f] = p.colors[at];
    }
    __syncthreads();
    for (uint32_t r = 0; r < kBinReplicas; ++r) if (lists[r + 1u] - lists[r] > kBinPage) pages_forget(p, b * kBinReplicas + r, lists[r + 1u] - lists[r], t, 256u);
    if (t < kBinReplicas) *list_cursor(p, b, t) = 0u;    // (the bin is somebody else's: nothing of it is blended here)
}

// table[s][b'] (what source s holds for my bin b') -> src_prefix[s][b'] (its exclusive scan over b'); one workgroup per source
__global__ __launch_bounds__(1024) void owner_prefix_kernel(const OwnerParams o)
{
    __shared__ unsigned long long lds[1024];
    const uint32_t s = blockIdx.x, per = (o.nb + 1023u) / 1024u, lo = threadIdx.x * per < o.nb ? threadIdx.x * per : o.nb, hi = lo + per < o.nb ? lo + per : o.nb;
    const uint32_t *row = o.table + (size_t)s * o.nb;
    unsigned long long n = 0, total = 0;
    for (uint32_t b = lo; b < hi; ++b) n += row[b];
    unsigned long long run = block_scan_1024(lds, n, total);
    for (uint32_t b = lo; b < hi; ++b) { o.src_prefix[(size_t)s * o.nb + b] = run; run += row[b]; }
}

// How the `incoming` places of a bin are spread over its lists: the same share for every list, appended behind what the list
// holds.  take(r): list r's share; a list holding e places that takes n more opens the pages (e + 255) / 256 (from page 1 on:
// page 0 is the list's own) ... (e + n - 1) / 256.
TH_D uint32_t owner_share(uint32_t incoming) { return (incoming + kBinReplicas - 1u) / kBinReplicas; }
TH_D uint32_t owner_take(uint32_t incoming, uint32_t r)
{
    const uint32_t len = owner_share(incoming);
    const unsigned long long begin = (unsigned long long)r * len;
    return incoming > begin ? (uint32_t)(incoming - begin < len ? incoming - begin : len) : 0u;
}
TH_D void owner_new_pages(uint32_t e, uint32_t n, uint32_t &first, uint32_t &count)
{
    first = (e + kBinPage - 1u) >> kPageShift;
    if (first == 0u) first = 1u;
    const uint32_t last = n ? (e + n - 1u) >> kPageShift : 0u;
    count = n && last >= first ? last - first + 1u : 0u;
}

// per bin of mine: places arriving from the other ranks (0 and a flag when a list would outgrow its page table)
__global__ __launch_bounds__(256) void owner_totals_kernel(const DepositParams p, const OwnerParams o)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= o.nb) return;
    unsigned long long total = 0;
    for (uint32_t s = 0; s < o.world; ++s) total += o.table[(size_t)s * o.nb + b];
    const uint32_t bin = o.bin_lo[o.rank] + b;
    bool full = total > (unsigned long long)p.max_pages * kBinPage * kBinReplicas;
    if (!full)
        for (uint32_t r = 0; r < kBinReplicas; ++r)
            full = full || (unsigned long long)*list_cursor(p, bin, r) + owner_take((uint32_t)total, r) > (unsigned long long)p.max_pages * kBinPage;
    if (full) { bins_flag(p, kBinsBinFull); total = 0; }
    o.bin_total[b] = (uint32_t)total;
}

// the pool pages every bin of mine opens for what arrives, and the first of them (o.bin_page: their exclusive scan behind the
// pages the emitting pass took: o.pool_used); the pool's use in totals
__global__ __launch_bounds__(1024) void owner_pages_kernel(const DepositParams p, const OwnerParams o)
{
    __shared__ unsigned long long lds[1024];
    const uint32_t per = (o.nb + 1023u) / 1024u, lo = threadIdx.x * per < o.nb ? threadIdx.x * per : o.nb, hi = lo + per < o.nb ? lo + per : o.nb;
    auto pages_of = [&](uint32_t b) {
        uint32_t n = 0;
        const uint32_t total = o.bin_total[b], bin = o.bin_lo[o.rank] + b;
        for (uint32_t r = 0; r < kBinReplicas && total; ++r) { uint32_t first, count; owner_new_pages(*list_cursor(p, bin, r), owner_take(total, r), first, count); n += count; }
        return n;
    };
    unsigned long long pages = 0, all = 0;
    for (uint32_t b = lo; b < hi; ++b) pages += pages_of(b);
    unsigned long long first = block_scan_1024(lds, pages, all);
    for (uint32_t b = lo; b < hi; ++b) {
        o.bin_page[b] = (uint32_t)(first > 0xffffffffull ? 0xffffffffull : first);
        first += pages_of(b);
    }
    if (threadIdx.x == 0u) {
        const unsigned long long used = (unsigned long long)o.pool_used + all;
        p.totals[kTotPool] = (uint32_t)(used > 0xffffffffull ? 0xffffffffull : used);
        if (used > p.pool_pages) bins_flag(p, kBinsPoolExhausted);      // (the host grows the store - keeping what is in it - and lays the bins out again)
    }
}

// a bin's page table and cursors moved on, one thread per bin
__global__ __launch_bounds__(256) void owner_layout_kernel(const DepositParams p, const OwnerParams o)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= o.nb || p.totals[kTotFlags] != 0u) return;
    const uint32_t total = o.bin_total[b], bin = o.bin_lo[o.rank] + b;
    if (total == 0u) return;
    uint32_t page = p.nbins * kBinReplicas + o.pool_used + o.bin_page[b];
    for (uint32_t r = 0; r < kBinReplicas; ++r) {
        uint32_t *cursor = list_cursor(p, bin, r);
        const uint32_t e = *cursor, n = owner_take(total, r);
        uint32_t first, count;
        owner_new_pages(e, n, first, count);
        for (uint32_t k = 0; k < count; ++k) p.page_table[(size_t)(bin * kBinReplicas + r) * p.max_pages + first + k] = page++;
        *cursor = e + n;
    }
}

template <bool PAIRS>
__global__ __launch_bounds__(256) void owner_insert_kernel(const DepositParams p, const OwnerParams o)
{
    __shared__ uint32_t was[kBinReplicas];              // where every list of the bin ended before the layout moved its cursor on
    const uint32_t b = blockIdx.x, bin = o.bin_lo[o.rank] + b, t = threadIdx.x;
    const uint32_t total = o.bin_total[b], len = owner_share(total);
    if (total == 0u || p.totals[kTotFlags] != 0u) return;
    if (t < kBinReplicas) was[t] = *list_cursor(p, bin, t) - owner_take(total, t);
    __syncthreads();
    uint32_t seen = 0;                                  // places of the bin's incoming sequence taken by the sources before s
    for (uint32_t s = 0; s < o.world; ++s) {
        const uint32_t n = o.table[(size_t)s * o.nb + b];
        const unsigned long long from = o.recv_base[s] + o.src_prefix[(size_t)s * o.nb + b];
        for (uint32_t f = t; f < n; f += 256u) {
            const uint32_t j = seen + f, r = j / len, v = was[r] + (j - r * len);
            const uint32_t at = place_of(p, bin * kBinReplicas + r, v);
            p.frag_keys[at] = o.in_keys[from + f];
            if constexpr (PAIRS) { p.colors[2u * (size_t)at] = o.in_colors[2u * (from + f)]; p.colors[2u * (size_t)at + 1u] = o.in_colors[2u * (from + f) + 1u]; }
            else p.colors[at] = o.in_colors[from + f];
        }
        seen += n;
    }
}

}  // namespace

// A band's first and last row of both state buffers gathered into texel order, f32 (what the neighbouring bands' lines read
// when the vertex lookup of the texture drifts across rows): through the source tables when the band is held in a slot order.
__global__ __launch_bounds__(256) void bins_edge_rows_kernel(const DepositParams p, float4 *rows)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 2u * p.W) return;
    const uint32_t col = i < p.W ? i : i - p.W, row = i < p.W ? 0u : p.rows - 1u;
    size_t at = dep_slot_of(p, (int)row, (int)col);
    if (at == ~(size_t)0) { *p.oob = 1u; at = 0; }
    float4 *out = rows + (i < p.W ? 0u : 2u * (size_t)p.W) + col;
    out[0] = dep_state(p, p.cur, at);
    out[p.W] = dep_state(p, p.prev, at);
}
void launch_bins_edge_rows(const DepositParams &p, float4 *rows, hipStream_t s)
{
    hipLaunchKernelGGL(bins_edge_rows_kernel, dim3((2u * p.W + 255u) / 256u), dim3(256), 0, s, p, rows);
}

void launch_bins_block_list(const DepositParams &p, uint8_t *flags, uint32_t *list, uint32_t *count, uint32_t *src_slots, hipStream_t s)
{
    const uint32_t blocks = (p.W * p.rows + 255u) / 256u;
    hipLaunchKernelGGL(bins_block_flags_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, s, p, flags, src_slots);
    hipLaunchKernelGGL(bins_block_list_kernel, dim3(1), dim3(1024), 0, s, (const uint8_t *)flags, blocks, list, count);
}

void launch_bins_fused(const DepositParams &p, hipStream_t s)
{
    const uint32_t blocks = p.draw_nblocks;
    // the pass's totals, the two lists' counters and the bins' cursors start from zero: one launch (three memsets are three
    // launches with their gaps, between a step and a draw that wait for each other)
    hipLaunchKernelGGL(bins_zero_kernel, dim3(128), dim3(256), 0, s, p.totals, kTotWords, p.list_n, (uint32_t)kListKinds * kDepLists * kDepListStride, p.bin_cursor, kBinReplicas * p.bin_stride);
    // (one short workgroup per listed block of 256 slots - exactly one: the kernel has no loop over blocks, see there.  Measured
    // and not kept in round 3, profiles/r3_b_fused_pass_experiments.txt: a resident grid of 4 / 8 / 16 workgroups per CU walking
    // the blocks; workgroups of 64 or 128 slots; register CAPS for 5, 6 or 8 waves per SIMD instead of 4 - spills; what got it to
    // six in round 5 was fewer registers needed, not fewer allowed: profiles/r5_h_emit_taken_apart.txt)
    // (DEAL: the rows of a wave's lines dealt evenly to its lanes; every lane walking its own line's rows was 0.65 against 0.58 ms)
    if (!p.packed && !p.src.row_index) hipLaunchKernelGGL((bins_fused_kernel<256u, true, true>), dim3(blocks ? blocks : 1u), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((bins_fused_kernel<256u, true, false>), dim3(blocks ? blocks : 1u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_listed_kernel, dim3(2u * kDepLists * 6u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(bins_span_kernel, dim3(kDepLists * (p.src.row_index ? 32u : 4u)), dim3(256), 0, s, p);      // (drifting lookups: tens of thousands of them; the lines are taken from a queue)
    hipLaunchKernelGGL(bins_plan_kernel, dim3((p.nbins + 255u) / 256u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(crowd_plan_kernel, dim3(1), dim3(1024), 0, s, p);
}


// ---- the bins on their way to their owners (row-band shards) --------------------------------------------------------------
// sender: places per bin, their offsets in the outgoing arrays, the owners' bounds (o.owner_bounds: world + 1 values)
void launch_bins_owner_counts(const DepositParams &p, const OwnerParams &o, hipStream_t s)
{
    hipLaunchKernelGGL(owner_places_kernel, dim3((p.nbins + 255u) / 256u), dim3(256), 0, s, p, o);
    hipLaunchKernelGGL(owner_counts_kernel, dim3(1), dim3(1024), 0, s, p, o);
}
// sender: every bin's fragments into o.out_keys / o.out_colors at o.offsets; the store is left ready for the next emit
void launch_bins_owner_extract(const DepositParams &p, const OwnerParams &o, hipStream_t s)
{
    if (p.mode == 2) hipLaunchKernelGGL(owner_extract_kernel<true>, dim3(p.nbins), dim3(256), 0, s, p, o);
    else hipLaunchKernelGGL(owner_extract_kernel<false>, dim3(p.nbins), dim3(256), 0, s, p, o);
}
// owner: the received parts (o.table, o.recv_base, o.in_keys / o.in_colors) appended to this context's own bins as if it had
// emitted them too (o.pool_used: the pool pages its emitting pass took); then the plan of the ordinary binned pass (large
// bins, totals).  p.totals must have been zeroed.
void launch_bins_owner_insert(const DepositParams &p, const OwnerParams &o, hipStream_t s)
{
    hipLaunchKernelGGL(owner_prefix_kernel, dim3(o.world), dim3(1024), 0, s, o);
    if (o.nb) {
        hipLaunchKernelGGL(owner_totals_kernel, dim3((o.nb + 255u) / 256u), dim3(256), 0, s, p, o);
        hipLaunchKernelGGL(owner_pages_kernel, dim3(1), dim3(1024), 0, s, p, o);
        hipLaunchKernelGGL(owner_layout_kernel, dim3((o.nb + 255u) / 256u), dim3(256), 0, s, p, o);
        if (p.mode == 2) hipLaunchKernelGGL(owner_insert_kernel<true>, dim3(o.nb), dim3(256), 0, s, p, o);
        else hipLaunchKernelGGL(owner_insert_kernel<false>, dim3(o.nb), dim3(256), 0, s, p, o);
    }
    hipLaunchKernelGGL(bins_plan_kernel, dim3((p.nbins + 255u) / 256u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(crowd_plan_kernel, dim3(1), dim3(1024), 0, s, p);
}

// p.nlarge: the large bins, as the plan counted them (totals[kTotLarge])
void launch_bins_regroup(const DepositParams &p, hipStream_t s)
{
    if (!p.nlarge) return;
    // (p.crowd_count is all zero: th_draw.hip clears it when it is allocated, crowd_scan_kernel behind every count it reads)
    hipLaunchKernelGGL(crowd_hist_kernel, dim3(p.nlarge * kBinReplicas), dim3(256), 0, s, p);
    hipLaunchKernelGGL(crowd_scan_kernel, dim3(p.nlarge), dim3(256), 0, s, p);
    hipLaunchKernelGGL(crowd_scatter_kernel, dim3(p.nlarge * kBinReplicas), dim3(256), 0, s, p);
}

// the giants (runs of more than kGiantRun fragments): parted, ordered window by window, walked - a run of ten thousand
// fragments is applied one after the other for a few hundred microseconds: the caller puts this on a stream of its own beside
// launch_bins_blend (disjoint texels), so that the walk overlaps with everything else instead of following it.  Last, the
// runs giant_part_kernel left alone (more of one bucket than a window holds), a workgroup each.
void launch_bins_part_giants(const DepositParams &p, hipStream_t s)
{
    if (!p.nlarge) return;
    hipLaunchKernelGGL(giant_part_kernel, dim3(2048), dim3(256), 0, s, p);
}
void launch_bins_blend_giants(const DepositParams &p, hipStream_t s)
{
    if (!p.nlarge) return;
    hipLaunchKernelGGL(giant_sort_kernel, dim3(4096), dim3(256), 0, s, p);
#define TH_GO(M) do { hipLaunchKernelGGL((run_walk_kernel<M, true>), dim3(512), dim3(256), 0, s, p); \
                      hipLaunchKernelGGL(crowd_blend_kernel<M>, dim3(256), dim3(256), 0, s, p, (const uint32_t *)p.crowd_giant, (const uint32_t *)(p.totals + kTotGiant)); } while (0)
    if (p.mode == 0) TH_GO(0); else if (p.mode == 1) TH_GO(1); else TH_GO(2);
#undef TH_GO
}
// ... and the runs in between (kWaveRun + 1 .. kGiantRun fragments), ordered by a wave each, then walked by a wave per run and
// target - two launches the caller can put on different streams (disjoint texels again; th_draw.hip)
void launch_bins_sort_long(const DepositParams &p, hipStream_t s)
{
    if (!p.nlarge) return;
    hipLaunchKernelGGL(long_sort_kernel, dim3(1024), dim3(256), 0, s, p);
}
void launch_bins_walk_long(const DepositParams &p, hipStream_t s)
{
    if (!p.nlarge) return;
#define TH_GO(M) hipLaunchKernelGGL((run_walk_kernel<M, false>), dim3(2048), dim3(256), 0, s, p)
    if (p.mode == 0) TH_GO(0); else if (p.mode == 1) TH_GO(1); else TH_GO(2);
#undef TH_GO
}

void launch_bins_blend_crowd(const DepositParams &p, hipStream_t s)
{
    if (!p.nlarge) return;
    // every run of up to kWaveRun fragments ordered by a wave of its own ...  (Up to kGiantRun - the 257..1024 class through a
    // 16-keys-per-lane instantiation over the list - was slower: a lane walking a thousand fragments holds its wave: 2.60
    // against 2.28 ms per crowded draw.)
    hipLaunchKernelGGL(crowd_sort_kernel, dim3(p.nlarge * (kBinTexels / 16u)), dim3(256), 0, s, p);
    // ... and walked by a lane of its own
    const uint32_t walkers = ((p.nlarge + 7u) & ~7u) * 16u;       // (eight bins side by side, one per XCD)
    if (p.mode == 0) hipLaunchKernelGGL(crowd_walk_kernel<0>, dim3(walkers), dim3(64), 0, s, p);
    else if (p.mode == 1) hipLaunchKernelGGL(crowd_walk_kernel<1>, dim3(walkers), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(crowd_walk_kernel<2>, dim3(walkers), dim3(64), 0, s, p);
}

// the bins of up to kBinCap places, a workgroup each.  Needs nothing from the host: launched right behind launch_bins_fused,
// it covers the read-back of the pass's totals (the kernel returns at once when the pass raised a flag)
void launch_bins_blend(const DepositParams &p, hipStream_t s)
{
    if (p.mode == 0) hipLaunchKernelGGL(bins_blend_kernel<0>, dim3(p.nbins), dim3(256), 0, s, p);
    else if (p.mode == 1) hipLaunchKernelGGL(bins_blend_kernel<1>, dim3(p.nbins), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(bins_blend_kernel<2>, dim3(p.nbins), dim3(256), 0, s, p);
}
size_t crowd_words_per_bin() { return 7u * kBinTexels + 1u; }       // counts, cursors, starts (+ 1), the long list, the giants' list, the giants' windows (2)

}  // namespace th

#ifdef TH_BLEND_STAMPS
extern "C" int th_debug_blend_stamps(unsigned long long *out)       // (diagnostic builds: read and clear)
{
    const unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(th::g_blend_stamps), sizeof zero) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(th::g_blend_stamps), zero, sizeof zero) != hipSuccess;
}
#endif
