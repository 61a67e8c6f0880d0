// th_order.hip - the slot orders of the ring buffers (texel order or a tile-sorted order, th_kernels.hip "Tile-sorted slot
// order"): policy, storage, the moves between orders; and the captured th_step_n sequences that name ring buffers.
#include "th_ctx.hpp"

using namespace thi;

namespace thi {

// ---- captured th_step_n sequences -------------------------------------------------------------
void destroy_graph(GraphEntry &g)
{
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.times_dev) (void)hipFree(g.times_dev);
    if (g.times_host) (void)hipHostFree(g.times_host);
    if (g.copied) (void)hipEventDestroy(g.copied);
    g = GraphEntry{};
}

void clear_graphs(th_context *c)
{
    if (!c->graphs.empty() && c->stream) (void)hipStreamSynchronize(c->stream);
    for (GraphEntry &g : c->graphs) destroy_graph(g);
    c->graphs.clear();
}

// ---- slot order management ---------------------------------------------------------------------
// Policy.  Sorting the slots by flow tile pays when the random flow gather is what a step waits for: the decoded
// plane does not fit one XCD's 4 MiB L2 and there are enough particles to amortise the sort (measurements:
// profiles/r1_c_bucketing.txt, r2_b_*).  th_options::bucket = 0 / 1 forces the layout off / on (the parity suite reruns
// under 1); resort_steps / rebucket_steps set the re-sort period of single-step / fused launches.
uint32_t tile_count(const th_context *c, uint32_t *tiles_x)
{
    const uint32_t tx = ((uint32_t)c->fw + (1u << kTileShift) - 1u) >> kTileShift;
    const uint32_t ty = ((uint32_t)c->fh + (1u << kTileShift) - 1u) >> kTileShift;
    if (tiles_x) *tiles_x = tx;
    return tx * ty;
}
bool sorting_possible(const th_context *c)
{
    const size_t flow_texels = (size_t)c->fw * c->fh;
    if (c->texels() < 2 * flow_texels) return false;           // the decoded plane is not used at all
    if (2 * ((size_t)tile_count(c, nullptr) + 1) > th::kMaxTileBins) return false;     // two sort classes per tile (+ the no-tap pair)
    if (c->opt.bucket == 0) return false;
    if (c->opt.bucket == 1) return true;
    return c->texels() >= ((size_t)1 << 20) && flow_texels * sizeof(float2) > ((size_t)3 << 20);
}

// Which rows of the state texture can draw() make lines of, and does every vertex of every line read the line's OWN
// particle?  Particles.generateLUT writes the vertex coordinates as i/(W-1), j/(2H-1) (src/particles.js:171-190) and the
// shader turns them back into a texel and a buffer with fp32 arithmetic (src/state/state-at-frame.glsl:12-22): vertex
// 2m of line m reads `previous` in the lower rows and `current` in the upper ones, vertex 2m+1 `current` - so the lines
// of the upper half (both vertices the same texel of the same buffer) have no length; and for some shapes (W >= 8192;
// heights such as 100, 1080, 3000) the lookup of a few rows / columns lands one texel beside the line's own: those texels'
// rows and columns are tabled (th::LineSources), so that a draw over a slot order finds them (th_bins.hip).
// Same operations as dep_fetch (th_raster.hpp).  Bit m of the table: row m can draw.
th_status line_rows(th_context *c)
{
    if (c->d_row_draws) return TH_OK;
    const int W = c->cfg.width, H = c->cfg.global_height, row0 = c->cfg.row0, rows = c->cfg.height;
    const double inv_x = 1.0 / (double)((W > 2 ? W : 2) - 1), inv_y = 1.0 / (double)((2 * H > 2 ? 2 * H : 2) - 1);
    auto nearest = [](float u, int n) { const float f = floorf(u * (float)n); return !(f > 0.0f) ? 0 : (f > (float)(n - 1) ? n - 1 : (int)f); };
    // what the lookups of OTHER lines land on: source columns (of every row) and source rows (of this band); a band's own first
    // and last row too when rows are looked up across rows anywhere in the texture - the neighbouring bands' lines may want them
    std::vector<uint16_t> col_index((size_t)W, 0xffffu), row_index((size_t)rows, 0xffffu);
    uint32_t ncols = 0, nrows = 0;
    bool local = true, rows_drift = false, cross = false;
    for (int i = 0; i < W; ++i) {
        const int col = nearest((float)((double)i * inv_x), W);
        if (col != i) { local = false; if (col_index[(size_t)col] == 0xffffu) { col_index[(size_t)col] = (uint16_t)(ncols < 0xffffu ? ncols : 0xfffeu); ++ncols; } }
    }
    auto source_row = [&](int g) {
        if (g < row0 || g >= row0 + rows) return;
        if (row_index[(size_t)(g - row0)] == 0xffffu) { row_index[(size_t)(g - row0)] = (uint16_t)(nrows < 0xffffu ? nrows : 0xfffeu); ++nrows; }
    };
    std::vector<uint32_t> bits(((size_t)H + 31) / 32, 0u);
    for (int m = 0; m < H; ++m) {
        int row[2];
        bool cur[2];
        for (int v = 0; v < 2; ++v) {
            const float uvy = (float)((double)(2 * m + v) * inv_y), near_index = uvy * (float)H, fl = floorf(near_index);
            cur[v] = near_index - fl > 0.25f;
            row[v] = nearest(fl / (float)H, H);
            if (row[v] != m) {
                local = false; rows_drift = true;
                if (m >= row0 && m < row0 + rows) { source_row(row[v]); if (row[v] < row0 || row[v] >= row0 + rows) cross = true; }
            }
        }
        if (!(row[0] == row[1] && cur[0] == cur[1])) bits[(size_t)m >> 5] |= 1u << (m & 31);
    }
    if (rows_drift && rows != H) { source_row(row0); source_row(row0 + rows - 1); }
    TH_HIP(hipMalloc((void **)&c->d_row_draws, bits.size() * sizeof(uint32_t)));
    TH_HIP(hipMemcpy(c->d_row_draws, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->src_nrows = c->src_ncols = 0; c->rows_cross_bands = false;
    if (!local && nrows < 0xfffeu && ncols < 0xfffeu && (size_t)nrows * (size_t)W + (size_t)ncols * (size_t)rows < ((size_t)1 << 31)) {
        TH_HIP(hipMalloc((void **)&c->src_row_index, (size_t)rows * sizeof(uint16_t)));
        TH_HIP(hipMalloc((void **)&c->src_col_index, (size_t)W * sizeof(uint16_t)));
        TH_HIP(hipMemcpy(c->src_row_index, row_index.data(), (size_t)rows * sizeof(uint16_t), hipMemcpyHostToDevice));
        TH_HIP(hipMemcpy(c->src_col_index, col_index.data(), (size_t)W * sizeof(uint16_t), hipMemcpyHostToDevice));
        c->src_nrows = nrows; c->src_ncols = ncols; c->rows_cross_bands = rows_drift && rows != H;
        (void)cross;                       // (whether THIS band looks across its edge: every band of such a texture exchanges its edge rows)
        c->lines_local = 0;
    } else c->lines_local = local ? 1 : 2;   // 2: lookups beside the line's own texel, and too many to table - such shapes draw in texel order
    return TH_OK;
}

th::TileGeom tile_geom(const th_context *c, const th_logic_uniforms &u)
{
    th::TileGeom g{};
    g.width = (uint32_t)c->cfg.width; g.pow2w = is_pow2(g.width) ? 1u : 0u; g.log2w = g.pow2w ? ilog2(g.width) : 0u;
    g.row0 = (uint32_t)c->cfg.row0; g.row_draws = c->d_row_draws;
    g.view_x = u.viewSize[0]; g.view_y = u.viewSize[1];
    g.half_fw = 0.5f * (float)c->fw; g.half_fh = 0.5f * (float)c->fh;
    g.fwm1 = (float)(c->fw - 1); g.fhm1 = (float)(c->fh - 1);
    g.ntiles = tile_count(c, &g.tiles_x);
    return g;
}
bool same_geom(const th::TileGeom &a, const th::TileGeom &b) { return memcmp(&a, &b, sizeof a) == 0; }

int order_of(const th_context *c, const float4 *buf)
{
    for (const auto &e : c->buf_order) if (e.first == buf) return e.second;
    return -1;
}
void set_order(th_context *c, float4 *buf, int order)
{
    for (size_t k = 0; k < c->buf_order.size(); ++k)
        if (c->buf_order[k].first == buf) {
            --c->orders[(size_t)c->buf_order[k].second].refs;
            c->buf_order.erase(c->buf_order.begin() + (long)k);
            break;
        }
    if (order >= 0) { c->buf_order.emplace_back(buf, order); ++c->orders[(size_t)order].refs; }
}
bool any_sorted(const th_context *c) { return !c->buf_order.empty(); }

th_status sort_storage(th_context *c)
{
    if (th_status s = line_rows(c)) return s;
    if (c->tile_mem) return TH_OK;
    const size_t n = c->texels();
    TH_HIP(hipMalloc((void **)&c->spare, n * sizeof(float4)));
    TH_HIP(hipMalloc((void **)&c->tile_mem, (kTileWords + 8 + 2 * th::kMaxTileBins) * sizeof(uint32_t)));
    TH_HIP(hipMemsetAsync(c->tile_mem, 0, (kTileWords + 8 + 2 * th::kMaxTileBins) * sizeof(uint32_t), c->stream));
    TH_HIP(hipMalloc((void **)&c->block_records, ((n + th::kTileChunk - 1) / th::kTileChunk) * sizeof(th::ChunkRecord)));
    TH_HIP(hipHostMalloc((void **)&c->miss_host, 2 * sizeof(uint32_t)));
    c->miss_host[0] = c->miss_host[1] = 0;
    c->max_chunks = (uint32_t)(n / th::kTileChunk) + th::kMaxTileBins + 8u;
    return TH_OK;
}

// an order no ring buffer is stored in (allocates the first few)
th_status free_order(th_context *c, int *out)
{
    for (size_t k = 0; k < c->orders.size(); ++k) if (c->orders[k].refs == 0) { *out = (int)k; return TH_OK; }
    th_context::SlotOrder o;
    TH_HIP(hipMalloc((void **)&o.perm, c->texels() * sizeof(uint32_t)));
    TH_HIP(hipMalloc((void **)&o.chunks, (size_t)c->max_chunks * sizeof(th::TileChunk)));
    TH_HIP(hipMalloc((void **)&o.records, (size_t)c->max_chunks * sizeof(th::ChunkRecord)));
    TH_HIP(hipMalloc((void **)&o.nchunks, sizeof(uint32_t)));
    c->orders.push_back(o);
    *out = (int)c->orders.size() - 1;
    return TH_OK;
}

// ---- the re-sort of a frame loop, beside its draw() -----------------------------------------------------------------------
// A frame is step(); draw().  Re-sorting inside the steps cost two of every `resort_steps` frames 0.15 and 0.35 ms extra (a
// COUNT pass, then the scan and a SCATTER pass) - in a loop a host runs at a fixed rate, the frames that matter.  While draws
// are going on, the re-sort is a plain move instead, started right behind a step on the draw's side stream: histogram of the
// step's OUTPUT, scan, a copy of it in the new order.  The draw that follows reads the same buffer (nobody writes it) and its
// first 0.6 ms are the emit pass, which waits on atomics and leaves the memory system idle; the next step takes the copy for
// its input - same content, new slots - and the buffer it replaces becomes the next copy's destination.
// Anything that touches the source or the sort's scratch in between drops the copy (asort_drop); the steps then go on in the
// old order and the re-sort is tried again behind the next one.
th_status asort_drop(th_context *c)
{
    if (!c->asort.pending) return TH_OK;
    // (the side stream's kernels use the sort's scratch and the order's arrays: whoever comes next on the main stream waits)
    TH_HIP(hipStreamWaitEvent(c->stream, c->asort.done, 0));
    c->asort.pending = c->asort.valid = false;
    c->asort.src = nullptr;
    return TH_OK;
}

th_status asort_start(th_context *c, const th::TileGeom &g, float4 *src, int src_order)
{
    if (th_status s = sort_storage(c)) return s;
    if (!c->asort.dst) {
        TH_HIP(hipMalloc((void **)&c->asort.dst, c->texels() * sizeof(float4)));
        TH_HIP(hipEventCreateWithFlags(&c->asort.ready, hipEventDisableTiming));
        TH_HIP(hipEventCreateWithFlags(&c->asort.done, hipEventDisableTiming));
    }
    int order = -1;
    if (th_status s = free_order(c, &order)) return s;
    th_context::SlotOrder &o = c->orders[(size_t)order];
    o.geom = g; o.fw = c->fw; o.fh = c->fh;
    th::TileSortParams b{};
    b.state = src; b.perm_in = src_order >= 0 ? c->orders[(size_t)src_order].perm : nullptr; b.count = (uint32_t)c->texels();
    b.g = g;
    b.hist = c->tile_mem; b.cursor = c->tile_mem + kTileWords / 2;
    b.totals = c->tile_mem + kTileWords + 8; b.starts = b.totals + th::kMaxTileBins;
    b.chunks = o.chunks; b.nchunks = o.nchunks;
    b.perm_out = o.perm;
    b.block_records = c->block_records;
    b.state_out = c->asort.dst;
    TH_HIP(hipEventRecord(c->asort.ready, c->stream));
    TH_HIP(hipStreamWaitEvent(c->side, c->asort.ready, 0));
    TH_HIP(hipMemsetAsync(b.hist, 0, kTileWords / 2 * sizeof(uint32_t), c->side));
    th::launch_tile_hist(b, c->side);
    th::launch_tile_scan(b, c->side);
    th::launch_tile_scatter(b, c->side);
    TH_HIP(hipGetLastError());
    TH_HIP(hipEventRecord(c->asort.done, c->side));
    ++c->sorts;
    o.stamp = c->sorts;
    c->counted.buf = nullptr;
    c->asort.pending = c->asort.valid = true;
    c->asort.src = src; c->asort.src_order = src_order; c->asort.order = order; c->asort.at_step = c->total_steps;
    return TH_OK;
}

// every ring buffer back to texel order (reports whether anything was launched)
th_status ensure_identity(th_context *c, bool *launched)
{
    if (launched) *launched = false;
    c->counted.buf = nullptr;
    if (!any_sorted(c)) return TH_OK;
    clear_graphs(c);                       // captured sequences name the ring buffers that are swapped below
    for (float4 *&b : c->ring) {
        const int o = order_of(c, b);
        if (o < 0) continue;
        th::launch_unpermute_state(c->spare, b, c->orders[(size_t)o].perm, (uint32_t)c->texels(), c->packed, c->stream);
        set_order(c, b, -1);
        state_moved(c, b, c->spare);
        float4 *t = b; b = c->spare; c->spare = t;
        if (launched) *launched = true;
    }
    TH_HIP(hipGetLastError());
    return TH_OK;
}

// Count the tiles of `state` (any slot order) and lay out a new order for it: tile starts, rank cursors, chunk table.
// The slots themselves are assigned by the kernel that moves the state (tile_scatter_kernel or a SCATTER step).
th_status begin_sort(th_context *c, const th::TileGeom &g, const float4 *state, const uint32_t *perm_in, int *order,
                     th::TileSortParams *params, bool have_hist)
{
    if (th_status s = sort_storage(c)) return s;
    if (th_status s = asort_drop(c)) return s;          // (one sort at a time: the scratch and the free orders are shared)
    if (th_status s = free_order(c, order)) return s;
    th_context::SlotOrder &o = c->orders[(size_t)*order];
    o.geom = g; o.fw = c->fw; o.fh = c->fh;
    th::TileSortParams b{};
    b.state = state; b.perm_in = perm_in; b.count = (uint32_t)c->texels();
    b.g = g;
    b.hist = c->tile_mem; b.cursor = c->tile_mem + kTileWords / 2;
    b.totals = c->tile_mem + kTileWords + 8; b.starts = b.totals + th::kMaxTileBins;
    b.chunks = o.chunks; b.nchunks = o.nchunks;
    b.perm_out = o.perm;
    b.block_records = have_hist ? nullptr : c->block_records;      // (only a tile_hist pass over the same blocks fills them)
    b.packed = c->packed ? 1u : 0u;
    if (!have_hist) {          // (a COUNT pass whose histogram was never used may have left counts behind)
        TH_HIP(hipMemsetAsync(b.hist, 0, kTileWords / 2 * sizeof(uint32_t), c->stream));
        th::launch_tile_hist(b, c->stream);
    }
    th::launch_tile_scan(b, c->stream);
    TH_HIP(hipMemsetAsync(c->tile_mem + kTileWords, 0, sizeof(uint32_t), c->stream));    // window misses
    TH_HIP(hipGetLastError());
    c->steps_since_sort = 0;
    ++c->sorts;
    o.stamp = c->sorts;
    c->counted.buf = nullptr;
    if (params) *params = b;
    return TH_OK;
}

// ring[1] into ring[0]'s slot order (through texel order): only when a draw meets the two in different orders - a
// re-sorting step moves its input along with its output while draws are going on (enqueue_step)
th_status align_slot_orders(th_context *c)
{
    const int o0 = order_of(c, c->ring[0]), o1 = order_of(c, c->ring[1]);
    if (o0 == o1) return TH_OK;
    if (th_status s = sort_storage(c)) return s;
    clear_graphs(c);
    float4 *&b = c->ring[1];
    if (o1 >= 0) {
        th::launch_unpermute_state(c->spare, b, c->orders[(size_t)o1].perm, (uint32_t)c->texels(), c->packed, c->stream);
        set_order(c, b, -1);
        state_moved(c, b, c->spare);
        float4 *t = b; b = c->spare; c->spare = t;
    }
    if (o0 >= 0) {
        th::launch_permute_state(c->spare, b, c->orders[(size_t)o0].perm, (uint32_t)c->texels(), c->packed, c->stream);
        state_moved(c, b, c->spare);
        float4 *t = b; b = c->spare; c->spare = t;
        set_order(c, b, o0);
    }
    TH_HIP(hipGetLastError());
    c->counted.buf = nullptr;
    return TH_OK;
}

}  // namespace thi

extern "C" {

th_status th_slot_order(th_context *c, th_slot_order_info *out)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(out, "null output");
    *out = th_slot_order_info{};
    out->sorted_buffers = (int32_t)c->buf_order.size();
    out->steps_since_sort = c->steps_since_sort;
    out->sorts = c->sorts;
    if (c->tile_mem) {
        uint32_t m = 0;
        TH_HIP(hipMemcpyAsync(&m, c->tile_mem + kTileWords, sizeof m, hipMemcpyDeviceToHost, c->stream));
        TH_HIP(hipStreamSynchronize(c->stream));
        out->window_misses = m;
    }
    return TH_OK;
}

}  // extern "C"
