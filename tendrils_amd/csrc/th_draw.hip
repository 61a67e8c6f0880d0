// th_draw.hip - Tendrils.draw() (src/index.js:278-337): the flow pass, the view pass, both in one call, the trail export;
// which pipeline draws (binned over sorted slots, th_bins.hip / stream-ordered in texel order, th_deposit.hip + th_sort.hip).
#include "th_ctx.hpp"
#include <chrono>

using namespace thi;

namespace thi {

// ---- flow deposit ------------------------------------------------------------------------------------------
int deposit_texel_bits(const th_context *c)
{
    const uint64_t texels = (uint64_t)c->fw * c->fh;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < texels) ++bits;
    return bits;
}

// ---- which pipeline draws -------------------------------------------------------------------------------------
// The binned pipeline (th_bins.hip) walks the particles by slot, in whatever order the ring is held; the stream-ordered one
// (th_deposit.hip) needs texel order.  th_draw_pipeline (or TH_DRAW=bins / stream at th_create) forces one; by default the binned pipeline draws
// whenever the integrator would step over tile-sorted slots (sorting_possible): a step() + draw() frame loop then never
// leaves the sorted order.
// Which pipeline a draw pass takes.  auto: wherever the integrator steps over tile-sorted slots the frame loop - step(); draw() -
// stays on them: the binned pipeline takes particles in any order, and it is ahead of the stream-ordered one from the first
// frame (1.3 against 2.2 ms per draw with both passes at C3) to the crowded target the wake leaves after a thousand frames
// (70-77 % of all fragments in bins of more than 4096, in texels with thousands of them: 1.3-1.7 against 1.7-2.3 ms;
// profiles/r4_g_giants.txt).  (Round 3 handed a crowded target over to the stream-ordered pipeline for a spell: it was level
// then.)  A binned pass that cannot go on - no memory for its store - is repeated in stream order, and the ring then keeps
// texel order for opt.rebucket_steps steps (deposit_prepare): that many frames pass before the bins are tried again.

float drawn_line_width(const th_context *c, int pass)
{
    const float w = c->line_width[pass];
    return w < c->line_range[0] ? c->line_range[0] : (w > c->line_range[1] ? c->line_range[1] : w);
}

static bool draw_uses_bins(th_context *c)
{
    const int policy = c->draw_pipeline != TH_DRAW_AUTO ? c->draw_pipeline : c->opt.draw;
    if (policy == 0) return false;
    if (c->cfg.height != c->cfg.global_height || c->fw > th::kBinsMaxExtent || c->fh > th::kBinsMaxExtent) return false;
    if (policy == 1) return true;
    // (lines wider than 2 cover more texels than a line's record holds: nearly all of them would leave the fused pass
    // for the long list, one atomic per fragment - the stream-ordered pipeline counts and scans instead)
    if (drawn_line_width(c, TH_PASS_FLOW) > 2.0f || drawn_line_width(c, TH_PASS_VIEW) > 2.0f) return false;
    return sorting_possible(c);
}

// per-line buffers + parameters.  want_bins: the caller can run the binned pipeline (*bins tells whether it will)
static th_status prepare_pass(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, bool use_bins);

// Does a sharded job draw through the bins?  Only what every rank sees alike may enter: the job's shapes, the lines' widths,
// the switches (set alike on all ranks) - not this band's size, not how crowded its last draw was.
bool binned_shards(const th_context *c)
{
    const int policy = c->draw_pipeline != TH_DRAW_AUTO ? c->draw_pipeline : c->opt.draw;
    if (policy == 0 || (c->lines_local != 1 && c->lines_local != 0)) return false;      // (2: lookups off the line's own texel that no table holds)
    if (c->fw > th::kBinsMaxExtent || c->fh > th::kBinsMaxExtent) return false;
    if (drawn_line_width(c, TH_PASS_FLOW) > 2.0f || drawn_line_width(c, TH_PASS_VIEW) > 2.0f) return false;
    if (policy == 1 || c->opt.bucket == 1) return true;
    if (c->opt.bucket == 0) return false;
    // (sorting_possible's rule on the job's average band: the integrator steps over sorted slots there)
    const size_t flow_texels = (size_t)c->fw * c->fh, per_rank = (size_t)c->cfg.width * c->cfg.global_height / (size_t)std::max(c->comm_world, 1);
    return per_rank >= ((size_t)1 << 20) && per_rank >= 2 * flow_texels && flow_texels * sizeof(float2) > ((size_t)3 << 20);
}

// the binned pass over a band's slots in whatever order they are held (row-band shards: th_shard.hip)
th_status deposit_prepare_bins(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p)
{
    TH_REQUIRE(u, "null uniforms");
    TH_REQUIRE(c->ring.size() >= 2, "draw needs at least 2 state buffers (have %zu)", c->ring.size());
    if (any_sorted(c)) if (th_status s = align_slot_orders(c)) return s;
    c->last_binned_draw = c->total_steps;
    return prepare_pass(c, u, p, true);
}

th_status deposit_prepare(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, bool want_bins, bool *bins)
{
    TH_REQUIRE(u, "null uniforms");
    TH_REQUIRE(c->ring.size() >= 2, "draw needs at least 2 state buffers (have %zu)", c->ring.size());
    // The auto policy counts FRAMES, and the passes of one frame - th_flow_deposit then th_view_draw, or two widths - take one
    // pipeline: a view pass that left the slot order its flow pass had drawn over would throw the order away in mid-frame.
    if (want_bins && c->draw_frame_step != c->total_steps) { ++c->draws; c->draw_frame_step = c->total_steps; c->frame_bins = -1; }
    bool use_bins = want_bins && (c->frame_bins >= 0 && c->draw_pipeline == TH_DRAW_AUTO ? c->frame_bins == 1 : draw_uses_bins(c));
    if (use_bins) {
        // the binned pipeline reads a line's vertices from the line's own slot - or, for the shapes whose vertex lookup lands on
        // another particle, through the table of where those particles lie in the slot order (line_rows, th::LineSources); the
        // few shapes with more such rows / columns than the table holds keep to the stream-ordered pipeline in texel order
        if (th_status s = line_rows(c)) return s;
        if (c->lines_local != 1 && c->lines_local != 0) use_bins = false;
        else if (any_sorted(c)) { if (th_status s = align_slot_orders(c)) return s; }
    }
    if (bins) *bins = use_bins;
    if (want_bins) c->frame_bins = use_bins ? 1 : 0;
    if (use_bins) c->last_binned_draw = c->total_steps;
    else {
        if (th_status s = ensure_identity(c)) return s;      // the vertex stream addresses particles in texel order
        c->hold_texel_order_until = c->total_steps + c->opt.rebucket_steps;   // a frame loop of step + draw stays in texel order
    }
    return prepare_pass(c, u, p, use_bins);
}

// per-line buffers + parameters of a pass whose pipeline is decided
static th_status prepare_pass(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, bool use_bins)
{
    const size_t lines = c->texels();
    TH_REQUIRE((size_t)c->fw * c->fh > 0 && (uint64_t)c->cfg.width * c->cfg.global_height < (1ull << 32), "bad shapes");
    if (c->dep_lines != lines) {
        TH_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->dep_count); (void)hipFree(c->dep_offset); (void)hipFree(c->dep_blocks); (void)hipFree(c->dep_record);
        (void)hipFree(c->dep_lists);
        c->dep_count = c->dep_offset = c->dep_blocks = c->dep_lists = nullptr; c->dep_record = nullptr;
        TH_HIP(hipMalloc((void **)&c->dep_count, lines * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->dep_offset, lines * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->dep_record, 2 * lines * sizeof(uint4)));
        TH_HIP(hipMalloc((void **)&c->dep_lists, th::deposit_list_words((uint32_t)c->cfg.width, (uint32_t)c->cfg.height, &c->dep_list_cap) * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->dep_blocks, (size_t)th::deposit_scan_words((uint32_t)c->cfg.width, (uint32_t)c->cfg.height) * sizeof(uint32_t)));
        if (!c->dep_total) TH_HIP(hipMalloc((void **)&c->dep_total, th::kTotWords * sizeof(uint32_t)));      // [0] total, [1] out-of-band flag, [2] largest bin, [3] large bins, [4] their blocks
        c->dep_lines = lines;
    }
    p = th::DepositParams{};
    // a packed ring is read in place, texel by texel, as what the stored texels decode to (dep_state: no f32 copy of the ring
    // per draw - two passes over 10.7 GB at config 5's size)
    p.cur = c->ring[0]; p.prev = c->ring[1];
    p.packed = c->packed ? 1u : 0u;
    p.flow = c->flow;
    p.W = (uint32_t)c->cfg.width; p.H = (uint32_t)c->cfg.global_height;
    p.row0 = (uint32_t)c->cfg.row0; p.rows = (uint32_t)c->cfg.height;
    p.fw = c->fw; p.fh = c->fh;
    p.view_x = u->viewSize[0]; p.view_y = u->viewSize[1]; p.time = u->time; p.speed_limit = u->speedLimit;
    p.line_half = 0.5f * drawn_line_width(c, TH_PASS_FLOW);       // (view_params: the view pass's)
    {
        const int lw = c->cfg.width > 2 ? c->cfg.width : 2, lh = 2 * c->cfg.global_height > 2 ? 2 * c->cfg.global_height : 2;
        p.inv_x = 1.0 / (double)(lw - 1); p.inv_y = 1.0 / (double)(lh - 1);
    }
    p.count = c->dep_count; p.offset = c->dep_offset; p.record = c->dep_record; p.oob = c->dep_total + 1;
    p.list_n = c->dep_lists; p.list_cap = c->dep_list_cap;
    {
        uint32_t cap = 0;
        p.lists = c->dep_lists + (th::deposit_list_words((uint32_t)c->cfg.width, (uint32_t)c->cfg.height, &cap) - (size_t)3 * 64 * cap);      // (three kinds of list, 64 segments each: th_raster.hpp)
    }
    p.halo_lo = c->halo_lo; p.halo_hi = c->halo_hi;
    if (!use_bins) TH_HIP(hipMemsetAsync(c->dep_total, 0, th::kTotWords * sizeof(uint32_t), c->stream));      // (bins: launch_bins_fused)
    if (th_status s = line_rows(c)) return s;
    p.row_draws = c->d_row_draws;
    if (c->lines_local == 0) {          // (the tables; `slot` stays null in texel order: a texel's slot is its index)
        p.src.row_index = c->src_row_index; p.src.col_index = c->src_col_index; p.src.nrows = c->src_nrows; p.src.ncols = c->src_ncols;
    }
    if (use_bins) {
        const int o = order_of(c, c->ring[0]);
        p.perm = o >= 0 ? c->orders[(size_t)o].perm : nullptr;
        p.bins_x = ((uint32_t)c->fw + (1u << th::kBinShift) - 1u) >> th::kBinShift;
        p.nbins = p.bins_x * (((uint32_t)c->fh + (1u << th::kBinShift) - 1u) >> th::kBinShift);
        if (c->bin_capacity < p.nbins) {
            TH_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->bin_mem); c->bin_mem = nullptr; c->bin_capacity = 0;
            (void)hipFree(c->chunk_table); c->chunk_table = nullptr;
            const size_t stride = ((size_t)p.nbins + 255) / 256 * 256 + 64;      // (the lists' cursors of one bin on different memory channels)
            TH_HIP(hipMalloc((void **)&c->bin_mem, (th::kBinReplicas * stride + 2 * (size_t)p.nbins + 2) * sizeof(uint32_t)));
            // (TH_OPT_BINS_PAGES: how far a list can grow at first - tests make it small to run the widening path)
            const int first = c->opt.bins_pages;
            c->bin_max_pages = first > 0 ? (uint32_t)first : (first < 0 ? (uint32_t)-first : th::kBinFirstPages);
            const size_t table = (size_t)p.nbins * th::kBinReplicas * c->bin_max_pages * sizeof(uint32_t);
            TH_HIP(hipMalloc((void **)&c->chunk_table, table));
            TH_HIP(hipMemsetAsync(c->chunk_table, 0, table, c->stream));        // (every reader of a list leaves its entries empty)
            c->bin_capacity = p.nbins;
        }
        p.bin_stride = (uint32_t)(((size_t)c->bin_capacity + 255) / 256 * 256 + 64);
        p.bin_cursor = c->bin_mem; p.large_bins = c->bin_mem + (size_t)th::kBinReplicas * p.bin_stride;
        p.large_key0 = p.large_bins + c->bin_capacity;
        p.page_table = c->chunk_table; p.max_pages = c->bin_max_pages;
        p.totals = c->dep_total; p.totals_host = nullptr; p.totals_seq = 0;       // (bins_expect, before the plan's kernels go out)
        // the blocks of slots with something to draw: listed once per slot order (one read-back per re-sort)
        const unsigned long long stamp = o >= 0 ? c->orders[(size_t)o].stamp : 0ull;
        // what the step that wrote these two buffers saw of their lines (th_step.hip): only for exactly this pair, order and view
        // (... and only where a line's two ends ARE the slot's own particle, now and a step ago: lines_local)
        if (c->opt.skip_unseen && c->lines_local == 1 && c->seen.bytes && c->seen.cur == c->ring[0] && c->seen.prev == c->ring[1] && p.cur == c->ring[0] && p.prev == c->ring[1] &&
            c->seen.order == o && c->seen.stamp == stamp && c->seen.view_x == p.view_x && c->seen.view_y == p.view_y && c->seen.fw == c->fw && c->seen.fh == c->fh &&
            drawn_line_width(c, TH_PASS_FLOW) <= 2.0f && drawn_line_width(c, TH_PASS_VIEW) <= 2.0f)
            p.block_seen = reinterpret_cast<const uint32_t *>(c->seen.bytes);
        if (!c->draw_blocks || c->draw_blocks_order != o || c->draw_blocks_stamp != stamp) {
            const size_t blocks = (c->texels() + 255) / 256;
            if (!c->draw_blocks) {
                TH_HIP(hipMalloc((void **)&c->draw_blocks, (blocks + 1) * sizeof(uint32_t)));
                TH_HIP(hipMalloc((void **)&c->draw_block_flags, blocks));
                if (c->lines_local == 0)
                    TH_HIP(hipMalloc((void **)&c->src_slots, ((size_t)c->src_nrows * c->cfg.width + (size_t)c->src_ncols * c->cfg.height + 1) * sizeof(uint32_t)));
            }
            th::launch_bins_block_list(p, c->draw_block_flags, c->draw_blocks + 1, c->draw_blocks, o >= 0 ? c->src_slots : nullptr, c->stream);
            TH_HIP(hipGetLastError());
            if (th_status s = read_back(c, &c->draw_nblocks, c->draw_blocks, sizeof(uint32_t))) return s;
            c->draw_blocks_order = o; c->draw_blocks_stamp = stamp;
        }
        p.draw_blocks = c->draw_blocks + 1; p.draw_nblocks = c->draw_nblocks;
        if (c->lines_local == 0 && o >= 0) p.src.slot = c->src_slots;
    }
    return TH_OK;
}

// scan of p.count (filled by the caller's marking pass) -> p.offset, total (one sync); reports band violations
th_status deposit_scan_total(th_context *c, const th::DepositParams &p, uint32_t *total)
{
    th::launch_deposit_scan(p, c->dep_blocks, c->dep_total, c->stream);
    uint32_t host[2] = {0, 0};
    if (th_status s = read_back(c, host, c->dep_total, sizeof host)) return s;
    if (host[1]) return fail(TH_ERR_UNSUPPORTED, "a line of this row band looks up a particle row outside the band (rows %d..%d of %d) and no halo row was supplied (th_deposit_set_halo)", c->cfg.row0, c->cfg.row0 + c->cfg.height, c->cfg.global_height);
    if (host[0] >= (1u << 31)) return fail(TH_ERR_UNSUPPORTED, "too many fragments for one draw (2^31 or more)");
    *total = host[0];
    return TH_OK;
}

// counts this context's fragments
th_status deposit_count(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, uint32_t *total)
{
    if (th_status s = deposit_prepare(c, u, p)) return s;
    th::launch_deposit_count(p, c->stream);
    return deposit_scan_total(c, p, total);
}

// per-fragment buffers for `total` fragments (grow-only)
th_status deposit_reserve(th_context *c, uint32_t total, bool wide, bool pairs)
{
    if (pairs && !c->dep_pairs) {                 // two varyings per fragment: the colour buffers at twice the size
        (void)hipFree(c->dep_colors); c->dep_colors = nullptr;
        (void)hipFree(c->dep_colors_sorted); c->dep_colors_sorted = nullptr;
        if (c->dep_capacity) TH_HIP(hipMalloc((void **)&c->dep_colors, 2 * c->dep_capacity * sizeof(float4)));
        c->dep_pairs = true;
    }
    if (c->dep_capacity < total) {
        for (uint32_t *&q : c->dep_u32) { (void)hipFree(q); q = nullptr; }
        for (unsigned long long *&q : c->dep_u64) { (void)hipFree(q); q = nullptr; }
        (void)hipFree(c->dep_colors); c->dep_colors = nullptr;
        (void)hipFree(c->dep_colors_sorted); c->dep_colors_sorted = nullptr;
        c->dep_capacity = 0; c->dep_wide = false;
        const size_t cap = (size_t)total + (size_t)total / 4 + 1024;
        for (uint32_t *&q : c->dep_u32) TH_HIP(hipMalloc((void **)&q, cap * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->dep_colors, (c->dep_pairs ? 2 : 1) * cap * sizeof(float4)));
        c->dep_capacity = cap;
    }
    if (!c->dep_colors_sorted) TH_HIP(hipMalloc((void **)&c->dep_colors_sorted, (c->dep_pairs ? 2 : 1) * c->dep_capacity * sizeof(float4)));
    if (wide && !c->dep_wide) {
        for (unsigned long long *&q : c->dep_u64) TH_HIP(hipMalloc((void **)&q, c->dep_capacity * sizeof(unsigned long long)));
        c->dep_wide = true;
    }
    return TH_OK;
}

th_status deposit_temp(th_context *c, size_t need)
{
    if (c->dep_temp_bytes < need) {
        (void)hipFree(c->dep_temp); c->dep_temp = nullptr; c->dep_temp_bytes = 0;
        TH_HIP(hipMalloc(&c->dep_temp, need + need / 4));
        c->dep_temp_bytes = need + need / 4;
    }
    return TH_OK;
}

// ---- view pass ---------------------------------------------------------------------------------------------
// The screen image and Tendrils.buffers at the target's shape (Tendrils.resize gives every buffer viewRes, src/index.js:404:
// a resized FBO starts empty); c->view = the bound one.
th_status view_storage(th_context *c)
{
    const bool shaped = c->view_w == c->fw && c->view_h == c->fh;
    if (c->view && shaped && (int32_t)c->view_ring.size() == c->view_buffers) return TH_OK;
    const size_t bytes = (size_t)c->fw * c->fh * sizeof(uchar4);
    auto fresh = [&](uchar4 **img) -> th_status {
        TH_HIP(hipMalloc((void **)img, bytes));
        TH_HIP(hipMemsetAsync(*img, 0, bytes, c->stream));       // a fresh drawing buffer is transparent black
        return TH_OK;
    };
    const uchar4 *was = c->view;
    int32_t bound = was && was != c->view_screen ? -2 : -1;      // (-2: a buffer that may be gone in a moment)
    for (size_t k = 0; k < c->view_ring.size(); ++k) if (c->view_ring[k] == was) bound = (int32_t)k;
    TH_HIP(hipStreamSynchronize(c->stream));
    c->view = nullptr;
    if (!shaped) {
        (void)hipFree(c->view_screen); c->view_screen = nullptr;
        for (uchar4 *&b : c->view_ring) { (void)hipFree(b); b = nullptr; }
        c->view_w = c->view_h = 0;
    }
    while ((int32_t)c->view_ring.size() > c->view_buffers) { (void)hipFree(c->view_ring.back()); c->view_ring.pop_back(); }
    while ((int32_t)c->view_ring.size() < c->view_buffers) c->view_ring.push_back(nullptr);
    if (!c->view_screen) if (th_status s = fresh(&c->view_screen)) return s;
    for (uchar4 *&b : c->view_ring) if (!b) if (th_status s = fresh(&b)) return s;
    c->view_w = c->fw; c->view_h = c->fh;
    c->view = bound >= 0 && bound < (int32_t)c->view_ring.size() ? c->view_ring[(size_t)bound] : c->view_screen;   // (a bound buffer that was removed: the screen)
    return TH_OK;
}

void view_fields(th_context *c, const th_render_uniforms *u, th::DepositParams &p)
{
    p.flow_decay = u->flowDecay; p.speed_alpha = u->speedAlpha; p.colormap_alpha = u->colorMapAlpha; p.sin_term = u->sinTerm;
    for (int k = 0; k < 4; ++k) { p.base_color[k] = u->baseColor[k]; p.flow_color[k] = u->flowColor[k]; }
    p.colormap = c->colormap; p.cw = c->cmap_w; p.ch = c->cmap_h;
}

th_status view_params(th_context *c, const th_render_uniforms *u, th::DepositParams &p, bool want_bins, bool *bins)
{
    TH_REQUIRE(u, "null uniforms");
    th_deposit_uniforms d{};
    d.viewSize[0] = u->viewSize[0]; d.viewSize[1] = u->viewSize[1]; d.time = u->time; d.speedLimit = u->speedLimit;
    if (th_status s = deposit_prepare(c, &d, p, want_bins, bins)) return s;
    p.mode = 1;
    p.line_half = 0.5f * drawn_line_width(c, TH_PASS_VIEW);
    view_fields(c, u, p);
    return TH_OK;
}

}  // namespace thi

static th_status export_run(th_context *c, th::DepositParams &p, float *lines, uint64_t capacity, uint64_t *count)
{
    th::launch_export_mark(p, c->stream);
    uint32_t total = 0;
    if (th_status s = deposit_scan_total(c, p, &total)) return s;
    *count = total;
    if (!lines || total == 0) return TH_OK;                  // size query
    TH_REQUIRE(capacity >= total, "line buffer holds %llu of %u lines", (unsigned long long)capacity, total);
    float *d_out = nullptr;
    TH_HIP(hipMalloc((void **)&d_out, (size_t)total * 12 * sizeof(float)));
    th::launch_export_write(p, d_out, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(lines, d_out, (size_t)total * 12 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    TH_HIP(e);
    return TH_OK;
}

// the fragments of the (prepared) pass `p`: count, emit, sort by texel, blend
static th_status deposit_run(th_context *c, th::DepositParams &p, uint64_t *fragments)
{
    // Same state, same view, same resolution as the pass before (the view pass after the flow pass of one draw()): the
    // lines cover the same texels in the same order - counts, offsets, records and the sorted order of the fragments are
    // still there, only the varyings differ.  (TH_DRAW_REUSE=0: every pass on its own.)
    const bool reuse = c->opt.draw_reuse && p.mode != 2 && c->drawn.valid && !c->drawn.binned && c->drawn.view_x == p.view_x && c->drawn.view_y == p.view_y &&
                       c->drawn.line_half == p.line_half;
    uint32_t total = 0;
    if (reuse) total = c->drawn.total;
    else {
        c->drawn.valid = false;
        th::launch_deposit_count(p, c->stream);
        if (th_status s = deposit_scan_total(c, p, &total)) return s;
    }
    if (fragments) *fragments = total;
    c->last_draw.pipeline = TH_DRAW_STREAM; c->last_draw.fragments = total; c->last_draw.crowded_fragments = 0;
    if (total == 0) return TH_OK;
    if (!reuse) if (th_status s = deposit_reserve(c, total, false, p.mode == 2)) return s;
    p.keys = c->dep_u32[0]; p.slots = c->dep_u32[1]; p.keys_sorted = c->dep_u32[2]; p.slots_sorted = c->dep_u32[3];
    p.colors = c->dep_colors; p.colors_sorted = c->dep_colors_sorted;
    if (reuse) {
        p.keys = nullptr;                        // (the keys are where the sort left them: only the varyings are written)
        if (c->drawn.sorted_in_a) { p.keys_sorted = c->dep_u32[0]; p.slots_sorted = c->dep_u32[1]; }
        th::launch_deposit_scatter(p, c->stream);
    } else {
        const int bits = th::deposit_key_bits(p);
        if (th_status s = deposit_temp(c, th::radix_sort_temp_bytes(total, 0, bits))) return s;
        th::launch_deposit_scatter(p, c->stream);
        const bool in_a = th::launch_radix_sort_u32(p.keys, p.slots, p.keys_sorted, p.slots_sorted, total, 0, bits, c->dep_temp, true, c->stream) == 0;
        if (in_a) { p.keys_sorted = c->dep_u32[0]; p.slots_sorted = c->dep_u32[1]; }        // an even number of passes ends in the (a) buffers
        c->drawn.valid = true; c->drawn.binned = false; c->drawn.view_x = p.view_x; c->drawn.view_y = p.view_y; c->drawn.line_half = p.line_half; c->drawn.total = total; c->drawn.sorted_in_a = in_a;
    }
    th::launch_deposit_blend(p, total, c->stream);
    TH_HIP(hipGetLastError());
    return TH_OK;
}

// the chunk store of the binned pipeline: nbins + pool chunks of keys (all empty) and varyings
// The varyings are sized by the pass that needs them (one float4 per place; two once a th_draw has run) and the pool starts
// at a quarter of the bins' first pages: the growth path is there (deposit_run_bins repeats a pass whose pool ran dry), and
// at 1920 x 1080 the store is 0.9 GB for a flow-only host, 1.4 GB with both passes - not 2.7 GB up front.  A store that
// cannot be allocated (a small device, many contexts) leaves the draw to the stream-ordered pipeline.
static th_status bins_store(th_context *c, uint32_t nbins, uint32_t pool, bool pairs)
{
    if (c->bins_keys && c->bins_store_bins == nbins && c->bins_pool >= pool && (c->bins_pairs || !pairs)) return TH_OK;
    TH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->bins_keys); (void)hipFree(c->bins_colors);
    c->bins_keys = nullptr; c->bins_colors = nullptr;
    pool = pool > c->bins_pool ? pool : c->bins_pool;
    pairs = pairs || c->bins_pairs;
    c->bins_pool = 0; c->bins_store_bins = 0;
    const size_t places = ((size_t)nbins * th::kBinReplicas + pool) * th::kBinPage;
    TH_REQUIRE(places < ((size_t)1 << 32), "the binned draw's chunk store would hold 2^32 places or more");
    if (hipMalloc((void **)&c->bins_keys, places * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&c->bins_colors, places * (pairs ? 2 : 1) * sizeof(float4)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(c->bins_keys); c->bins_keys = nullptr; c->bins_colors = nullptr;
        return kRetryInStreamOrder;
    }
    TH_HIP(hipMemsetAsync(c->bins_keys, 0xff, places * sizeof(unsigned long long), c->stream));
    c->bins_pool = pool; c->bins_store_bins = nbins; c->bins_pairs = pairs;
    return TH_OK;
}

constexpr double kEarlyBlendShare = 0.5;            // (of a draw's fragments in crowded bins: see deposit_run_bins)
// ---- the binned pipeline (th_bins.hip) over the (prepared) pass `p`, in three parts ------------------------------------------
namespace thi {

th_status bins_streams(th_context *c)
{
    if (c->side) return TH_OK;
    TH_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    TH_HIP(hipEventCreateWithFlags(&c->forked, hipEventDisableTiming));
    TH_HIP(hipEventCreateWithFlags(&c->joined, hipEventDisableTiming));
    TH_HIP(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
    TH_HIP(hipEventCreateWithFlags(&c->joined2, hipEventDisableTiming));
    TH_HIP(hipEventCreateWithFlags(&c->regrouped, hipEventDisableTiming));
    TH_HIP(hipHostMalloc((void **)&c->bins_totals_host, (th::kTotWords + 1) * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));
    memset(c->bins_totals_host, 0, (th::kTotWords + 1) * sizeof(uint32_t));
    TH_HIP(hipHostGetDevicePointer((void **)&c->bins_totals_dev, c->bins_totals_host, 0));
    return TH_OK;
}

// the pass's totals to the host over the side stream (behind `forked`, recorded by the caller on the main stream)
// crowd_plan_kernel, the last kernel of a pass's plan, has been launched with p.totals_host / p.totals_seq (bins_expect): wait
// for the sequence number to arrive behind the totals.  (Polled: the device writes host memory; a copy on the side stream had
// to be scheduled behind an event, run and be signalled before the host woke up - 43 us in which nothing ran.)
static void bins_expect(th_context *c, th::DepositParams &p)
{
    p.totals_host = c->bins_totals_dev; p.totals_seq = ++c->totals_seq;
    if (p.totals_seq == 0u) p.totals_seq = ++c->totals_seq;          // (0: what the word holds before the first pass)
}
static th_status bins_totals(th_context *c, const th::DepositParams &p)
{
    volatile uint32_t *seq = c->bins_totals_host + th::kTotWords;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; __atomic_load_n(seq, __ATOMIC_ACQUIRE) != p.totals_seq; ++spins) {
        if ((spins & 1023u) == 1023u) {
            // a kernel that faulted never writes the word: ask the runtime now and then, give up after a long while
            const hipError_t e = hipStreamQuery(c->stream);
            if (e != hipSuccess && e != hipErrorNotReady) return fail(TH_ERR_HIP, "the binned pass failed: %s", hipGetErrorString(e));
            if (e == hipSuccess && __atomic_load_n(seq, __ATOMIC_ACQUIRE) != p.totals_seq) {
                // (the stream is done and the word is not there: read the totals the slow way)
                TH_HIP(hipMemcpy(c->bins_totals_host, c->dep_total, th::kTotWords * sizeof(uint32_t), hipMemcpyDeviceToHost));
                return TH_OK;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return fail(TH_ERR_HIP, "the binned pass's totals did not arrive");
        }
    }
    return TH_OK;
}

th_status bins_store_for(th_context *c, th::DepositParams &p, uint32_t at_least)
{
    // (TH_OPT_BINS_POOL: the first pool's size in pages - tests make it small to run the growth path)
    const uint32_t pool0 = c->opt.bins_pool;
    uint32_t pool = c->bins_pool ? c->bins_pool : (pool0 ? pool0 : (p.nbins * 4u > 4096u ? p.nbins * 4u : 4096u));
    if (pool < at_least) pool = at_least;
    if (th_status s = bins_store(c, p.nbins, pool, p.mode == 2)) return s;
    p.frag_keys = c->bins_keys; p.colors = c->bins_colors; p.pool_pages = c->bins_pool;
    return TH_OK;
}

// every list's page table four times as wide (`keep`: with its entries - row-band shards: the owner's own bins are laid out
// already; otherwise the table is empty, as every pass leaves it)
th_status bins_table_widen(th_context *c, th::DepositParams &p, bool keep)
{
    const uint32_t had = c->bin_max_pages, wide = had * 4u < th::kBinPagesLimit ? had * 4u : th::kBinPagesLimit;
    if (c->opt.bins_pages < 0 || wide <= had) return kRetryInStreamOrder;
    const size_t lists = (size_t)c->bin_capacity * th::kBinReplicas;
    uint32_t *table = nullptr;
    if (hipMalloc((void **)&table, lists * wide * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return kRetryInStreamOrder; }
    TH_HIP(hipMemsetAsync(table, 0, lists * wide * sizeof(uint32_t), c->stream));
    if (keep) TH_HIP(hipMemcpy2DAsync(table, (size_t)wide * sizeof(uint32_t), c->chunk_table, (size_t)had * sizeof(uint32_t), (size_t)had * sizeof(uint32_t), lists, hipMemcpyDeviceToDevice, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->chunk_table);
    c->chunk_table = table; c->bin_max_pages = wide;
    p.page_table = table; p.max_pages = wide;
    return TH_OK;
}

// a larger pool behind the same bins WITH what the store holds (row-band shards: the owner's own bins wait in it for the
// other ranks' fragments); page ids stay what they are
th_status bins_store_grow_keep(th_context *c, th::DepositParams &p, uint32_t pool)
{
    if (pool <= c->bins_pool) return TH_OK;
    const size_t had = ((size_t)c->bins_store_bins * th::kBinReplicas + c->bins_pool) * th::kBinPage;
    const size_t places = ((size_t)c->bins_store_bins * th::kBinReplicas + pool) * th::kBinPage, per = c->bins_pairs ? 2 : 1;
    TH_REQUIRE(places < ((size_t)1 << 32), "the binned draw's chunk store would hold 2^32 places or more");
    unsigned long long *keys = nullptr;
    float4 *colors = nullptr;
    TH_HIP(hipMalloc((void **)&keys, places * sizeof(unsigned long long)));
    if (hipMalloc((void **)&colors, places * per * sizeof(float4)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(keys); return fail(TH_ERR_HIP, "the binned store could not grow to %zu places", places); }
    TH_HIP(hipMemcpyAsync(keys, c->bins_keys, had * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    TH_HIP(hipMemcpyAsync(colors, c->bins_colors, had * per * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->bins_keys); (void)hipFree(c->bins_colors);
    c->bins_keys = keys; c->bins_colors = colors; c->bins_pool = pool;
    p.frag_keys = keys; p.colors = colors; p.pool_pages = pool;
    return TH_OK;
}

// Part 1: rasterise + emit into the bins, the plan; repeated with a larger pool when the pool ran dry (nothing has been
// blended).  blend_early: the ordinary bins' blend goes out right behind the pass and covers the totals' read-back.
// Leaves the totals in c->bins_totals_host.  kRetryInStreamOrder: a bin outgrew its lists (or the store cannot be had).
th_status bins_pass_emit(th_context *c, th::DepositParams &p, bool blend_early)
{
    c->drawn.valid = false;
    if (th_status s = bins_streams(c)) return s;
    uint32_t *host = c->bins_totals_host;
    // A pass that emitted and was never blended (th_shard.hip: a peer gave up or failed after this rank's own emit had
    // succeeded) left keys in its places and page ids in its lists' tables: a page id found there before this pass publishes
    // its own would send fragments to pages another list owns now, and stale keys would blend as fragments nobody drew.
    if (c->bins_dirty) {
        if (c->bins_keys) TH_HIP(hipMemsetAsync(c->bins_keys, 0xff, ((size_t)c->bins_store_bins * th::kBinReplicas + c->bins_pool) * th::kBinPage * sizeof(unsigned long long), c->stream));
        if (c->chunk_table) TH_HIP(hipMemsetAsync(c->chunk_table, 0, (size_t)c->bin_capacity * th::kBinReplicas * c->bin_max_pages * sizeof(uint32_t), c->stream));
        c->bins_dirty = false;
    }
    for (int attempt = 0;; ++attempt) {
        if (th_status s = bins_store_for(c, p, 0)) return s;
        bins_expect(c, p);
        th::launch_bins_fused(p, c->stream);
        // the totals come back over the side stream while the ordinary bins are already being blended (the kernel looks at the
        // pass's flags itself): the host's round trip - it sizes the crowded bins' launches - costs the GPU nothing
        TH_HIP(hipEventRecord(c->forked, c->stream));
        if (blend_early) th::launch_bins_blend(p, c->stream);
        if (th_status s = bins_totals(c, p)) return s;
        const uint32_t flags = host[th::kTotFlags];
        if (host[th::kTotOob]) {          // (a store nobody will blend: wiped by the next pass)
            c->bins_dirty = true;
            return fail(TH_ERR_UNSUPPORTED, "a line of this row band looks up a particle row outside the band (rows %d..%d of %d) and no halo row was supplied (th_deposit_set_halo)", c->cfg.row0, c->cfg.row0 + c->cfg.height, c->cfg.global_height);
        }
        if (flags == 0) { c->bins_dirty = true; return TH_OK; }        // (until bins_pass_finish has sent the blends after it)
        // nothing has been blended yet: the store is wiped, and the pass is repeated with a larger pool - or, when a bin
        // outgrew its chunk table or a line its reservation, left to the stream-ordered pipeline
        TH_HIP(hipMemsetAsync(c->bins_keys, 0xff, ((size_t)c->bins_store_bins * th::kBinReplicas + c->bins_pool) * th::kBinPage * sizeof(unsigned long long), c->stream));
        TH_HIP(hipMemsetAsync(c->chunk_table, 0, (size_t)c->bin_capacity * th::kBinReplicas * c->bin_max_pages * sizeof(uint32_t), c->stream));
        if ((flags & ~(th::kBinsPoolExhausted | th::kBinsBinFull)) || attempt >= 6) return kRetryInStreamOrder;
        // (a bin that outgrew its lists - the wake of a long-running loop draws half a million fragments into 16 x 16 texels -
        // gets a wider page table: a list's pages x 4, up to kBinPagesLimit)
        if (flags & th::kBinsBinFull) { if (th_status s = bins_table_widen(c, p, false)) return s; }
        if (flags & th::kBinsPoolExhausted) {
            const uint32_t want = 2u * host[th::kTotPool] + 64;      // (generously: growing the store costs a frame's worth of time)
            if (th_status s = bins_store(c, p.nbins, want, p.mode == 2)) return s;
        }
    }
}

// Part 2: with the totals in c->bins_totals_host - the crowded bins regrouped and blended on the side streams, the ordinary
// bins' blend unless it went out early.
th_status bins_pass_finish(th_context *c, th::DepositParams &p, uint64_t *fragments, bool blended_early)
{
    uint32_t *host = c->bins_totals_host;
    const uint32_t total = host[th::kTotFragments], nlarge = host[th::kTotLarge];
    if (fragments) *fragments = total;
    c->last_draw.pipeline = TH_DRAW_BINS; c->last_draw.fragments = total; c->last_draw.crowded_fragments = host[th::kTotCrowdKeys];
    if (host[th::kTotCrowdKeys] == 0xffffffffu) return fail(TH_ERR_UNSUPPORTED, "too many fragments in crowded bins for one draw (2^32 or more places)");
    if (c->crowd_capacity < nlarge) {
        (void)hipFree(c->crowd_mem); c->crowd_mem = nullptr; c->crowd_capacity = 0;
        const uint32_t cap = std::min(p.nbins, std::max(2u * nlarge + 256u, p.nbins / 4u));      // (a quarter of the bins at once: no growth step by step)
        TH_HIP(hipMalloc((void **)&c->crowd_mem, (size_t)cap * th::crowd_words_per_bin() * sizeof(uint32_t)));
        // (the fragment counts per texel - the first cap * 256 words - start from zero; crowd_scan_kernel leaves them so)
        TH_HIP(hipMemsetAsync(c->crowd_mem, 0, (size_t)cap * 256 * sizeof(uint32_t), c->stream));
        TH_HIP(hipStreamSynchronize(c->stream));         // (whichever stream the regroup runs on: a new buffer is rare)
        c->crowd_capacity = cap;
    }
    if (c->crowd_keys_cap < host[th::kTotCrowdKeys]) {
        (void)hipFree(c->crowd_keys); (void)hipFree(c->crowd_sorted); (void)hipFree(c->crowd_parted); (void)hipFree(c->crowd_windows);
        c->crowd_keys = nullptr; c->crowd_sorted = nullptr; c->crowd_parted = nullptr; c->crowd_windows = nullptr; c->crowd_keys_cap = 0;
        // (room for every fragment of a pass like this one and a quarter more: the share of the crowded bins grows from a third
        // to three quarters over the first hundred frames of a loop - sized by what it is now, the four buffers were freed and
        // allocated again every few frames on the way, milliseconds each time)
        const size_t cap = std::max(2 * (size_t)host[th::kTotCrowdKeys], (size_t)total + total / 4) + ((size_t)1 << 20);
        TH_HIP(hipMalloc((void **)&c->crowd_keys, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->crowd_sorted, cap * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->crowd_parted, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->crowd_windows, (cap / 512 + 2) * 3 * sizeof(uint32_t)));      // (th_bins.hip: giant_part_kernel)
        c->crowd_keys_cap = cap;
    }
    p.nlarge = nlarge;
    p.crowd_count = c->crowd_mem; p.crowd_cursor = c->crowd_mem + (size_t)c->crowd_capacity * 256; p.crowd_start = p.crowd_cursor + (size_t)c->crowd_capacity * 256;
    p.crowd_long = p.crowd_start + (size_t)c->crowd_capacity * 257; p.crowd_giant = p.crowd_long + (size_t)c->crowd_capacity * 256;
    p.crowd_giant_win = p.crowd_giant + (size_t)c->crowd_capacity * 256;
    p.crowd_keys = c->crowd_keys; p.crowd_sorted = c->crowd_sorted; p.crowd_parted = c->crowd_parted;
    p.crowd_windows = c->crowd_windows; p.crowd_windows_cap = (uint32_t)(c->crowd_keys_cap / 512 + 2);
    if (nlarge && !blended_early) {
        // A crowded target: the crowded bins' short runs - regroup, sort, walk: the longest chain of the draw - stay on the MAIN
        // stream, right behind the plan (no event to wait for) and right in front of whatever the host sends next (a frame
        // loop's next step: behind a chain that ends on a side stream it waited 17-26 us for the join); beside them, on the
        // side streams, the ordinary bins' blend (disjoint texels, kernels that wait on chains and loads rather than fill the
        // chip) and the long runs - the walk of the longest run, one fragment after the other, overlaps with everything else.
        th::launch_bins_regroup(p, c->stream);
        th::launch_bins_part_giants(p, c->stream);          // (the long runs' stream is the longer one: its first kernel here)
        TH_HIP(hipEventRecord(c->regrouped, c->stream));
        TH_HIP(hipStreamWaitEvent(c->side2, c->forked, 0));       // (recorded behind the emitting pass and its plan)
        th::launch_bins_blend(p, c->side2);
        TH_HIP(hipEventRecord(c->joined2, c->side2));
        TH_HIP(hipStreamWaitEvent(c->side, c->regrouped, 0));
        th::launch_bins_blend_giants(p, c->side);
        th::launch_bins_sort_long(p, c->side);
        th::launch_bins_walk_long(p, c->side);
        TH_HIP(hipEventRecord(c->joined, c->side));
        th::launch_bins_blend_crowd(p, c->stream);
        // (both side streams waited for by the main stream: a wait costs it ~9 us even when the event has long been signalled, but
        // with the crowded bins' sort at 140 us the ordinary bins' blend ends LAST as often as not - the long runs' last kernel
        // waiting for it, so that the main stream had one event to wait for, stood idle in half of the frames)
        TH_HIP(hipStreamWaitEvent(c->stream, c->joined, 0)); TH_HIP(hipStreamWaitEvent(c->stream, c->joined2, 0));
    } else if (nlarge) {
        // The ordinary bins' blend went out early on the main stream (few fragments in crowded bins last time): the crowded
        // bins on two streams of their own beside it - their fragments regrouped by texel, then the long runs on one stream
        // and the short runs on the other.
        TH_HIP(hipStreamWaitEvent(c->side2, c->forked, 0));       // (recorded behind the emitting pass and its plan: the ordinary bins' blend need not be waited for)
        th::launch_bins_regroup(p, c->side2);
        th::launch_bins_part_giants(p, c->side2);
        TH_HIP(hipEventRecord(c->regrouped, c->side2));
        TH_HIP(hipStreamWaitEvent(c->side, c->regrouped, 0));
        th::launch_bins_blend_giants(p, c->side);
        // ... and the runs in between behind them.  (A stream of their own shares a hardware queue with one of the others -
        // four per process by default - and holds that one's kernels back; behind the ordinary bins' blend on the main stream,
        // or ordered behind the short runs and walked on the main stream: the same within 1-3 % one way or the other, by the
        // phase of the loop - the blend phase of a crowded draw is bound by the chip's throughput, not by one stream's chain:
        // profiles/r4_g_giants.txt)
        th::launch_bins_sort_long(p, c->side);
        th::launch_bins_walk_long(p, c->side);
        TH_HIP(hipEventRecord(c->joined, c->side));
        th::launch_bins_blend_crowd(p, c->side2);
        TH_HIP(hipEventRecord(c->joined2, c->side2));
        TH_HIP(hipStreamWaitEvent(c->stream, c->joined, 0)); TH_HIP(hipStreamWaitEvent(c->stream, c->joined2, 0));
    } else if (!blended_early) th::launch_bins_blend(p, c->stream);
    TH_HIP(hipGetLastError());
    c->bins_dirty = false;          // (every reader of a list leaves its places and its table entries empty)
    return TH_OK;
}

// (row-band shards, th_shard.hip) the received bins are in the store, laid out by launch_bins_owner_insert: the totals of its
// plan to the host
void bins_pass_expect(th_context *c, th::DepositParams &p) { bins_expect(c, p); }
th_status bins_pass_totals(th_context *c, const th::DepositParams &p)
{
    TH_HIP(hipEventRecord(c->forked, c->stream));
    return bins_totals(c, p);
}

}  // namespace thi

static th_status deposit_run_bins(th_context *c, th::DepositParams &p, uint64_t *fragments)
{
    if (th_status s = bins_streams(c)) return s;
    // Which comes first behind the emitting pass: the ordinary bins' blend - it needs nothing from the host and covers the
    // read-back - or, on a crowded target, the crowded bins' kernels: their long runs (walked by one thread each, on the
    // side stream) are then the longest chain of the draw and must start as early as they can.  Decided by the last draw.
    const bool early = !(c->last_draw.pipeline == TH_DRAW_BINS && (double)c->last_draw.crowded_fragments > kEarlyBlendShare * (double)c->last_draw.fragments);
    if (th_status s = bins_pass_emit(c, p, early)) return s;
    return bins_pass_finish(c, p, fragments, early);
}

extern "C" {

th_status th_export_lines(th_context *c, const th_deposit_uniforms *u, float *lines, uint64_t capacity, uint64_t *count)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(count, "null count");
    th::DepositParams p;
    if (th_status s = deposit_prepare(c, u, p)) return s;
    return export_run(c, p, lines, capacity, count);
}

th_status th_flow_deposit(th_context *c, const th_deposit_uniforms *u, uint64_t *fragments)
{
    if (th_status s = use(c)) return s;
    c->last_draw.sent_bytes = c->last_draw.received_bytes = 0;      // (a local draw moves nothing between ranks)
    if (c->cfg.height != c->cfg.global_height)
        return fail(TH_ERR_UNSUPPORTED, "flow deposit on a row-band shard (%d of %d rows): use th_deposit_emit / th_deposit_merge with the exchange of tendrils_amd/sharding.py", c->cfg.height, c->cfg.global_height);
    for (int pass = 0;; ++pass) {            // (a binned pass that gives up before blending is repeated in stream order)
        th::DepositParams p;
        bool bins = false;
        if (th_status s = deposit_prepare(c, u, p, pass == 0, &bins)) return s;
        if (!bins) return deposit_run(c, p, fragments);
        const th_status s = deposit_run_bins(c, p, fragments);
        if (s != kRetryInStreamOrder) return s;
        c->frame_bins = 0;                  // (the other passes of this frame as well)
    }
}

// Both passes of Tendrils.draw() (src/index.js:278-337) in one: the lines are rasterised, scanned, emitted and sorted once,
// every fragment carries the flow pass's varying and the view pass's colour side by side, one gather brings both into
// the sorted order and each target is blended from its half.  The two passes must agree on what they draw: the same
// viewSize, time and speedLimit (what Tendrils.draw() hands to both); results are those of th_flow_deposit followed
// by th_view_draw.
th_status th_draw(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *fragments)
{
    if (th_status s = use(c)) return s;
    c->last_draw.sent_bytes = c->last_draw.received_bytes = 0;      // (a local draw moves nothing between ranks)
    TH_REQUIRE(du && ru, "null uniforms");
    if (c->cfg.height != c->cfg.global_height)
        return fail(TH_ERR_UNSUPPORTED, "draw on a row-band shard (%d of %d rows): the passes go through th_deposit_emit / th_deposit_merge and th_view_emit / th_view_merge with the owners' exchange in between", c->cfg.height, c->cfg.global_height);
    TH_REQUIRE(memcmp(du->viewSize, ru->viewSize, sizeof du->viewSize) == 0 && memcmp(&du->time, &ru->time, sizeof du->time) == 0 &&
               memcmp(&du->speedLimit, &ru->speedLimit, sizeof du->speedLimit) == 0,
               "the two passes of one draw share viewSize, time and speedLimit");
    if (drawn_line_width(c, TH_PASS_FLOW) != drawn_line_width(c, TH_PASS_VIEW)) {       // two widths: two rasterisations
        if (th_status s = th_flow_deposit(c, du, fragments)) return s;
        return th_view_draw(c, ru, nullptr);
    }
    if (th_status s = view_storage(c)) return s;
    for (int pass = 0;; ++pass) {
        th::DepositParams p;
        bool bins = false;
        if (th_status s = deposit_prepare(c, du, p, pass == 0, &bins)) return s;
        p.mode = 2;
        view_fields(c, ru, p);
        p.view = c->view;
        if (!bins) return deposit_run(c, p, fragments);
        const th_status s = deposit_run_bins(c, p, fragments);
        if (s != kRetryInStreamOrder) return s;
        c->frame_bins = 0;                  // (the other passes of this frame as well)
    }
}

th_status th_view_draw(th_context *c, const th_render_uniforms *u, uint64_t *fragments)
{
    if (th_status s = use(c, true)) return s;
    c->last_draw.sent_bytes = c->last_draw.received_bytes = 0;      // (a local draw moves nothing between ranks)
    if (c->cfg.height != c->cfg.global_height)
        return fail(TH_ERR_UNSUPPORTED, "view pass on a row-band shard (%d of %d rows): use th_view_emit / th_view_merge with the owners' exchange in between", c->cfg.height, c->cfg.global_height);
    if (th_status s = view_storage(c)) return s;
    for (int pass = 0;; ++pass) {
        th::DepositParams p;
        bool bins = false;
        if (th_status s = view_params(c, u, p, pass == 0, &bins)) return s;
        p.view = c->view;
        if (!bins) return deposit_run(c, p, fragments);
        const th_status s = deposit_run_bins(c, p, fragments);
        if (s != kRetryInStreamOrder) return s;
        c->frame_bins = 0;                  // (the other passes of this frame as well)
    }
}

th_status th_view_fill(th_context *c, const float rgba[4])
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(rgba, "null colour");
    if (th_status s = view_storage(c)) return s;
    th::launch_view_fill(c->view, (size_t)c->view_w * c->view_h, make_float4(rgba[0], rgba[1], rgba[2], rgba[3]), c->stream);
    TH_HIP(hipGetLastError());
    return TH_OK;
}

th_status th_view_clear(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    TH_HIP(hipMemsetAsync(c->view, 0, (size_t)c->view_w * c->view_h * sizeof(uchar4), c->stream));
    return TH_OK;
}

// Tendrils.buffers (src/index.js:172-184, 359-391)
th_status th_view_buffers(th_context *c, int32_t count)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(count >= 0 && count <= 64, "bad number of view buffers %d", count);
    c->view_buffers = count;
    return view_storage(c);
}

th_status th_view_bind(th_context *c, int32_t index)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    TH_REQUIRE(index >= -1 && index < (int32_t)c->view_ring.size(), "no view buffer %d (there are %zu)", index, c->view_ring.size());
    c->view = index < 0 ? c->view_screen : c->view_ring[(size_t)index];
    c->view_bound = index;
    return TH_OK;
}

th_status th_view_copy(th_context *c, int32_t index)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    if (index < 0 || index >= (int32_t)c->view_ring.size()) return TH_OK;       // src/index.js:371: `if(index < this.buffers.length)`
    th::launch_view_copy(c->view, c->view_ring[(size_t)index], (size_t)c->view_w * c->view_h, c->stream);
    TH_HIP(hipGetLastError());
    return TH_OK;
}

th_status th_view_step_buffers(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    if (c->view_ring.size() > 1) {               // src/utils/index.js:1-7: array.unshift(array.pop())
        uchar4 *last = c->view_ring.back();
        c->view_ring.pop_back();
        c->view_ring.insert(c->view_ring.begin(), last);
    }
    return TH_OK;
}

th_status th_view_download(th_context *c, uint8_t *rgba8)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(rgba8, "null pixels");
    if (th_status s = view_storage(c)) return s;
    TH_HIP(hipMemcpyAsync(rgba8, c->view, (size_t)c->view_w * c->view_h * sizeof(uchar4), hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_colormap_upload(th_context *c, const float *rgba, int32_t w, int32_t h)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(rgba && w > 0 && h > 0 && (uint64_t)w * h < (1ull << 28), "bad colour map %dx%d", w, h);
    if (w != c->cmap_w || h != c->cmap_h) {
        TH_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->colormap);
        c->colormap = nullptr; c->cmap_w = c->cmap_h = 0;
        TH_HIP(hipMalloc((void **)&c->colormap, (size_t)w * h * sizeof(float4)));
        c->cmap_w = w; c->cmap_h = h;
    }
    TH_HIP(hipMemcpyAsync(c->colormap, rgba, (size_t)w * h * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_export_view_lines(th_context *c, const th_render_uniforms *u, float *lines, uint64_t capacity, uint64_t *count)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(count, "null count");
    th::DepositParams p;
    if (th_status s = view_params(c, u, p)) return s;
    return export_run(c, p, lines, capacity, count);
}

th_status th_draw_query(th_context *c, th_draw_info *out)
{
    TH_REQUIRE(c && out, "null argument");
    *out = c->last_draw;
    return TH_OK;
}

th_status th_line_width(th_context *c, int32_t pass, float width)
{
    TH_REQUIRE(c, "null context");
    TH_REQUIRE(pass == TH_PASS_FLOW || pass == TH_PASS_VIEW, "unknown pass %d", pass);
    TH_REQUIRE(width > 0.0f, "line width %g (gl.lineWidth: INVALID_VALUE, the width stays %g)", (double)width, (double)c->line_width[pass]);
    c->line_width[pass] = width;
    return TH_OK;
}

th_status th_line_width_range(th_context *c, float lo, float hi)
{
    TH_REQUIRE(c, "null context");
    TH_REQUIRE(lo > 0.0f && lo <= 1.0f && hi >= 1.0f && hi <= th::kMaxLineWidth, "line width range [%g, %g]: need 0 < lo <= 1 <= hi <= %g",
               (double)lo, (double)hi, (double)th::kMaxLineWidth);
    c->line_range[0] = lo; c->line_range[1] = hi;
    return TH_OK;
}

th_status th_line_width_query(th_context *c, int32_t pass, float *width, float *drawn, float *range)
{
    TH_REQUIRE(c, "null context");
    TH_REQUIRE(pass == TH_PASS_FLOW || pass == TH_PASS_VIEW, "unknown pass %d", pass);
    if (width) *width = c->line_width[pass];
    if (drawn) *drawn = drawn_line_width(c, pass);
    if (range) { range[0] = c->line_range[0]; range[1] = c->line_range[1]; }
    return TH_OK;
}

th_status th_draw_pipeline(th_context *c, int32_t which)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(which == TH_DRAW_AUTO || which == TH_DRAW_STREAM || which == TH_DRAW_BINS, "unknown draw pipeline %d", which);
    c->draw_pipeline = which;
    return TH_OK;
}

}  // extern "C"
