// th_ctx.hpp - what the translation units behind include/tendrils_hip.h share: the context, the error helpers and the
// internal functions one unit offers the others (namespace thi).  Host side only; the kernels' interface is th_kernels.hpp.
//   th_api.hip    context life cycle, textures, read-backs, timers, options
//   th_order.hip  tile-sorted slot orders of the ring buffers, captured th_step_n graphs
//   th_step.hip   Particles.step: th_step / th_step_n
//   th_spawn.hip  the spawners
//   th_draw.hip   Tendrils.draw(): flow pass, view pass, trail export (binned and stream-ordered pipeline)
//   th_shard.hip  row-band shards: emit / merge, th_draw_sharded, the job's communicator, gathers, counter all-reduce
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "th_kernels.hpp"
#include "th_math.hpp"

namespace thi {
th_status fail(th_status code, const char *fmt, ...);      // records the message th_last_error() returns; returns `code`
std::string &last_error();
}  // namespace thi

#define TH_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return thi::fail(TH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define TH_REQUIRE(cond, ...)                            \
    do {                                                 \
        if (!(cond)) return thi::fail(TH_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// Per-context switches (th_option_set / th_option_get).  A context starts from the environment variables of the same
// names, read when it is created - not once per process - so one process can hold contexts on different paths.
struct th_options {
    int bucket = -1;                     // TH_BUCKET: tile-sorted slot order never (0) / always (1) / when it pays (-1)
    int resort_steps = 64;               // TH_RESORT_STEPS: re-sort period of single-step launches
    int rebucket_steps = 256;            // TH_REBUCKET_STEPS: ... of fused launches
    bool fuse = true;                    // TH_FUSE: temporal fusion in th_step_n
    bool graph = true;                   // TH_GRAPH: captured graphs in th_step_n
    bool force_generic = false;          // TH_FORCE_GENERIC: every step through the reference-order kernel
    int draw = -1;                       // TH_DRAW=stream (0) / bins (1): the default of th_draw_pipeline's AUTO
    bool draw_reuse = true;              // TH_DRAW_REUSE: the stream-ordered view pass reuses the flow pass's geometry
    uint32_t bins_pool = 0;              // TH_BINS_POOL: first size of the binned pipeline's page pool (0: by the target's size)
    int bins_pages = 0;                  // TH_BINS_PAGES: pages a bin's list can grow to at first (0: kBinFirstPages); negative: that many and never more
    int inject_failure = 0;              // (tests) the next th_draw_sharded fails on THIS rank at stage 1 / 2 / 3: the ranks must all leave
    bool skip_unseen = true;             // TH_SKIP_UNSEEN: draw() skips the blocks of slots whose lines the step saw end up outside the view (th_step.hip)
    bool async_sort = true;              // TH_ASYNC_SORT: a frame loop's re-sort runs beside its draw() instead of inside two of its steps (th_step.hip)
};

// One captured th_step_n sequence (see th_step_n).
struct GraphEntry {
    int32_t n = 0, mode = 0;
    uint32_t flags = 0;
    std::vector<float4 *> ring;          // ring order at capture time
    th::LogicParams key{};               // launch parameters (the fields same_key() compares)
    hipGraphExec_t exec = nullptr;
    float *times_dev = nullptr, *times_host = nullptr;
    hipEvent_t copied = nullptr;         // times_host -> times_dev copy of the last replay
};

struct th_context {
    th_config cfg{};
    th_options opt{};
    hipStream_t stream = nullptr;
    std::vector<float4 *> ring;          // ring[0] = buffers[0] (most recent); TH_STATE_F16: packed, 8 B per texel
    bool packed = false;                 // cfg.state_format == TH_STATE_F16
    float4 *tmp[3] = {nullptr, nullptr, nullptr};   // f32 staging for the non-hot operations on a packed ring
    float4 *flow = nullptr;
    float2 *flow_dec = nullptr;          // per-step decoded plane (launch_flow_decode)
    float *flow3 = nullptr;              // the flow texels' x, y, z alone (fused passes: th_step_n packs them once per call)
    int32_t fw = 0, fh = 0;
    float4 *targets = nullptr;
    bool targets_checked = true, targets_nonfinite = false;   // fresh texture = zeros
    float4 *lut = nullptr, *lut_block = nullptr;      // gradient table (inside lut_block, behind the hash tables)
    uchar4 *frames[2] = {nullptr, nullptr};
    int32_t frw = 0, frh = 0;
    unsigned int *d_flag = nullptr;
    th::StatsPartial *partials = nullptr;
    // the statistics a fused th_step_n launch took of the state it wrote (LogicParams::stats_part): valid while ring[0] is
    // that buffer and nothing has written it (use() without keeps_lines drops them)
    th::StatsPartial *fused_parts = nullptr;
    uint32_t fused_parts_cap = 0;
    struct { bool valid = false; const float4 *buf = nullptr; float limit = 0.0f; uint32_t nparts = 0; } fused_stats;
    th_counters *d_counters = nullptr;
    // th_draw_sharded: the neighbours' edge rows, the owners' counts, what this rank received
    float4 *x_halo = nullptr;            // [lo: cur row, prev row | hi: cur row, prev row], `width` texels each
    unsigned long long *x_counts = nullptr;   // device: bounds (33) | send counts (32) | recv counts (32)
    unsigned long long *x_keys = nullptr;
    float4 *x_colors = nullptr;
    size_t x_capacity = 0;
    float4 *gathered = nullptr;          // row-band shard: a copy of the WHOLE particle texture (th_state_gather / _ptr) ...
    const void *gathered_of = nullptr;   // ... of this ring buffer, for the spawners that sample arbitrary particles
    void *comm = nullptr;                // communicator of the job's ranks (th_comm_init), one rank per context ...
    const th::Transport *transport = nullptr;   // ... and how its ranks exchange bytes (RCCL; in-process for tests)
    uint32_t *d_status = nullptr;        // the word the ranks agree on (agree_status)
    void *own_mem = nullptr;             // th_draw_sharded through the bins: counts, offsets, tables (th_bins.hip: OwnerParams)
    uint32_t own_bins = 0;
    bool sharded_draw_ready = false;     // th_draw_sharded has allocated its fixed buffers (and the ranks agreed that all did)
    int32_t comm_rank = 0, comm_world = 1;
    // flow deposit scratch (grow-only): per-flow-texel counters and the fragment lists
    uint32_t *dep_count = nullptr, *dep_offset = nullptr, *dep_blocks = nullptr, *dep_total = nullptr;   // per line; scan scratch
    uint4 *dep_record = nullptr;         // per line: the texels of a short line
    uint32_t *dep_lists = nullptr;       // slow / long line lists (counters first)
    uint32_t dep_owners = 1;             // th_deposit_set_owners: ranks owning flow texels in the sharded deposit
    bool dep_pairs = false;              // the colour buffers hold two varyings per fragment (th_draw)
    // the geometry of the last draw pass (fragment counts, offsets, records, the sorted fragment order): the flow pass
    // and the view pass of one draw() rasterise the same lines at the same resolution
    float line_width[2] = {1.0f, 1.0f}, line_range[2] = {1.0f, 1.0f};     // th_line_width (per pass: TH_PASS_FLOW, TH_PASS_VIEW) / th_line_width_range
    struct { bool valid = false, binned = false; float view_x = 0, view_y = 0, line_half = 0; uint32_t total = 0, nlarge = 0, nblocks = 0; bool sorted_in_a = false; } drawn;
    uint32_t dep_list_cap = 0;
    // binned pipeline (th_bins.hip): the bins' cursors | the large bins | first block of each (+ 1) | first regrouped key of each (+ 1)
    uint32_t *bin_mem = nullptr;
    uint32_t bin_capacity = 0;
    uint32_t *chunk_table = nullptr;     // per list x bin_max_pages: the pages a list has grown by
    // the blocks of 256 slots with a line that can draw, for the slot order ring[0] is held in (th_bins.hip: bins_block_list_kernel)
    uint32_t *draw_blocks = nullptr, draw_nblocks = 0;
    uint8_t *draw_block_flags = nullptr;
    int draw_blocks_order = -2;
    unsigned long long draw_blocks_stamp = 0;
    uint32_t bin_max_pages = 0;          // (widened when a bin outgrows its lists: bins_table_widen)
    bool bins_dirty = false;             // an emitting pass filled the store and no blend has emptied it since (a sharded draw that ended
                                         // between the two): the next emitting pass wipes it first
    unsigned long long *bins_keys = nullptr;   // the page store: (bins x kBinReplicas + bins_pool) pages of kBinPage places - keys (~0 = empty) ...
    float4 *bins_colors = nullptr;       // ... and varyings (two per place once a th_draw has run)
    uint32_t bins_pool = 0, bins_store_bins = 0;
    bool bins_pairs = false;
    uint32_t *crowd_mem = nullptr;       // per large bin: fragments per texel, first fragment of every texel, fill cursors, long runs
    uint32_t crowd_capacity = 0;
    unsigned long long *crowd_keys = nullptr;  // the large bins' fragments regrouped by texel
    uint32_t *crowd_sorted = nullptr;          // ... their places, run by run in blend order
    unsigned long long *crowd_parted = nullptr;  // ... the giants' keys parted by stream index, and their windows (th_bins.hip: giant_*_kernel)
    uint32_t *crowd_windows = nullptr;
    size_t crowd_keys_cap = 0;
    hipStream_t side = nullptr;                // the long runs of a crowded target are blended beside everything else
    hipEvent_t forked = nullptr, joined = nullptr;
    hipStream_t side2 = nullptr;               // ... and the crowded bins' short runs beside both
    hipEvent_t joined2 = nullptr, regrouped = nullptr;
    uint32_t *bins_totals_host = nullptr;      // (pinned, coherent; kTotWords + 1 words) the binned pass's totals, written by the plan's last kernel, and the sequence number behind them
    uint32_t *bins_totals_dev = nullptr;       // ... as the device addresses it
    uint32_t totals_seq = 0;
    bool mrg_pairs = false, x_pairs = false;   // the merge / exchange colour buffers hold two varyings per fragment (th_draw_emit / _merge)
    void *pinned = nullptr;                    // (pinned, kPinnedBytes) small read-backs: a pageable hipMemcpyAsync costs ~0.15 ms per call
    int lines_local = -1;                // every vertex of every line reads the line's own particle (line_rows)
    uint32_t *d_row_draws = nullptr;     // bit per global row: the row's lines can draw (line_rows)
    // ... and when not (lines_local == 0): the rows / columns whose texels some OTHER line's vertex reads, and - per slot order -
    // where those texels lie (th::LineSources; th_bins.hip: bins_block_flags_kernel)
    uint16_t *src_row_index = nullptr, *src_col_index = nullptr;
    uint32_t src_nrows = 0, src_ncols = 0;
    bool rows_cross_bands = false;       // some line of this band looks a row of a neighbouring band up (halo rows needed)
    uint32_t *src_slots = nullptr;       // (allocated with draw_blocks; valid for draw_blocks_order / _stamp)
    float4 *edge_rows = nullptr;         // th_draw_sharded through the bins: this band's edge rows gathered into texel order (4 x W)
    int draw_pipeline = TH_DRAW_AUTO;    // th_draw_pipeline
    th_draw_info last_draw{};            // th_draw_query
    long long draws = 0;                          // (`draws` counts frames: the passes drawn at one total_steps share a count ...
    long long draw_frame_step = -1;               //  ... and a pipeline: th_draw.hip, draw_uses_bins)
    int frame_bins = -1;
    long long last_binned_draw = -(1ll << 40);   // total_steps at the last draw over slot order
    uint32_t *dep_u32[4] = {nullptr, nullptr, nullptr, nullptr};     // per fragment: keys, slots, and both sorted
    unsigned long long *dep_u64[2] = {nullptr, nullptr};             // sharded form: (texel, stream index) keys, sorted
    float4 *dep_colors_sorted = nullptr;
    bool dep_wide = false;
    const float4 *halo_lo = nullptr, *halo_hi = nullptr;             // caller-owned neighbour rows (th_deposit_set_halo)
    unsigned long long *mrg_keys = nullptr, *mrg_keys2 = nullptr;    // th_deposit_merge scratch (sort ping-pong)
    uint32_t *mrg_vals[2] = {nullptr, nullptr};
    size_t mrg_capacity = 0;
    float4 *mrg_colors = nullptr;        // the received varyings gathered into texel order
    float4 *dep_colors = nullptr;
    void *dep_temp = nullptr;
    size_t dep_lines = 0, dep_capacity = 0, dep_temp_bytes = 0;
    uchar4 *view = nullptr;              // the BOUND view image (RGBA8, flow shape): what the view pass, fills, clears and read-backs touch
    int32_t view_w = 0, view_h = 0;
    uchar4 *view_screen = nullptr;       // the drawing buffer (bound unless th_view_bind chose a buffer), lazily allocated
    std::vector<uchar4 *> view_ring;     // Tendrils.buffers (src/index.js:172-184): off-screen view images, in ring order
    int32_t view_buffers = 0;            // how many the host asked for (th_view_buffers)
    int32_t view_bound = -1;             // ring position of the bound image when it was bound; -1: the screen (the pointer `view` is what counts)
    float4 *colormap = nullptr;          // tendrils.colorMap (nullptr = the 1x1 zero texture)
    int32_t cmap_w = 0, cmap_h = 0;
    float4 *image = nullptr;             // PixelSpawner's own buffer (TH_SOURCE_IMAGE)
    int32_t iw = 0, ih = 0;
    unsigned long long *d_respawned = nullptr;   // [0]: particles replaced by respawn passes, [1]: scratch (passes into `targets`)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool kernel_timing = false;          // th_kernel_timing: event pair around every logic launch
    std::vector<hipEvent_t> kt_events;   // pairs (start, stop); kt_used of them recorded
    size_t kt_used = 0;
    std::vector<GraphEntry> graphs;      // th_step_n cache

    // Tile-sorted slot orders (th_kernels.hip "Tile-sorted slot order"); lazily allocated.  Every ring buffer is in
    // texel order or in one of `orders` (a step that re-sorts writes its output in a new order while its input keeps
    // the old one, so two orders can be alive at a time).
    struct SlotOrder {
        uint32_t *perm = nullptr;            // slot -> particle id
        th::TileChunk *chunks = nullptr;     // chunk table
        th::ChunkRecord *records = nullptr;  // per chunk: tiles of the next positions (written by a COUNT pass)
        uint32_t *nchunks = nullptr;
        th::TileGeom geom{};                 // key function the order was sorted with
        int32_t fw = 0, fh = 0;
        int refs = 0;                        // ring buffers stored in this order
        unsigned long long stamp = 0;        // c->sorts when the order was laid out (what is cached per order - the draw's block list - knows it by this)
    };
    std::vector<SlotOrder> orders;
    std::vector<std::pair<float4 *, int>> buf_order;   // ring buffers held in a sorted order (absent = texel order)
    float4 *spare = nullptr;             // spare state buffer (ensure_identity moves through it)
    uint32_t *tile_mem = nullptr;        // hist | cursor (kSortReplicas x kMaxTileBins words each) | misses (8 words) | totals | starts (kMaxTileBins each)
    th::ChunkRecord *block_records = nullptr;   // per 4096-slot block: tile_hist's table for tile_scatter
    uint32_t max_chunks = 0;
    int steps_since_sort = 0;
    unsigned long long sorts = 0;
    long long total_steps = 0, hold_texel_order_until = 0;   // texel-order consumers (draw) keep the layout off for a period
    uint32_t *miss_host = nullptr;       // pinned: window misses since the last sort, as of some recent launch
    // A re-sort under way beside a draw() (th_step.hip "the re-sort of a frame loop"): the step's output `src` (held in order
    // `src_order`) is being copied into `dst` in the new order `order` on the side stream; the next step takes the copy for
    // its input when nothing has touched `src` since (state_written / state_moved), anything else drops it.
    struct {
        bool pending = false, valid = false;
        const float4 *src = nullptr;
        int src_order = -1, order = -1;
        float4 *dst = nullptr;               // (allocated once)
        hipEvent_t ready = nullptr, done = nullptr;
        long long at_step = -1;              // total_steps when it was started
    } asort;
    // what a single step saw of its lines (LogicParams::seen): valid for a draw() that reads exactly these two buffers in this
    // order through this view
    struct {
        uint8_t *bytes = nullptr;            // texels / 64 of them (a multiple of 4)
        const float4 *cur = nullptr, *prev = nullptr;
        int order = -1;
        unsigned long long stamp = 0;
        float view_x = 0, view_y = 0;
        int32_t fw = 0, fh = 0;
    } seen;
    // a COUNT pass has histogrammed the tiles of the state it wrote: valid for a SCATTER pass that reads exactly that
    struct { const float4 *buf = nullptr; int order = -1; th::TileGeom geom{}; long long at_step = -1; } counted;

    size_t texels() const { return (size_t)cfg.width * cfg.height; }
    size_t state_bytes() const { return texels() * (packed ? sizeof(uint2) : sizeof(float4)); }
};

namespace thi {

// ---- th_api.hip ------------------------------------------------------------------------------------------------------
inline bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }
inline uint32_t ilog2(uint32_t v) { uint32_t r = 0; while (v >>= 1) ++r; return r; }
th_status use(th_context *c, bool keeps_lines = false);
th_status alloc_state(th_context *c, float4 **out);
th_status resolve_target(th_context *c, int32_t target, bool rotate_ok, float4 **out);
th_status rect_ok(th_context *c, int32_t x0, int32_t y0, int32_t w, int32_t h);
th_status staging(th_context *c, int k, float4 **out);
th_status unpacked_view(th_context *c, float4 *buf, int k, float4 **out);
th_status render_target(th_context *c, float4 *buf, int k, float4 **out);
th_status commit_target(th_context *c, float4 *buf, float4 *rendered);
constexpr size_t kPinnedBytes = 1024;
th_status read_back(th_context *c, void *host, const void *dev, size_t bytes);
// the gathered whole-texture copy (th_state_gather / _ptr) is a copy of one ring buffer's CONTENT: writing that buffer ends
// its validity, moving the content to another allocation (slot-order moves through `spare`) takes the association along
inline void state_written(th_context *c, const float4 *buf)
{
    if (c->gathered_of == (const void *)buf) c->gathered_of = nullptr;
    if (c->asort.src == buf) c->asort.valid = false;          // (a re-sort of that content under way: its copy is stale)
    if (c->seen.cur == buf || c->seen.prev == buf) c->seen.cur = c->seen.prev = nullptr;
}
inline void state_moved(th_context *c, const float4 *from, const float4 *to)
{
    if (c->gathered_of == (const void *)from) c->gathered_of = to;
    if (c->asort.src == from) c->asort.valid = false;
    if (c->seen.cur == from || c->seen.prev == from) c->seen.cur = c->seen.prev = nullptr;
}

// ---- th_order.hip ----------------------------------------------------------------------------------------------------
void destroy_graph(GraphEntry &g);
void clear_graphs(th_context *c);
constexpr int kTileShift = 5;            // 32 x 32 texel tiles (th_kernels.hip kTile)
constexpr size_t kTileWords = 2 * (size_t)th::kSortReplicas * th::kMaxTileBins;   // histogram + cursors, all copies
uint32_t tile_count(const th_context *c, uint32_t *tiles_x);
bool sorting_possible(const th_context *c);
th_status line_rows(th_context *c);
th::TileGeom tile_geom(const th_context *c, const th_logic_uniforms &u);
bool same_geom(const th::TileGeom &a, const th::TileGeom &b);
int order_of(const th_context *c, const float4 *buf);
void set_order(th_context *c, float4 *buf, int order);
bool any_sorted(const th_context *c);
th_status sort_storage(th_context *c);
th_status free_order(th_context *c, int *out);
th_status asort_drop(th_context *c);
th_status asort_start(th_context *c, const th::TileGeom &g, float4 *src, int src_order);
th_status ensure_identity(th_context *c, bool *launched = nullptr);
th_status begin_sort(th_context *c, const th::TileGeom &g, const float4 *state, const uint32_t *perm_in, int *order,
                     th::TileSortParams *params, bool have_hist = false);
th_status align_slot_orders(th_context *c);

// ---- th_draw.hip -----------------------------------------------------------------------------------------------------
int deposit_texel_bits(const th_context *c);
float drawn_line_width(const th_context *c, int pass);
th_status deposit_prepare(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, bool want_bins = false, bool *bins = nullptr);
th_status deposit_scan_total(th_context *c, const th::DepositParams &p, uint32_t *total);
th_status deposit_count(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, uint32_t *total);
th_status deposit_reserve(th_context *c, uint32_t total, bool wide, bool pairs = false);
th_status deposit_temp(th_context *c, size_t need);
th_status view_storage(th_context *c);
void view_fields(th_context *c, const th_render_uniforms *u, th::DepositParams &p);
th_status view_params(th_context *c, const th_render_uniforms *u, th::DepositParams &p, bool want_bins = false, bool *bins = nullptr);
// the binned pipeline in parts (deposit_run_bins = emit + finish; row-band shards put the owners' exchange in between)
constexpr th_status kRetryInStreamOrder = -1;        // (internal) the binned pass gave up before it touched a target
th_status bins_store_for(th_context *c, th::DepositParams &p, uint32_t at_least);
th_status bins_store_grow_keep(th_context *c, th::DepositParams &p, uint32_t pool);
th_status bins_table_widen(th_context *c, th::DepositParams &p, bool keep);       // kRetryInStreamOrder: as wide as it goes (or no memory)
th_status bins_pass_emit(th_context *c, th::DepositParams &p, bool blend_early);
void bins_pass_expect(th_context *c, th::DepositParams &p);          // before the plan's kernels are launched with p ...
th_status bins_pass_totals(th_context *c, const th::DepositParams &p);  // ... their totals in c->bins_totals_host
th_status bins_pass_finish(th_context *c, th::DepositParams &p, uint64_t *fragments, bool blended_early);
bool binned_shards(const th_context *c);              // a sharded draw() of this job goes through the bins (the same answer on every rank)
th_status deposit_prepare_bins(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p);

}  // namespace thi
