// th_api.hip - the C ABI declared in include/tendrils_hip.h: context, state ring,
// flow/targets/frame textures, launch selection.  Host side only; kernels live in
// th_kernels.hip.  Compiled with -ffp-contract=off (the noise gradient table is
// built here in strict fp32, see build_gradient_table).
//
// Reference objects replaced (paths relative to the reference tree):
//   Particles            src/particles.js:43-196   (ring of N x N RGBA32F FBOs, step, spawn upload)
//   utils.step           src/utils/index.js:1-7    (ring rotation: pop -> unshift)
//   Tendrils.flow/targets src/index.js:102-105,207,231-236,405
//   OpticalFlow buffers  src/optical-flow/index.js:43-70
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

namespace {

thread_local std::string g_error;

th_status fail(th_status code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define TH_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(TH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define TH_REQUIRE(cond, ...)                            \
    do {                                                 \
        if (!(cond)) return fail(TH_ERR_INVALID, __VA_ARGS__); \
    } while (0)

bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }
uint32_t ilog2(uint32_t v) { uint32_t r = 0; while (v >>= 1) ++r; return r; }

// Table of the normalised simplex-noise gradient as a function of the argument
// of the LAST permute() (th_math.hpp kLutMin..kLutMax): folds one permute, the
// octahedron decode and the taylorInvSqrt scale of glsl-noise simplex/3d into a
// 16-byte LDS lookup.  Strict fp32, reference operation order.
void build_gradient_table(float4 *lut)
{
    for (int a = th::kLutMin; a <= th::kLutMax; ++a) {
        float p = th::permute_ref((float)a);
        float gx, gy, gz;
        th::gradient_ref(p, gx, gy, gz);
        lut[a - th::kLutMin] = make_float4(gx, gy, gz, 0.0f);
    }
}

// Largest s2 with sqrt_rn(s2) <= limit (sqrt_rn monotonic), so that
// `0 < s2 <= cap` <=> `0 < speed <= speedLimit` <=> min(speed,limit)/speed == 1.
float s2_cap_for(float limit)
{
    if (!(limit > 0.0f)) return -1.0f;                       // never take the shortcut
    if (std::isinf(limit)) return std::numeric_limits<float>::max();
    double sq = (double)limit * (double)limit;
    if (sq >= (double)std::numeric_limits<float>::max()) return std::numeric_limits<float>::max();
    float c = (float)sq;
    while (sqrtf(c) > limit) c = nextafterf(c, 0.0f);
    for (;;) {
        float n = nextafterf(c, std::numeric_limits<float>::infinity());
        if (std::isinf(n) || sqrtf(n) > limit) break;
        c = n;
    }
    return c;
}

bool finite_uniforms(const th_logic_uniforms &u)
{
    const float *f = reinterpret_cast<const float *>(&u);
    for (size_t k = 0; k < sizeof(u) / sizeof(float); ++k)
        if (!std::isfinite(f[k])) return false;
    return true;
}

}  // namespace

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
    th_counters *d_counters = nullptr;
    // th_draw_sharded: the neighbours' edge rows, the owners' counts, what this rank received
    float4 *x_halo = nullptr;            // [lo: cur row, prev row | hi: cur row, prev row], `width` texels each
    unsigned long long *x_counts = nullptr;   // device: bounds (33) | send counts (32) | recv counts (32)
    unsigned long long *x_keys = nullptr;
    float4 *x_colors = nullptr;
    size_t x_capacity = 0;
    float4 *gathered = nullptr;          // row-band shard: a copy of the WHOLE particle texture (th_state_gather / _ptr) ...
    const void *gathered_of = nullptr;   // ... of this ring buffer, for the spawners that sample arbitrary particles
    void *comm = nullptr;                // RCCL communicator of the job's ranks (th_comm_init), one rank per context
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
    uint32_t *chunk_table = nullptr;     // per list x kBinMaxPages: the pages a list has grown by
    unsigned long long *bins_keys = nullptr;   // the page store: (bins x kBinReplicas + bins_pool) pages of kBinPage places - keys (~0 = empty) ...
    float4 *bins_colors = nullptr;       // ... and varyings (two per place once a th_draw has run)
    uint32_t bins_pool = 0, bins_store_bins = 0;
    bool bins_pairs = false;
    uint32_t *crowd_mem = nullptr;       // per large bin: fragments per texel, first fragment of every texel, fill cursors, long runs
    uint32_t crowd_capacity = 0;
    unsigned long long *crowd_keys = nullptr;  // the large bins' fragments regrouped by texel
    uint32_t *crowd_sorted = nullptr;          // ... their places, run by run in blend order
    size_t crowd_keys_cap = 0;
    hipStream_t side = nullptr;                // the long runs of a crowded target are blended beside everything else
    hipEvent_t forked = nullptr, joined = nullptr;
    hipStream_t side2 = nullptr;               // ... and the crowded bins' short runs beside both
    hipEvent_t joined2 = nullptr, regrouped = nullptr;
    uint32_t *bins_totals_host = nullptr;      // (pinned) the binned pass's totals, read back over the side stream
    bool mrg_pairs = false, x_pairs = false;   // the merge / exchange colour buffers hold two varyings per fragment (th_draw_emit / _merge)
    void *pinned = nullptr;                    // (pinned, kPinnedBytes) small read-backs: a pageable hipMemcpyAsync costs ~0.15 ms per call
    int lines_local = -1;                // every vertex of every line reads the line's own particle (line_rows)
    uint32_t *d_row_draws = nullptr;     // bit per global row: the row's lines can draw (line_rows)
    int draw_pipeline = TH_DRAW_AUTO;    // th_draw_pipeline
    th_draw_info last_draw{};            // th_draw_query
    // auto policy: the binned pipeline while the target is not crowded (th_api.hip: draw_uses_bins)
    long long draws = 0, stream_until = 0;
    int crowded_streak = 0, stream_spell = 0;
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
    uchar4 *view = nullptr;              // the view pass's RGBA8 drawing buffer (flow shape), lazily allocated
    int32_t view_w = 0, view_h = 0;
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
    };
    std::vector<SlotOrder> orders;
    std::vector<std::pair<float4 *, int>> buf_order;   // ring buffers held in a sorted order (absent = texel order)
    float4 *spare = nullptr;             // spare state buffer (ensure_identity moves through it)
    uint32_t *tile_mem = nullptr;        // hist | cursor (kSortReplicas x kMaxTileBins words each) | misses
    th::ChunkRecord *block_records = nullptr;   // per 4096-slot block: tile_hist's table for tile_scatter
    uint32_t max_chunks = 0;
    int steps_since_sort = 0;
    unsigned long long sorts = 0;
    long long total_steps = 0, hold_texel_order_until = 0;   // texel-order consumers (draw) keep the layout off for a period
    uint32_t *miss_host = nullptr;       // pinned: window misses since the last sort, as of some recent launch
    // a COUNT pass has histogrammed the tiles of the state it wrote: valid for a SCATTER pass that reads exactly that
    struct { const float4 *buf = nullptr; int order = -1; th::TileGeom geom{}; long long at_step = -1; } counted;

    size_t texels() const { return (size_t)cfg.width * cfg.height; }
    size_t state_bytes() const { return texels() * (packed ? sizeof(uint2) : sizeof(float4)); }
};

namespace {

// keeps_lines: the entry point leaves the particle state and the per-line / per-fragment buffers of the last draw pass
// alone, so that a view pass can still reuse the flow pass's geometry (deposit_run)
th_status use(th_context *c, bool keeps_lines = false)
{
    if (!c) return fail(TH_ERR_INVALID, "null context");
    TH_HIP(hipSetDevice(c->cfg.device));
    if (!keeps_lines) c->drawn.valid = false;
    return TH_OK;
}

th_status alloc_state(th_context *c, float4 **out)
{
    TH_HIP(hipMalloc((void **)out, c->state_bytes()));
    // gl-fbo attachments start zero-filled (all-zero bits are the zero state in the packed format too)
    TH_HIP(hipMemsetAsync(*out, 0, c->state_bytes(), c->stream));
    return TH_OK;
}

th_status resolve_target(th_context *c, int32_t target, bool rotate_ok, float4 **out)
{
    if (target == TH_TARGET_RING) {
        if (!rotate_ok) return fail(TH_ERR_INVALID, "TH_TARGET_RING not valid here");
        float4 *last = c->ring.back();               // utils.step: pop -> unshift
        c->ring.pop_back();
        c->ring.insert(c->ring.begin(), last);
        *out = c->ring[0];
    } else if (target == TH_TARGET_TARGETS) {
        *out = c->targets;
        c->targets_checked = false;
    } else if (target >= 0 && target < (int32_t)c->ring.size()) {
        *out = c->ring[target];
    } else {
        return fail(TH_ERR_INVALID, "bad render target %d (ring has %zu buffers)", target, c->ring.size());
    }
    return TH_OK;
}

th_status rect_ok(th_context *c, int32_t x0, int32_t y0, int32_t w, int32_t h)
{
    TH_REQUIRE(x0 >= 0 && y0 >= 0 && w > 0 && h > 0 && x0 + w <= c->cfg.width && y0 + h <= c->cfg.height,
               "rectangle (%d,%d %dx%d) outside the %dx%d state texture", x0, y0, w, h, c->cfg.width, c->cfg.height);
    return TH_OK;
}


// ---- packed ring: f32 staging for everything except the hot step ---------------------------------
th_status staging(th_context *c, int k, float4 **out)
{
    if (!c->tmp[k]) TH_HIP(hipMalloc((void **)&c->tmp[k], c->texels() * sizeof(float4)));
    *out = c->tmp[k];
    return TH_OK;
}

// f32 view of ring buffer `buf` in staging slot k (a no-op for an f32 ring: returns the buffer itself)
th_status unpacked_view(th_context *c, float4 *buf, int k, float4 **out)
{
    if (!c->packed) { *out = buf; return TH_OK; }
    if (th_status s = staging(c, k, out)) return s;
    th::launch_unpack_state(*out, buf, (uint32_t)c->texels(), c->stream);
    TH_HIP(hipGetLastError());
    return TH_OK;
}

// where a pass that renders into ring buffer `buf` should write (staging slot k when packed) ...
th_status render_target(th_context *c, float4 *buf, int k, float4 **out)
{
    if (!c->packed || buf == c->targets) { *out = buf; return TH_OK; }
    return staging(c, k, out);
}

// ... and the commit of that staging buffer into the packed ring buffer
th_status commit_target(th_context *c, float4 *buf, float4 *rendered)
{
    if (rendered == buf) return TH_OK;
    th::launch_pack_state(buf, rendered, (uint32_t)c->texels(), c->stream);
    TH_HIP(hipGetLastError());
    return TH_OK;
}

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
// profiles/r1_c_bucketing.txt, r2_b_*).  TH_BUCKET=0/1 forces the layout off/on (the parity suite reruns under 1);
// TH_RESORT_STEPS / TH_REBUCKET_STEPS set the re-sort period of single-step / fused launches.
int bucket_policy()
{
    static const int v = [] { const char *e = getenv("TH_BUCKET"); return e ? atoi(e) : -1; }();
    return v;
}
int rebucket_period()
{
    static const int v = [] { const char *e = getenv("TH_REBUCKET_STEPS"); int n = e ? atoi(e) : 256; return n > 0 ? n : 256; }();
    return v;
}
int resort_period()
{
    static const int v = [] { const char *e = getenv("TH_RESORT_STEPS"); int n = e ? atoi(e) : 64; return n > 0 ? n : 64; }();
    return v;
}
constexpr int kTileShift = 5;            // 32 x 32 texel tiles (th_kernels.hip kTile)
constexpr size_t kTileWords = 2 * (size_t)th::kSortReplicas * th::kMaxTileBins;   // histogram + cursors, all copies
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
    if (bucket_policy() == 0) return false;
    if (bucket_policy() == 1) return true;
    return c->texels() >= ((size_t)1 << 20) && flow_texels * sizeof(float2) > ((size_t)3 << 20);
}

// Which rows of the state texture can draw() make lines of, and does every vertex of every line read the line's OWN
// particle?  Particles.generateLUT writes the vertex coordinates as i/(W-1), j/(2H-1) (src/particles.js:171-190) and the
// shader turns them back into a texel and a buffer with fp32 arithmetic (src/state/state-at-frame.glsl:12-22): vertex
// 2m of line m reads `previous` in the lower rows and `current` in the upper ones, vertex 2m+1 `current` - so the lines
// of the upper half (both vertices the same texel of the same buffer) have no length; and for some shapes (W >= 8192;
// heights such as 100, 1080, 3000) the lookup of a few rows / columns lands one texel beside the line's own.
// Same operations as dep_fetch (th_raster.hpp).  Bit m of the table: row m can draw.
th_status line_rows(th_context *c)
{
    if (c->d_row_draws) return TH_OK;
    const int W = c->cfg.width, H = c->cfg.global_height;
    const double inv_x = 1.0 / (double)((W > 2 ? W : 2) - 1), inv_y = 1.0 / (double)((2 * H > 2 ? 2 * H : 2) - 1);
    auto nearest = [](float u, int n) { const float f = floorf(u * (float)n); return !(f > 0.0f) ? 0 : (f > (float)(n - 1) ? n - 1 : (int)f); };
    bool local = true;
    for (int i = 0; i < W && local; ++i) local = nearest((float)((double)i * inv_x), W) == i;
    std::vector<uint32_t> bits(((size_t)H + 31) / 32, 0u);
    for (int m = 0; m < H; ++m) {
        int row[2];
        bool cur[2];
        for (int v = 0; v < 2; ++v) {
            const float uvy = (float)((double)(2 * m + v) * inv_y), near_index = uvy * (float)H, fl = floorf(near_index);
            cur[v] = near_index - fl > 0.25f;
            row[v] = nearest(fl / (float)H, H);
            local = local && row[v] == m;
        }
        if (!(row[0] == row[1] && cur[0] == cur[1])) bits[(size_t)m >> 5] |= 1u << (m & 31);
    }
    TH_HIP(hipMalloc((void **)&c->d_row_draws, bits.size() * sizeof(uint32_t)));
    TH_HIP(hipMemcpy(c->d_row_draws, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->lines_local = local ? 1 : 0;
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
    TH_HIP(hipMalloc((void **)&c->tile_mem, (kTileWords + 8) * sizeof(uint32_t)));
    TH_HIP(hipMemsetAsync(c->tile_mem, 0, (kTileWords + 8) * sizeof(uint32_t), c->stream));
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

// every ring buffer back to texel order (reports whether anything was launched)
th_status ensure_identity(th_context *c, bool *launched = nullptr)
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
        float4 *t = b; b = c->spare; c->spare = t;
        if (launched) *launched = true;
    }
    TH_HIP(hipGetLastError());
    return TH_OK;
}

// Count the tiles of `state` (any slot order) and lay out a new order for it: tile starts, rank cursors, chunk table.
// The slots themselves are assigned by the kernel that moves the state (tile_scatter_kernel or a SCATTER step).
th_status begin_sort(th_context *c, const th::TileGeom &g, const float4 *state, const uint32_t *perm_in, int *order,
                     th::TileSortParams *params, bool have_hist = false)
{
    if (th_status s = sort_storage(c)) return s;
    if (th_status s = free_order(c, order)) return s;
    th_context::SlotOrder &o = c->orders[(size_t)*order];
    o.geom = g; o.fw = c->fw; o.fh = c->fh;
    th::TileSortParams b{};
    b.state = state; b.perm_in = perm_in; b.count = (uint32_t)c->texels();
    b.g = g;
    b.hist = c->tile_mem; b.cursor = c->tile_mem + kTileWords / 2;
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
    c->counted.buf = nullptr;
    if (params) *params = b;
    return TH_OK;
}

}  // namespace

extern "C" {

int32_t th_abi_version(void) { return TH_ABI_VERSION; }

const char *th_last_error(void) { return g_error.c_str(); }

th_status th_device_count(int32_t *count)
{
    TH_REQUIRE(count, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(TH_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return TH_OK;
}

th_status th_create(const th_config *cfg, th_context **out)
{
    TH_REQUIRE(cfg && out, "null argument");
    *out = nullptr;
    TH_REQUIRE(cfg->width > 0 && cfg->height > 0, "state shape must be positive (got %dx%d)", cfg->width, cfg->height);
    TH_REQUIRE(cfg->num_buffers >= 0 && cfg->num_buffers <= 64, "num_buffers out of range");
    TH_REQUIRE(cfg->mode == TH_MODE_EXACT || cfg->mode == TH_MODE_FAST, "unknown mode %d", cfg->mode);
    TH_REQUIRE(cfg->state_format == TH_STATE_F32 || cfg->state_format == TH_STATE_F16, "unknown state format %d", cfg->state_format);
    TH_REQUIRE((uint64_t)cfg->width * (uint64_t)cfg->height < (1ull << 31), "more than 2^31 texels per context");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(TH_ERR_NO_DEVICE, "no HIP device available");
    TH_REQUIRE(cfg->device >= 0 && cfg->device < n, "device %d out of range (have %d)", cfg->device, n);
    hipDeviceProp_t prop;
    TH_HIP(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(TH_ERR_NO_DEVICE, "device %d is %s; this library ships gfx950 code only", cfg->device, prop.gcnArchName);

    th_context *c = new (std::nothrow) th_context;
    if (!c) return fail(TH_ERR_INVALID, "out of host memory");
    c->cfg = *cfg;
    c->packed = cfg->state_format == TH_STATE_F16;
    if (c->cfg.global_height <= 0) c->cfg.global_height = c->cfg.height;
    if (c->cfg.row0 < 0 || c->cfg.row0 + c->cfg.height > c->cfg.global_height) {
        delete c;
        return fail(TH_ERR_INVALID, "row band [%d,%d) outside global height %d", cfg->row0, cfg->row0 + cfg->height, cfg->global_height);
    }
    th_status st = TH_OK;
    auto body = [&]() -> th_status {
        TH_HIP(hipSetDevice(c->cfg.device));
        TH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        TH_HIP(hipEventCreate(&c->ev0));
        TH_HIP(hipEventCreate(&c->ev1));
        // one block: [hash tables (filled on the device) | gradient table]; c->lut points at the gradient table
        const size_t hv = (size_t)th::hash_table_vectors();
        TH_HIP(hipMalloc((void **)&c->lut_block, (hv + th::kLutSize) * sizeof(float4)));
        c->lut = c->lut_block + hv;
        std::vector<float4> lut(th::kLutSize);
        build_gradient_table(lut.data());
        TH_HIP(hipMemcpy(c->lut, lut.data(), lut.size() * sizeof(float4), hipMemcpyHostToDevice));
        th::launch_hash_tables(c->lut_block, c->stream);
        TH_HIP(hipGetLastError());
        TH_HIP(hipMalloc((void **)&c->d_flag, sizeof(unsigned int)));
        TH_HIP(hipMalloc((void **)&c->partials, th::kStatsBlocks * sizeof(th::StatsPartial)));
        TH_HIP(hipMalloc((void **)&c->d_counters, sizeof(th_counters)));
        TH_HIP(hipMalloc((void **)&c->d_respawned, 2 * sizeof(unsigned long long)));
        TH_HIP(hipMemset(c->d_respawned, 0, 2 * sizeof(unsigned long long)));
        // Tendrils ctor: flow and targets start as 1x1 float FBOs (src/index.js:102-105);
        // setupParticles gives targets the particle shape (src/index.js:207).
        c->fw = c->fh = 1;
        TH_HIP(hipMalloc((void **)&c->flow, sizeof(float4)));
        TH_HIP(hipMemsetAsync(c->flow, 0, sizeof(float4), c->stream));
        TH_HIP(hipMalloc((void **)&c->flow_dec, sizeof(float2)));
        // targets is an RGBA32F texture in every state format
        TH_HIP(hipMalloc((void **)&c->targets, c->texels() * sizeof(float4)));
        TH_HIP(hipMemsetAsync(c->targets, 0, c->texels() * sizeof(float4), c->stream));
        for (int k = 0; k < c->cfg.num_buffers; ++k) {
            float4 *b = nullptr;
            if (th_status s = alloc_state(c, &b)) return s;
            c->ring.push_back(b);
        }
        if (th_status s = line_rows(c)) return s;          // (before anything builds a TileGeom)
        TH_HIP(hipStreamSynchronize(c->stream));
        return TH_OK;
    };
    st = body();
    if (st != TH_OK) { std::string keep = g_error; th_destroy(c); g_error = keep; return st; }
    *out = c;
    return TH_OK;
}

th_status th_destroy(th_context *c)
{
    if (!c) return TH_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { (void)th::comm_destroy(c->comm); c->comm = nullptr; }
    for (float4 *b : c->ring) (void)hipFree(b);
    (void)hipFree(c->flow); (void)hipFree(c->flow_dec); (void)hipFree(c->flow3); (void)hipFree(c->targets); (void)hipFree(c->lut_block);
    (void)hipFree(c->frames[0]); (void)hipFree(c->frames[1]);
    (void)hipFree(c->dep_count); (void)hipFree(c->dep_offset); (void)hipFree(c->dep_blocks); (void)hipFree(c->dep_total);
    (void)hipFree(c->dep_record); (void)hipFree(c->dep_lists); (void)hipFree(c->mrg_keys2); (void)hipFree(c->mrg_colors);
    (void)hipFree(c->bin_mem); (void)hipFree(c->d_row_draws); (void)hipFree(c->crowd_mem); (void)hipFree(c->chunk_table);
    (void)hipFree(c->bins_keys); (void)hipFree(c->bins_colors); (void)hipFree(c->crowd_keys); (void)hipFree(c->crowd_sorted); (void)hipFree(c->gathered);
    if (c->forked) (void)hipEventDestroy(c->forked);
    if (c->joined) (void)hipEventDestroy(c->joined);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->joined2) (void)hipEventDestroy(c->joined2);
    if (c->regrouped) (void)hipEventDestroy(c->regrouped);
    if (c->side2) (void)hipStreamDestroy(c->side2);
    if (c->bins_totals_host) (void)hipHostFree(c->bins_totals_host);
    if (c->pinned) (void)hipHostFree(c->pinned);
    (void)hipFree(c->x_halo); (void)hipFree(c->x_counts); (void)hipFree(c->x_keys); (void)hipFree(c->x_colors);
    for (uint32_t *q : c->dep_u32) (void)hipFree(q);
    for (unsigned long long *q : c->dep_u64) (void)hipFree(q);
    (void)hipFree(c->dep_colors_sorted); (void)hipFree(c->mrg_keys); (void)hipFree(c->mrg_vals[0]); (void)hipFree(c->mrg_vals[1]);
    (void)hipFree(c->dep_colors); (void)hipFree(c->dep_temp);
    (void)hipFree(c->image); (void)hipFree(c->view); (void)hipFree(c->colormap);
    (void)hipFree(c->d_flag); (void)hipFree(c->partials); (void)hipFree(c->d_counters); (void)hipFree(c->d_respawned);
    clear_graphs(c);
    for (float4 *t : c->tmp) (void)hipFree(t);
    for (th_context::SlotOrder &o : c->orders) { (void)hipFree(o.perm); (void)hipFree(o.chunks); (void)hipFree(o.records); (void)hipFree(o.nchunks); }
    (void)hipFree(c->spare); (void)hipFree(c->tile_mem); (void)hipFree(c->block_records);
    if (c->miss_host) (void)hipHostFree(c->miss_host);
    for (hipEvent_t e : c->kt_events) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TH_OK;
}

th_status th_set_mode(th_context *c, int32_t mode)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(mode == TH_MODE_EXACT || mode == TH_MODE_FAST, "unknown mode %d", mode);
    c->cfg.mode = mode;
    return TH_OK;
}

th_status th_setup(th_context *c, int32_t num_buffers)
{
    if (th_status s = use(c)) return s;
    clear_graphs(c);
    if (th_status s = ensure_identity(c)) return s;      // these operate in texel order
    TH_REQUIRE(num_buffers >= 0 && num_buffers <= 64, "num_buffers out of range");
    while ((int32_t)c->ring.size() < num_buffers) {          // src/particles.js:83-86: push
        float4 *b = nullptr;
        if (th_status s = alloc_state(c, &b)) return s;
        c->ring.push_back(b);
    }
    if ((int32_t)c->ring.size() > num_buffers) TH_HIP(hipStreamSynchronize(c->stream));
    while ((int32_t)c->ring.size() > num_buffers) {          // :89-91: pop().dispose()
        TH_HIP(hipFree(c->ring.back()));
        c->ring.pop_back();
    }
    c->cfg.num_buffers = num_buffers;
    return TH_OK;
}

th_status th_num_buffers(th_context *c, int32_t *out)
{
    TH_REQUIRE(c && out, "null argument");
    *out = (int32_t)c->ring.size();
    return TH_OK;
}

th_status th_upload_state(th_context *c, int32_t buffer, const float *rgba, int32_t x0, int32_t y0, int32_t w, int32_t h)
{
    if (th_status s = use(c)) return s;
    if (th_status s = ensure_identity(c)) return s;      // these operate in texel order
    TH_REQUIRE(rgba, "null pixels");
    if (th_status s = rect_ok(c, x0, y0, w, h)) return s;
    TH_REQUIRE(buffer >= -1 && buffer < (int32_t)c->ring.size(), "bad buffer index %d", buffer);
    int first = buffer < 0 ? 0 : buffer, last = buffer < 0 ? (int)c->ring.size() - 1 : buffer;
    for (int b = first; b <= last; ++b) {
        float4 *view = nullptr;                 // packed ring: edit an f32 copy, then re-pack
        if (th_status s = unpacked_view(c, c->ring[b], 0, &view)) return s;
        float4 *dst = view + (size_t)y0 * c->cfg.width + x0;
        TH_HIP(hipMemcpy2DAsync(dst, (size_t)c->cfg.width * sizeof(float4), rgba, (size_t)w * sizeof(float4),
                                (size_t)w * sizeof(float4), h, hipMemcpyHostToDevice, c->stream));
        if (th_status s = commit_target(c, c->ring[b], view)) return s;
    }
    TH_HIP(hipStreamSynchronize(c->stream));   // the caller may reuse `rgba` immediately (setPixels semantics)
    return TH_OK;
}

th_status th_download_state(th_context *c, int32_t buffer, float *rgba, int32_t x0, int32_t y0, int32_t w, int32_t h)
{
    if (th_status s = use(c)) return s;
    if (th_status s = ensure_identity(c)) return s;      // these operate in texel order
    TH_REQUIRE(rgba, "null pixels");
    if (th_status s = rect_ok(c, x0, y0, w, h)) return s;
    TH_REQUIRE(buffer >= 0 && buffer < (int32_t)c->ring.size(), "bad buffer index %d", buffer);
    float4 *view = nullptr;
    if (th_status s = unpacked_view(c, c->ring[buffer], 0, &view)) return s;
    const float4 *src = view + (size_t)y0 * c->cfg.width + x0;
    TH_HIP(hipMemcpy2DAsync(rgba, (size_t)w * sizeof(float4), src, (size_t)c->cfg.width * sizeof(float4),
                            (size_t)w * sizeof(float4), h, hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_flow_resize(th_context *c, int32_t w, int32_t h)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(w > 0 && h > 0 && w < (1 << 24) && h < (1 << 24) && (uint64_t)w * h < (1ull << 28), "bad flow shape %dx%d", w, h);
    if (w == c->fw && h == c->fh) return TH_OK;              // gl-fbo: same shape is a no-op
    TH_HIP(hipStreamSynchronize(c->stream));
    clear_graphs(c);
    TH_HIP(hipFree(c->flow));
    TH_HIP(hipFree(c->flow_dec));
    (void)hipFree(c->flow3);
    c->flow = nullptr; c->flow_dec = nullptr; c->flow3 = nullptr;
    TH_HIP(hipMalloc((void **)&c->flow, (size_t)w * h * sizeof(float4)));
    TH_HIP(hipMalloc((void **)&c->flow_dec, (size_t)w * h * sizeof(float2)));
    TH_HIP(hipMemsetAsync(c->flow, 0, (size_t)w * h * sizeof(float4), c->stream));
    c->fw = w; c->fh = h;
    return TH_OK;
}

th_status th_flow_upload(th_context *c, const float *rgba)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(rgba, "null pixels");
    TH_HIP(hipMemcpyAsync(c->flow, rgba, (size_t)c->fw * c->fh * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_flow_download(th_context *c, float *rgba)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(rgba, "null pixels");
    TH_HIP(hipMemcpyAsync(rgba, c->flow, (size_t)c->fw * c->fh * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_flow_clear(th_context *c)
{
    if (th_status s = use(c)) return s;
    TH_HIP(hipMemsetAsync(c->flow, 0, (size_t)c->fw * c->fh * sizeof(float4), c->stream));
    return TH_OK;
}

th_status th_targets_upload(th_context *c, const float *rgba)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(rgba, "null pixels");
    TH_HIP(hipMemcpyAsync(c->targets, rgba, c->texels() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    c->targets_checked = false;
    return TH_OK;
}

th_status th_targets_download(th_context *c, float *rgba)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(rgba, "null pixels");
    TH_HIP(hipMemcpyAsync(rgba, c->targets, c->texels() * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_targets_clear(th_context *c)
{
    if (th_status s = use(c)) return s;
    TH_HIP(hipMemsetAsync(c->targets, 0, c->texels() * sizeof(float4), c->stream));
    c->targets_checked = true;
    c->targets_nonfinite = false;
    return TH_OK;
}

// Build the launch parameters of one integrator pass and pick the kernel variant.
// ---- one integrator pass = plan (host decisions, may synchronise) + enqueue (launches only) -------
struct StepPlan {
    th::LogicParams p{};         // everything except in / out / perm / time_dev
    bool noise = false, use_targets = false, pow2 = false, decoded = false, generic = false;
    bool may_sort = false;       // this pass may run on (and produce) tile-sorted slots
};

// Pick the kernel variant and bring the slot layout up to date.  `u.time` must be the time of
// largest magnitude the plan will be used with (it only enters the domain checks here).
static th_status plan_step(th_context *c, const th_logic_uniforms &u, int32_t target, StepPlan &plan)
{
    const uint32_t W = (uint32_t)c->cfg.width, H = (uint32_t)c->cfg.global_height;
    th::LogicParams &p = plan.p;
    p = th::LogicParams{};
    p.flow = c->flow; p.flow_dec = c->flow_dec; p.targets = c->targets; p.lut = c->lut;
    p.count = (uint32_t)c->texels();
    p.width = W;
    p.row0 = (uint32_t)c->cfg.row0;
    p.wf = (float)W; p.hf = (float)H;
    plan.pow2 = is_pow2(W) && is_pow2(H);
    p.log2w = plan.pow2 ? ilog2(W) : 0;
    p.inv_w = 1.0f / p.wf; p.inv_h = 1.0f / p.hf; p.inv_wh = 1.0f / (p.wf * p.hf);
    p.fw = c->fw; p.fh = c->fh;
    p.fwf = (float)c->fw; p.fhf = (float)c->fh;
    p.half_fw = 0.5f * p.fwf; p.half_fh = 0.5f * p.fhf;
    p.fwm1 = (float)(c->fw - 1); p.fhm1 = (float)(c->fh - 1);
    p.u = u;
    p.s2_cap = s2_cap_for(u.speedLimit);

    // Preconditions of the specialised path (DESIGN.md "fast-path domain").
    static const bool force_generic = getenv("TH_FORCE_GENERIC") != nullptr;   // test hook
    plan.generic = force_generic || !finite_uniforms(u);
    plan.noise = u.noiseWeight != 0.0f;
    plan.use_targets = u.target != 0.0f;
    if (!plan.generic) {
        // i = (x+.5 + (y+.5)W)/(WH) lies in (0, 1]; bound |vary(base, i, v)| <= |base|(1+|v|)
        double nscale = std::fabs((double)u.noiseScale) * (1.0 + std::fabs((double)u.varyNoiseScale)) * 1.001;
        double ntime = std::fabs((double)u.time) * std::fabs((double)u.noiseSpeed) *
                       (1.0 + std::fabs((double)u.varyNoiseSpeed)) * 1.001;
        if (ntime + 1237.0 >= (double)th::kNoiseDomain) plan.generic = true;  // z = uv + noiseTime (+1234.5678)
        double bound = nscale > 0.0 ? (double)th::kNoiseDomain / nscale : 3.0e38;
        // capped below |inert| = 1e6: a lane inside the bound cannot be inert, so the specialised path tests
        // the bound only and the inert pass-through lives on the (reference-order) fallback path
        p.pos_bound = (float)std::fmin(bound * 0.999, 999999.0);
        if (!(p.pos_bound > 0.0f)) plan.generic = true;
    }
    if (!plan.generic && !plan.use_targets) {
        // target == 0 multiplies (targets - pos) by an exact zero; dropping the read is only
        // value-preserving when the texture holds no NaN/Inf.
        if (!c->targets_checked) {
            unsigned int flag = 0;
            TH_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(unsigned int), c->stream));
            th::launch_finite_check(c->targets, c->texels(), c->d_flag, c->stream);
            TH_HIP(hipMemcpyAsync(&flag, c->d_flag, sizeof flag, hipMemcpyDeviceToHost, c->stream));
            TH_HIP(hipStreamSynchronize(c->stream));
            c->targets_nonfinite = flag != 0;
            c->targets_checked = true;
        }
        plan.use_targets = c->targets_nonfinite;
    }
    // Decode the flow once per step when that is cheaper than decoding per particle: it shrinks the
    // random-gather footprint (the L2/Infinity-Fabric miss traffic is what bounds this kernel).
    const size_t flow_texels = (size_t)c->fw * c->fh;
    plan.decoded = !plan.generic && c->texels() >= 2 * flow_texels;

    // Slot layout (texel order or a tile-sorted order): only ring -> ring passes of the specialised f32 kernels run on
    // sorted slots; the callers bring the layout up to date.
    plan.may_sort = plan.decoded && target == TH_TARGET_RING && sorting_possible(c) &&
                    c->total_steps >= c->hold_texel_order_until;
    return TH_OK;
}

// what a captured th_step_n sequence depends on besides the ring order and the kernel flags (`time` excluded: it lives in
// device memory); an explicit field list - the struct has padding and fields the captured launches never read
static bool same_key(const th::LogicParams &a, const th::LogicParams &b)
{
    th_logic_uniforms ua = a.u, ub = b.u;
    ua.time = ub.time = 0.0f;
    return a.flow == b.flow && a.flow_dec == b.flow_dec && a.targets == b.targets && a.lut == b.lut &&
           a.count == b.count && a.width == b.width && a.log2w == b.log2w && a.row0 == b.row0 &&
           a.wf == b.wf && a.hf == b.hf && a.fw == b.fw && a.fh == b.fh &&
           memcmp(&ua, &ub, sizeof ua) == 0 && a.s2_cap == b.s2_cap && a.pos_bound == b.pos_bound;
}

static uint32_t plan_flags(const StepPlan &plan)
{
    return (plan.noise ? 1u : 0u) | (plan.use_targets ? 2u : 0u) | (plan.pow2 ? 4u : 0u) | (plan.decoded ? 8u : 0u) |
           (plan.generic ? 16u : 0u);
}

static th_status timing_events(th_context *c, hipEvent_t *k0, hipEvent_t *k1)
{
    if (c->kt_used + 2 > c->kt_events.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        TH_HIP(hipEventCreate(&a)); TH_HIP(hipEventCreate(&b));
        c->kt_events.push_back(a); c->kt_events.push_back(b);
    }
    *k0 = c->kt_events[c->kt_used]; *k1 = c->kt_events[c->kt_used + 1];
    c->kt_used += 2;
    return TH_OK;
}

// Rotate / resolve the render target and launch (flow decode +) the integrator.  Launches only:
// safe inside a stream capture.  `time_dev` (optional) overrides plan.p.u.time on the device.
// `sorted`: the pass may read and write tile-sorted slots (else every ring buffer is in texel order already).
static th_status enqueue_step(th_context *c, const StepPlan &plan, int32_t target, float time, const float *time_dev,
                              bool timing, bool sorted = false)
{
    th::LogicParams p = plan.p;
    float4 *out = nullptr;
    if (th_status s = resolve_target(c, target, true, &out)) return s;
    // A packed (TH_STATE_F16) ring runs the packed kernel on the default path (ring -> ring, specialised
    // kernel); explicit targets and the generic kernel go through f32 staging.
    const bool packed_kernel = c->packed && target == TH_TARGET_RING && !plan.generic;
    float4 *in = c->ring[1], *rt = out;     // Particles.step binds buffers[1] as `particles` (src/particles.js:139)
    if (c->packed && !packed_kernel) {
        if (th_status s = unpacked_view(c, c->ring[1], 1, &in)) return s;
        if (th_status s = render_target(c, out, 0, &rt)) return s;
    }
    p.in = in;
    p.out = rt;
    p.u.time = time;
    p.time_dev = time_dev;

    // Sorted slots.  The input keeps its order; the output is written either at the same slots or - every
    // resort_period() steps, and when the input is not sorted yet or was sorted for another view / field shape - at
    // the slots of a new sort keyed on the input positions (counted just before the launch).
    int in_order = sorted ? order_of(c, in) : -1, out_order = -1;
    bool use_sorted = false, scatter = false, count = false, gather = false;
    if (sorted && packed_kernel) {
        // packed ring: the plain grid-stride kernel over the sorted slots; a re-sort is a plain move of the input
        // (tile_hist, scan, tile_scatter into the spare buffer, which then takes the input's place in the ring)
        const th::TileGeom g = tile_geom(c, p.u);
        const bool stale = in_order >= 0 && (!same_geom(c->orders[(size_t)in_order].geom, g) ||
                                             c->orders[(size_t)in_order].fw != c->fw || c->orders[(size_t)in_order].fh != c->fh);
        if (in_order < 0 || stale || c->steps_since_sort >= resort_period()) {
            int fresh = -1;
            th::TileSortParams b;
            if (th_status s = begin_sort(c, g, in, in_order >= 0 ? c->orders[(size_t)in_order].perm : nullptr, &fresh, &b)) return s;
            b.state_out = c->spare;
            th::launch_tile_scatter(b, c->stream);
            TH_HIP(hipGetLastError());
            clear_graphs(c);               // captured sequences name the ring buffers: one of them changes places with the spare
            float4 *old = in;
            for (float4 *&r : c->ring) if (r == old) r = c->spare;
            c->spare = old;
            set_order(c, old, -1);
            in = c->ring[1];
            set_order(c, in, fresh);
            p.in = in;
            in_order = fresh;
        }
        p.perm = c->orders[(size_t)in_order].perm;
        out_order = in_order;
    } else if (sorted) {
        const th::TileGeom g = tile_geom(c, p.u);
        const bool stale = in_order >= 0 && (!same_geom(c->orders[(size_t)in_order].geom, g) ||
                                             c->orders[(size_t)in_order].fw != c->fw || c->orders[(size_t)in_order].fh != c->fh);
        if (stale) {                       // the chunk table no longer describes the field: start over from texel order
            if (th_status s = ensure_identity(c)) return s;
            in = c->ring[1]; out = rt = c->ring[0];
            p.in = in; p.out = rt;
            in_order = -1;
        }
        scatter = in_order < 0 || c->steps_since_sort >= resort_period();
        use_sorted = true;
        // between two sorts the pass is the plain grid-stride kernel over the sorted slots (taps gathered from the
        // decoded plane: a wave's taps fall into one neighbourhood); the chunk kernel counts and scatters around a re-sort
        gather = !scatter && plan.decoded && c->steps_since_sort + 1 < resort_period();
        p.geom = g;
        if (in_order >= 0) {
            const th_context::SlotOrder &o = c->orders[(size_t)in_order];
            p.perm = o.perm; p.chunks = o.chunks; p.nchunks = o.nchunks; p.records = o.records;
        }
        if (scatter) {
            // counted by the pass that wrote `in`?  Then the histogram is complete and every chunk has its table.
            const bool counted = in_order >= 0 && c->counted.buf == in && c->counted.order == in_order &&
                                 same_geom(c->counted.geom, g) && c->counted.at_step == c->total_steps;
            set_order(c, out, -1);         // the output buffer's old content (and order) dies here
            th::TileSortParams b;
            if (th_status s = begin_sort(c, g, in, in_order >= 0 ? c->orders[(size_t)in_order].perm : nullptr, &out_order, &b, counted)) return s;
            p.cursor = b.cursor; p.perm_out = b.perm_out;
            p.use_records = counted ? 1u : 0u;
            // draws over the slot order are going on (th_bins.hip): the pass moves its INPUT along to the new slots, so
            // that buffers[0] and buffers[1] - the two ends of every line - stay in one order
            if (in == c->ring[1] && c->total_steps - c->last_binned_draw <= 2ll * resort_period()) p.in_moved = c->spare;
        } else {
            out_order = in_order;
            count = !gather && c->steps_since_sort + 1 >= resort_period();      // the next pass will re-sort: count for it
            if (count) {
                if (th_status s = sort_storage(c)) return s;
                p.hist = c->tile_mem;
                TH_HIP(hipMemsetAsync(p.hist, 0, kTileWords / 2 * sizeof(uint32_t), c->stream));
            }
        }
    }

    if (plan.decoded)
        th::launch_flow_decode(c->flow, c->flow_dec, (size_t)c->fw * c->fh, time, time_dev, p.u.flowDecay, c->stream);

    hipEvent_t k0 = nullptr, k1 = nullptr;
    if (timing && c->kernel_timing) {
        if (th_status s = timing_events(c, &k0, &k1)) return s;
        TH_HIP(hipEventRecord(k0, c->stream));
    }
    if (gather) {
        th::launch_logic(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, plan.decoded, plan.generic, packed_kernel, c->stream);
    } else if (use_sorted)
        th::launch_logic_sorted(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, in_order >= 0, scatter, count, c->max_chunks, c->stream);
    else
        th::launch_logic(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, plan.decoded, plan.generic, packed_kernel,
                         c->stream);
    if (k1) TH_HIP(hipEventRecord(k1, c->stream));
    TH_HIP(hipGetLastError());
    if (target == TH_TARGET_RING || (target >= 0 && target < (int32_t)c->ring.size())) set_order(c, out, out_order);
    if (p.in_moved) {                       // the moved copy takes the input's place in the ring
        clear_graphs(c);
        float4 *old = c->ring[1];
        set_order(c, old, -1);
        c->ring[1] = c->spare; c->spare = old;
        set_order(c, c->ring[1], out_order);
    }
    if (c->packed && !packed_kernel)
        if (th_status s = commit_target(c, out, rt)) return s;
    ++c->steps_since_sort; ++c->total_steps;
    if (count) { c->counted.buf = out; c->counted.order = out_order; c->counted.geom = p.geom; c->counted.at_step = c->total_steps; }
    else c->counted.buf = nullptr;
    return TH_OK;
}

th_status th_step(th_context *c, const th_logic_uniforms *u, int32_t target)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(u, "null uniforms");
    // Particles.step reads this.buffers[1] (src/particles.js:139): needs >= 2 buffers
    TH_REQUIRE(c->ring.size() >= 2, "step needs at least 2 state buffers (have %zu)", c->ring.size());
    StepPlan plan;
    if (th_status s = plan_step(c, *u, target, plan)) return s;
    const bool sorted = plan.may_sort && !plan.generic;
    if (!sorted) if (th_status s = ensure_identity(c)) return s;
    return enqueue_step(c, plan, target, u->time, nullptr, true, sorted);
}

// n fixed-step Tendrils.step() calls.  The launch sequence (2 kernels per step) is captured once into
// a hipGraph per (n, uniforms, ring order, layout) and replayed; the per-step `time` values live in a
// small device array refreshed before every replay, so replays need no node updates.
th_status th_step_n(th_context *c, const th_logic_uniforms *u, double time0, double dt_ms, int32_t n)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(u && n >= 0, "bad arguments");
    TH_REQUIRE(c->ring.size() >= 2, "step needs at least 2 state buffers (have %zu)", c->ring.size());
    if (n == 0) return TH_OK;
    th_logic_uniforms v = *u;
    v.dt = (float)dt_ms;
    std::vector<float> times((size_t)n);
    double t = time0, tmax = 0.0;
    for (int32_t k = 0; k < n; ++k) {
        t += dt_ms;                                   // src/timer.js:28-31: time accumulates in double
        times[(size_t)k] = (float)t;
        if (std::fabs(t) > std::fabs(tmax)) tmax = t;
    }
    v.time = (float)tmax;
    StepPlan plan;
    if (th_status s = plan_step(c, v, TH_TARGET_RING, plan)) return s;

    // Temporal fusion (logic_fused_kernel): all n steps of a particle in one pass, <= kMaxFusedSteps per launch.
    // Needs the plain 2-buffer ring (only the last two states survive n rotations) and the specialised kernel;
    // both ring formats.  TH_FUSE=0 turns it off (the tests compare both paths).
    static const bool fuse_on = [] { const char *e = getenv("TH_FUSE"); return !e || atoi(e) != 0; }();
    if (fuse_on && n >= 2 && c->ring.size() == 2 && !plan.generic) {
        // Slot layout of the fused passes: the newest state (ring[0]) may be in a tile-sorted order; both outputs of a
        // pass keep the slots of its input.  (Re)sorted every rebucket_period() steps by a plain move into the other
        // buffer, whose content (state n-1 of the previous call) the pass overwrites anyway.
        if (plan.may_sort) {
            const th::TileGeom g = tile_geom(c, plan.p.u);
            int o = order_of(c, c->ring[0]);
            const bool stale = o >= 0 && (!same_geom(c->orders[(size_t)o].geom, g) || c->orders[(size_t)o].fw != c->fw ||
                                          c->orders[(size_t)o].fh != c->fh);
            if (o < 0 || stale || c->steps_since_sort >= rebucket_period()) {
                float4 *cur = c->ring[0], *other = c->ring[1];
                set_order(c, other, -1);
                int fresh = -1;
                th::TileSortParams b;
                if (th_status s = begin_sort(c, g, cur, o >= 0 ? c->orders[(size_t)o].perm : nullptr, &fresh, &b)) return s;
                b.state_out = other;
                th::launch_tile_scatter(b, c->stream);
                TH_HIP(hipGetLastError());
                set_order(c, other, fresh);
                set_order(c, cur, -1);                 // (its content is dead: the sorted copy is the newest state now)
                c->ring[0] = other; c->ring[1] = cur;
            }
        } else if (th_status s = ensure_identity(c)) return s;
        // The field does not change inside the call.  Without the noise the pass waits for its taps (a dependent gather per
        // step): the field's x, y, z packed 12 B apart once per call - three quarters of the footprint, and the band one
        // XCD taps fits its L2 (0.574 -> 0.546 ms per 20-step launch at C3; with the noise on the pass is bound by its
        // arithmetic and the packing pass only costs: 1.829 against 1.818 + 0.01)
        const bool pack3 = !plan.noise;
        if (pack3) {
            if (!c->flow3) TH_HIP(hipMalloc((void **)&c->flow3, (size_t)c->fw * c->fh * 3 * sizeof(float)));
            th::launch_flow_pack3(c->flow, c->flow3, (size_t)c->fw * c->fh, c->stream);
        }
        {
            int32_t done = 0;
            while (done < n) {
                const int32_t m = std::min<int32_t>(n - done, (int32_t)th::kMaxFusedSteps);
                th::LogicParams p = plan.p;
                p.flow3 = pack3 ? c->flow3 : nullptr;
                float4 *cur = c->ring[0], *other = c->ring[1];
                const int order = order_of(c, cur);
                p.in = cur;
                // a lane only ever touches its own texel, so one of the two outputs may overwrite the input;
                // after m rotations of [cur, other]: m even -> [cur, other], m odd -> [other, cur]
                p.out = (m & 1) ? other : cur;             // state m     (ends up in buffers[0])
                p.out_prev = (m & 1) ? cur : other;        // state m - 1 (ends up in buffers[1])
                p.perm = order >= 0 ? c->orders[(size_t)order].perm : nullptr;
                p.nsteps = (uint32_t)m;
                for (int32_t k = 0; k < m; ++k) p.times[k] = times[(size_t)(done + k)];
                hipEvent_t k0 = nullptr, k1 = nullptr;
                if (c->kernel_timing) {
                    if (th_status s = timing_events(c, &k0, &k1)) return s;
                    TH_HIP(hipEventRecord(k0, c->stream));
                }
                th::launch_logic_fused(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, c->packed, c->stream);
                if (k1) TH_HIP(hipEventRecord(k1, c->stream));
                TH_HIP(hipGetLastError());
                set_order(c, other, order);                // both outputs sit at the input's slots
                c->counted.buf = nullptr;
                if (m & 1) { c->ring[0] = other; c->ring[1] = cur; }
                c->steps_since_sort += m; c->total_steps += m;
                done += m;
            }
            return TH_OK;
        }
    }

    // everything below runs in texel order
    if (th_status s = ensure_identity(c)) return s;
    static const bool graphs_on = [] { const char *e = getenv("TH_GRAPH"); return !e || atoi(e) != 0; }();
    if (!graphs_on || n < 2 || (c->packed && plan.generic)) {
        for (int32_t k = 0; k < n; ++k) {
            if (k) { v.time = times[(size_t)k]; if (th_status s = plan_step(c, v, TH_TARGET_RING, plan)) return s; }
            if (th_status s = enqueue_step(c, plan, TH_TARGET_RING, times[(size_t)k], nullptr, true)) return s;
        }
        return TH_OK;
    }

    // cache lookup: same n, same parameters (time excluded), same ring order and layout
    th::LogicParams key = plan.p;
    key.u.time = 0.0f;
    GraphEntry *hit = nullptr;
    for (GraphEntry &g : c->graphs)
        if (g.n == n && g.mode == c->cfg.mode && g.ring == c->ring && g.flags == plan_flags(plan) && same_key(g.key, key)) { hit = &g; break; }
    if (!hit) {
        if (c->graphs.size() >= 8) { destroy_graph(c->graphs.front()); c->graphs.erase(c->graphs.begin()); }
        GraphEntry g;
        g.n = n; g.mode = c->cfg.mode; g.ring = c->ring;
        g.flags = plan_flags(plan); g.key = key;
        TH_HIP(hipMalloc((void **)&g.times_dev, (size_t)n * sizeof(float)));
        TH_HIP(hipHostMalloc((void **)&g.times_host, (size_t)n * sizeof(float)));
        TH_HIP(hipEventCreate(&g.copied));
        const std::vector<float4 *> ring_before = c->ring;
        const int since_before = c->steps_since_sort;
        const long long total_before = c->total_steps;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        th_status st = TH_OK;
        if (e == hipSuccess) {
            for (int32_t k = 0; k < n && st == TH_OK; ++k)
                st = enqueue_step(c, plan, TH_TARGET_RING, 0.0f, g.times_dev + k, false);
            hipGraph_t graph = nullptr;
            e = hipStreamEndCapture(c->stream, &graph);
            if (e == hipSuccess && st == TH_OK) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
        }
        c->ring = ring_before;                         // the capture only recorded; nothing ran yet
        c->steps_since_sort = since_before;
        c->total_steps = total_before;
        if (e != hipSuccess || st != TH_OK) {
            destroy_graph(g);
            if (st != TH_OK) return st;
            return fail(TH_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e));
        }
        c->graphs.push_back(g);
        hit = &c->graphs.back();
    }
    TH_HIP(hipEventSynchronize(hit->copied));          // previous replay's copy out of times_host is done
    memcpy(hit->times_host, times.data(), (size_t)n * sizeof(float));
    TH_HIP(hipMemcpyAsync(hit->times_dev, hit->times_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    TH_HIP(hipGraphLaunch(hit->exec, c->stream));
    for (int32_t k = 0; k < n; ++k) {                  // host-side ring bookkeeping of the n rotations
        float4 *last = c->ring.back();
        c->ring.pop_back();
        c->ring.insert(c->ring.begin(), last);
    }
    c->steps_since_sort += n; c->total_steps += n;
    // times_host must stay untouched until the copy has run; a later replay of this entry waits here
    TH_HIP(hipEventRecord(hit->copied, c->stream));
    return TH_OK;
}

th_status th_spawn_init(th_context *c, int32_t target)
{
    if (th_status s = use(c)) return s;
    if (th_status s = ensure_identity(c)) return s;      // these operate in texel order
    float4 *out = nullptr;
    if (target == TH_TARGET_RING) TH_REQUIRE(!c->ring.empty(), "no state buffers");
    if (th_status s = resolve_target(c, target, true, &out)) return s;
    // src/spawn/init/index.frag:5-10
    float4 *rt = nullptr;
    if (th_status s = render_target(c, out, 0, &rt)) return s;
    th::launch_fill(rt, make_float4(th::kInert, th::kInert, 0.0f, 0.0f), c->texels(), c->stream);
    TH_HIP(hipGetLastError());
    return commit_target(c, out, rt);
}

th_status th_spawn_ball(th_context *c, const th_spawn_ball_uniforms *u, int32_t target)
{
    if (th_status s = use(c)) return s;
    if (th_status s = ensure_identity(c)) return s;      // these operate in texel order
    TH_REQUIRE(u, "null uniforms");
    if (target == TH_TARGET_RING) TH_REQUIRE(!c->ring.empty(), "no state buffers");
    float4 *out = nullptr;
    if (th_status s = resolve_target(c, target, true, &out)) return s;
    float4 *rt = nullptr;
    if (th_status s = render_target(c, out, 0, &rt)) return s;
    th::SpawnBallParams p{};
    p.out = rt; p.count = (uint32_t)c->texels(); p.width = (uint32_t)c->cfg.width; p.row0 = (uint32_t)c->cfg.row0;
    p.u = *u;
    th::launch_spawn_ball(p, c->stream);
    if (target != TH_TARGET_TARGETS) th::launch_counter_add(c->d_respawned, c->texels(), c->stream);
    TH_HIP(hipGetLastError());
    return commit_target(c, out, rt);
}

static th_status spawn_from_data(th_context *c, const th_spawn_sample_uniforms *u, int32_t source, int32_t target, bool direct)
{
    if (th_status s = use(c)) return s;
    if (th_status s = ensure_identity(c)) return s;      // these operate in texel order
    TH_REQUIRE(u, "null uniforms");
    if (!direct) TH_REQUIRE(u->samples >= 0 && u->samples <= 64, "samples out of range");
    TH_REQUIRE(direct || (u->apply >= 0 && u->apply <= 3), "unknown apply mode %d", u->apply);
    // the pass reads `particles` = buffers[1] like every Particles.step (src/particles.js:139)
    TH_REQUIRE(c->ring.size() >= 2, "spawn pass needs at least 2 state buffers (have %zu)", c->ring.size());
    float4 *out = nullptr;
    if (th_status s = resolve_target(c, target, true, &out)) return s;
    float4 *rt = nullptr, *particles = nullptr;
    if (th_status s = render_target(c, out, 0, &rt)) return s;
    if (th_status s = unpacked_view(c, c->ring[1], 1, &particles)) return s;
    th::SpawnSampleParams p{};
    p.particles = particles;
    p.out = rt;
    // `source` names the spawnData texture in the ring order the pass sees (after the rotation)
    if (source == TH_SOURCE_FLOW) { p.data = c->flow; p.dw = c->fw; p.dh = c->fh; }
    else if (source == TH_SOURCE_IMAGE) {
        TH_REQUIRE(c->image, "no spawn image (call th_spawn_image_upload)");
        p.data = c->image; p.dw = c->iw; p.dh = c->ih;
    } else if (source >= 0 && source < (int32_t)c->ring.size()) {
        float4 *data = nullptr;
        if (c->cfg.height != c->cfg.global_height) {
            // a row-band shard: the pass samples ARBITRARY particles (src/demo.main.js:433-441) - from the copy of the whole
            // texture the ranks gathered beforehand (th_state_gather, or a host's own transport through th_state_gather_ptr)
            TH_REQUIRE(c->gathered && c->gathered_of == (const void *)c->ring[(size_t)source],
                       "sampling the particle texture on a row-band shard (%d of %d rows) reads every band: gather buffer %d first (th_state_gather / th_state_gather_ptr)",
                       c->cfg.height, c->cfg.global_height, source);
            data = c->gathered;
        } else if (source == 1) data = particles;
        else if (th_status s = unpacked_view(c, c->ring[source], 2, &data)) return s;
        p.data = data; p.dw = c->cfg.width; p.dh = c->cfg.global_height;
    } else return fail(TH_ERR_INVALID, "bad spawnData source %d", source);
    p.count = (uint32_t)c->texels(); p.width = (uint32_t)c->cfg.width; p.row0 = (uint32_t)c->cfg.row0;
    p.wf = (float)c->cfg.width; p.hf = (float)c->cfg.global_height;
    p.u = *u;
    p.accepted = c->d_respawned + (target == TH_TARGET_TARGETS ? 1 : 0);
    if (direct) th::launch_spawn_direct(p, c->stream); else th::launch_spawn_sample(p, c->stream);
    TH_HIP(hipGetLastError());
    return commit_target(c, out, rt);
}

th_status th_spawn_sample(th_context *c, const th_spawn_sample_uniforms *u, int32_t source, int32_t target)
{
    return spawn_from_data(c, u, source, target, false);
}

th_status th_spawn_direct(th_context *c, const th_spawn_sample_uniforms *u, int32_t source, int32_t target)
{
    return spawn_from_data(c, u, source, target, true);
}

static th_status image_resize(th_context *c, int32_t w, int32_t h);

th_status th_spawn_image_upload(th_context *c, const float *rgba, int32_t w, int32_t h)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(rgba, "null pixels");
    if (th_status s = image_resize(c, w, h)) return s;
    TH_HIP(hipMemcpyAsync(c->image, rgba, (size_t)w * h * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

static th_status image_resize(th_context *c, int32_t w, int32_t h)
{
    TH_REQUIRE(w > 0 && h > 0 && w < (1 << 24) && h < (1 << 24) && (uint64_t)w * h < (1ull << 28), "bad image %dx%d", w, h);
    if (w != c->iw || h != c->ih) {
        TH_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->image);
        c->image = nullptr; c->iw = c->ih = 0;
        TH_HIP(hipMalloc((void **)&c->image, (size_t)w * h * sizeof(float4)));
        c->iw = w; c->ih = h;
    }
    return TH_OK;
}

th_status th_spawn_image_triangles(th_context *c, const float *positions, int32_t triangles, const float viewSize[2],
                                   const float color[4], int32_t w, int32_t h)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(triangles >= 0 && triangles <= (1 << 20) && (positions || triangles == 0) && viewSize && color, "bad arguments");
    if (th_status s = image_resize(c, w, h)) return s;
    TH_HIP(hipMemsetAsync(c->image, 0, (size_t)w * h * sizeof(float4), c->stream));      // gl.clear(COLOR_BUFFER_BIT)
    if (triangles == 0) return TH_OK;
    float *d_pos = nullptr;
    th::TrianglePoly *d_polys = nullptr;
    TH_HIP(hipMalloc((void **)&d_pos, (size_t)triangles * 6 * sizeof(float)));
    TH_HIP(hipMalloc((void **)&d_polys, (size_t)triangles * sizeof(th::TrianglePoly)));
    TH_HIP(hipMemcpyAsync(d_pos, positions, (size_t)triangles * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    th::launch_triangles(d_pos, triangles, viewSize[0], viewSize[1], make_float4(color[0], color[1], color[2], color[3]),
                         d_polys, c->image, w, h, c->stream);
    hipError_t e = hipGetLastError();
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_pos); (void)hipFree(d_polys);
    TH_HIP(e);
    return TH_OK;
}

th_status th_spawn_image_download(th_context *c, float *rgba)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(rgba && c->image, "no spawn image");
    TH_HIP(hipMemcpyAsync(rgba, c->image, (size_t)c->iw * c->ih * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

// ---- flow deposit ------------------------------------------------------------------------------------------
static int deposit_texel_bits(const th_context *c)
{
    const uint64_t texels = (uint64_t)c->fw * c->fh;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < texels) ++bits;
    return bits;
}

// ---- which pipeline draws -------------------------------------------------------------------------------------
// The binned pipeline (th_bins.hip) walks the particles by slot, in whatever order the ring is held; the stream-ordered one
// (th_deposit.hip) needs texel order.  TH_DRAW=bins / stream forces one (tests); by default the binned pipeline draws
// whenever the integrator would step over tile-sorted slots (sorting_possible): a step() + draw() frame loop then never
// leaves the sorted order.
static int draw_policy()
{
    static const int v = [] { const char *e = getenv("TH_DRAW"); return !e ? -1 : (!strcmp(e, "bins") ? 1 : (!strcmp(e, "stream") ? 0 : -1)); }();
    return v;
}

// Which pipeline a draw pass takes.  auto: wherever the integrator steps over tile-sorted slots the frame loop - step(); draw() -
// stays on them: the binned pipeline takes particles in any order.  It is ahead while the target is not crowded (first ~60
// frames at C3: 1.7 against 2.3 ms per draw with both passes) and level with the stream-ordered pipeline once the wake has made
// the particles converge (70-76 % of all fragments in bins of more than 4096, in texels with hundreds and thousands of them:
// 2.0-2.7 ms either way; profiles/r3_b_fused_pass_experiments.txt) - restoring GL's order per texel then means sorting most
// fragments by stream index, an order the stream-ordered pipeline gets for free from walking particles in texel order.  Beyond
// that, auto hands over: when more than kCrowdedShare of a binned pass's fragments fell into large bins three passes in a row,
// the next kStreamSpell passes (doubling, up to 4096, while it stays so) go to the stream-ordered pipeline, then the binned one
// is tried again.
constexpr double kCrowdedShare = 0.8;
constexpr int kStreamSpell = 256;

static float drawn_line_width(const th_context *c, int pass)
{
    const float w = c->line_width[pass];
    return w < c->line_range[0] ? c->line_range[0] : (w > c->line_range[1] ? c->line_range[1] : w);
}

static bool draw_uses_bins(th_context *c)
{
    const int policy = c->draw_pipeline != TH_DRAW_AUTO ? c->draw_pipeline : draw_policy();
    if (policy == 0) return false;
    if (c->cfg.height != c->cfg.global_height || c->fw > th::kBinsMaxExtent || c->fh > th::kBinsMaxExtent) return false;
    if (policy == 1) return true;
    if (c->draws < c->stream_until) return false;
    // (lines wider than 2 cover more texels than a line's record holds: nearly all of them would leave the fused pass
    // for the long list, one atomic per fragment - the stream-ordered pipeline counts and scans instead)
    if (drawn_line_width(c, TH_PASS_FLOW) > 2.0f || drawn_line_width(c, TH_PASS_VIEW) > 2.0f) return false;
    return sorting_possible(c);
}

// ring[1] into ring[0]'s slot order (through texel order): only when a draw meets the two in different orders - a
// re-sorting step moves its input along with its output while draws are going on (enqueue_step)
static th_status align_slot_orders(th_context *c)
{
    const int o0 = order_of(c, c->ring[0]), o1 = order_of(c, c->ring[1]);
    if (o0 == o1) return TH_OK;
    if (th_status s = sort_storage(c)) return s;
    clear_graphs(c);
    float4 *&b = c->ring[1];
    if (o1 >= 0) {
        th::launch_unpermute_state(c->spare, b, c->orders[(size_t)o1].perm, (uint32_t)c->texels(), c->packed, c->stream);
        set_order(c, b, -1);
        float4 *t = b; b = c->spare; c->spare = t;
    }
    if (o0 >= 0) {
        th::launch_permute_state(c->spare, b, c->orders[(size_t)o0].perm, (uint32_t)c->texels(), c->packed, c->stream);
        float4 *t = b; b = c->spare; c->spare = t;
        set_order(c, b, o0);
    }
    TH_HIP(hipGetLastError());
    c->counted.buf = nullptr;
    return TH_OK;
}

// per-line buffers + parameters.  want_bins: the caller can run the binned pipeline (*bins tells whether it will)
static th_status deposit_prepare(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, bool want_bins = false, bool *bins = nullptr)
{
    TH_REQUIRE(u, "null uniforms");
    TH_REQUIRE(c->ring.size() >= 2, "draw needs at least 2 state buffers (have %zu)", c->ring.size());
    if (want_bins) ++c->draws;
    bool use_bins = want_bins && draw_uses_bins(c);
    if (use_bins) {
        // the binned pipeline reads every vertex of a line from the line's own slot: shapes whose vertex lookup lands on
        // another particle (line_rows) keep to the stream-ordered pipeline in texel order
        if (th_status s = line_rows(c)) return s;
        if (c->lines_local != 1) use_bins = false;
        else if (any_sorted(c)) { if (th_status s = align_slot_orders(c)) return s; }
    }
    if (bins) *bins = use_bins;
    if (use_bins) c->last_binned_draw = c->total_steps;
    else {
        if (th_status s = ensure_identity(c)) return s;      // the vertex stream addresses particles in texel order
        c->hold_texel_order_until = c->total_steps + rebucket_period();   // a frame loop of step + draw stays in texel order
    }
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
        if (!c->dep_total) TH_HIP(hipMalloc((void **)&c->dep_total, 8 * sizeof(uint32_t)));      // [0] total, [1] out-of-band flag, [2] largest bin, [3] large bins, [4] their blocks
        c->dep_lines = lines;
    }
    p = th::DepositParams{};
    {   // a packed ring is read through f32 views (what the stored texels decode to)
        float4 *cur = nullptr, *prev = nullptr;
        if (th_status s = unpacked_view(c, c->ring[0], 0, &cur)) return s;
        if (th_status s = unpacked_view(c, c->ring[1], 1, &prev)) return s;
        p.cur = cur; p.prev = prev;
    }
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
        p.lists = c->dep_lists + (th::deposit_list_words((uint32_t)c->cfg.width, (uint32_t)c->cfg.height, &cap) - (size_t)2 * 64 * cap);
    }
    p.halo_lo = c->halo_lo; p.halo_hi = c->halo_hi;
    TH_HIP(hipMemsetAsync(c->dep_total, 0, 8 * sizeof(uint32_t), c->stream));
    if (th_status s = line_rows(c)) return s;
    p.row_draws = c->d_row_draws;
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
            const size_t table = (size_t)p.nbins * th::kBinReplicas * th::kBinMaxPages * sizeof(uint32_t);
            TH_HIP(hipMalloc((void **)&c->chunk_table, table));
            TH_HIP(hipMemsetAsync(c->chunk_table, 0, table, c->stream));        // (every reader of a list leaves its entries empty)
            c->bin_capacity = p.nbins;
        }
        p.bin_stride = (uint32_t)(((size_t)c->bin_capacity + 255) / 256 * 256 + 64);
        p.bin_cursor = c->bin_mem; p.large_bins = c->bin_mem + (size_t)th::kBinReplicas * p.bin_stride;
        p.large_key0 = p.large_bins + c->bin_capacity;
        p.page_table = c->chunk_table;
        p.totals = c->dep_total;
    }
    return TH_OK;
}

// a few words from the device to the host, through pinned memory, with the stream's work before them finished
constexpr size_t kPinnedBytes = 1024;
static th_status read_back(th_context *c, void *host, const void *dev, size_t bytes)
{
    TH_REQUIRE(bytes <= kPinnedBytes, "read_back of %zu bytes", bytes);
    if (!c->pinned) TH_HIP(hipHostMalloc(&c->pinned, kPinnedBytes, hipHostMallocDefault));
    TH_HIP(hipMemcpyAsync(c->pinned, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    memcpy(host, c->pinned, bytes);
    return TH_OK;
}

// scan of p.count (filled by the caller's marking pass) -> p.offset, total (one sync); reports band violations
static th_status deposit_scan_total(th_context *c, const th::DepositParams &p, uint32_t *total)
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
static th_status deposit_count(th_context *c, const th_deposit_uniforms *u, th::DepositParams &p, uint32_t *total)
{
    if (th_status s = deposit_prepare(c, u, p)) return s;
    th::launch_deposit_count(p, c->stream);
    return deposit_scan_total(c, p, total);
}

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

th_status th_export_lines(th_context *c, const th_deposit_uniforms *u, float *lines, uint64_t capacity, uint64_t *count)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(count, "null count");
    th::DepositParams p;
    if (th_status s = deposit_prepare(c, u, p)) return s;
    return export_run(c, p, lines, capacity, count);
}

// per-fragment buffers for `total` fragments (grow-only)
static th_status deposit_reserve(th_context *c, uint32_t total, bool wide, bool pairs = false)
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

static th_status deposit_temp(th_context *c, size_t need)
{
    if (c->dep_temp_bytes < need) {
        (void)hipFree(c->dep_temp); c->dep_temp = nullptr; c->dep_temp_bytes = 0;
        TH_HIP(hipMalloc(&c->dep_temp, need + need / 4));
        c->dep_temp_bytes = need + need / 4;
    }
    return TH_OK;
}

// the fragments of the (prepared) pass `p`: count, emit, sort by texel, blend
static th_status deposit_run(th_context *c, th::DepositParams &p, uint64_t *fragments)
{
    // Same state, same view, same resolution as the pass before (the view pass after the flow pass of one draw()): the
    // lines cover the same texels in the same order - counts, offsets, records and the sorted order of the fragments are
    // still there, only the varyings differ.  (TH_DRAW_REUSE=0: every pass on its own.)
    static const bool reuse_allowed = [] { const char *e = getenv("TH_DRAW_REUSE"); return !e || atoi(e) != 0; }();
    const bool reuse = reuse_allowed && p.mode != 2 && c->drawn.valid && !c->drawn.binned && c->drawn.view_x == p.view_x && c->drawn.view_y == p.view_y &&
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
static th_status bins_store(th_context *c, uint32_t nbins, uint32_t pool, bool pairs)
{
    if (c->bins_keys && c->bins_store_bins == nbins && c->bins_pool >= pool && (c->bins_pairs || !pairs)) return TH_OK;
    TH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->bins_keys); (void)hipFree(c->bins_colors);
    c->bins_keys = nullptr; c->bins_colors = nullptr;
    pool = pool > c->bins_pool ? pool : c->bins_pool;
    pairs = true;           // (room for both varyings of a th_draw from the start: growing the store later costs a frame)
    c->bins_pool = 0; c->bins_store_bins = 0;
    const size_t places = ((size_t)nbins * th::kBinReplicas + pool) * th::kBinPage;
    TH_REQUIRE(places < ((size_t)1 << 32), "the binned draw's chunk store would hold 2^32 places or more");
    TH_HIP(hipMalloc((void **)&c->bins_keys, places * sizeof(unsigned long long)));
    TH_HIP(hipMalloc((void **)&c->bins_colors, places * (pairs ? 2 : 1) * sizeof(float4)));
    TH_HIP(hipMemsetAsync(c->bins_keys, 0xff, places * sizeof(unsigned long long), c->stream));
    c->bins_pool = pool; c->bins_store_bins = nbins; c->bins_pairs = pairs;
    return TH_OK;
}

constexpr double kEarlyBlendShare = 0.5;            // (of a draw's fragments in crowded bins: see deposit_run_bins)
constexpr th_status kRetryInStreamOrder = -1;        // (internal) the binned pass gave up before it touched a target

// the binned pipeline (th_bins.hip) over the (prepared) pass `p`: rasterise + emit into the bins, plan, per-bin order + blend
static th_status deposit_run_bins(th_context *c, th::DepositParams &p, uint64_t *fragments)
{
    c->drawn.valid = false;
    if (!c->side) {
        TH_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        TH_HIP(hipEventCreateWithFlags(&c->forked, hipEventDisableTiming));
        TH_HIP(hipEventCreateWithFlags(&c->joined, hipEventDisableTiming));
        TH_HIP(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
        TH_HIP(hipEventCreateWithFlags(&c->joined2, hipEventDisableTiming));
        TH_HIP(hipEventCreateWithFlags(&c->regrouped, hipEventDisableTiming));
        TH_HIP(hipHostMalloc((void **)&c->bins_totals_host, th::kTotWords * sizeof(uint32_t), hipHostMallocDefault));
    }
    uint32_t *host = c->bins_totals_host;
    // Which comes first behind the emitting pass: the ordinary bins' blend - it needs nothing from the host and covers the
    // read-back - or, on a crowded target, the crowded bins' kernels: their long runs (walked by one thread each, on the
    // side stream) are then the longest chain of the draw and must start as early as they can.  Decided by the last draw.
    const bool early = !(c->last_draw.pipeline == TH_DRAW_BINS && (double)c->last_draw.crowded_fragments > kEarlyBlendShare * (double)c->last_draw.fragments);
    for (int attempt = 0;; ++attempt) {
        // (TH_BINS_POOL: the first pool's size in pages - tests make it small to run the growth path)
        static const uint32_t pool0 = [] { const char *e = getenv("TH_BINS_POOL"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u; }();
        const uint32_t pool = c->bins_pool ? c->bins_pool : (pool0 ? pool0 : (p.nbins * 16u > 16384u ? p.nbins * 16u : 16384u));
        if (th_status s = bins_store(c, p.nbins, pool, p.mode == 2)) return s;
        p.frag_keys = c->bins_keys; p.colors = c->bins_colors; p.pool_pages = c->bins_pool;
        if (attempt) TH_HIP(hipMemsetAsync(c->dep_total, 0, th::kTotWords * sizeof(uint32_t), c->stream));
        th::launch_bins_fused(p, c->stream);
        // the totals come back over the side stream while the ordinary bins are already being blended (the kernel looks at the
        // pass's flags itself): the host's round trip - it sizes the crowded bins' launches - costs the GPU nothing
        TH_HIP(hipEventRecord(c->forked, c->stream));
        if (early) th::launch_bins_blend(p, c->stream);
        TH_HIP(hipStreamWaitEvent(c->side, c->forked, 0));
        TH_HIP(hipMemcpyAsync(host, c->dep_total, th::kTotWords * sizeof(uint32_t), hipMemcpyDeviceToHost, c->side));
        TH_HIP(hipStreamSynchronize(c->side));
        const uint32_t flags = host[th::kTotFlags];
        if (flags == 0) break;
        // nothing has been blended yet: the store is wiped, and the pass is repeated with a larger pool - or, when a bin
        // outgrew its chunk table or a line its reservation, left to the stream-ordered pipeline
        TH_HIP(hipMemsetAsync(c->bins_keys, 0xff, ((size_t)c->bins_store_bins * th::kBinReplicas + c->bins_pool) * th::kBinPage * sizeof(unsigned long long), c->stream));
        TH_HIP(hipMemsetAsync(c->chunk_table, 0, (size_t)c->bin_capacity * th::kBinReplicas * th::kBinMaxPages * sizeof(uint32_t), c->stream));
        if ((flags & ~th::kBinsPoolExhausted) || attempt >= 2) return kRetryInStreamOrder;
        const uint32_t want = 2u * host[th::kTotPool] + 64;      // (generously: growing the store costs a frame's worth of time)
        if (th_status s = bins_store(c, p.nbins, want, p.mode == 2)) return s;
    }
    const uint32_t total = host[th::kTotFragments], nlarge = host[th::kTotLarge];
    if (fragments) *fragments = total;
    c->last_draw.pipeline = TH_DRAW_BINS; c->last_draw.fragments = total; c->last_draw.crowded_fragments = host[th::kTotCrowdKeys];
    {   // (auto policy: see draw_uses_bins)
        const bool crowded = total > 0 && (double)host[th::kTotCrowdKeys] > kCrowdedShare * (double)total;
        c->crowded_streak = crowded ? c->crowded_streak + 1 : 0;
        if (!crowded) c->stream_spell = 0;
        if (c->crowded_streak >= 3 || (crowded && c->stream_spell > 0)) {
            c->stream_spell = c->stream_spell ? std::min(2 * c->stream_spell, 4096) : kStreamSpell;
            c->stream_until = c->draws + c->stream_spell;
            c->crowded_streak = 0;
        }
    }
    if (host[th::kTotCrowdKeys] == 0xffffffffu) return fail(TH_ERR_UNSUPPORTED, "too many fragments in crowded bins for one draw (2^32 or more places)");
    if (c->crowd_capacity < nlarge) {
        (void)hipFree(c->crowd_mem); c->crowd_mem = nullptr; c->crowd_capacity = 0;
        const uint32_t cap = 2u * nlarge + 256;
        TH_HIP(hipMalloc((void **)&c->crowd_mem, (size_t)cap * th::crowd_words_per_bin() * sizeof(uint32_t)));
        c->crowd_capacity = cap;
    }
    if (c->crowd_keys_cap < host[th::kTotCrowdKeys]) {
        (void)hipFree(c->crowd_keys); (void)hipFree(c->crowd_sorted); c->crowd_keys = nullptr; c->crowd_sorted = nullptr; c->crowd_keys_cap = 0;
        const size_t cap = 2 * (size_t)host[th::kTotCrowdKeys] + ((size_t)1 << 20);
        TH_HIP(hipMalloc((void **)&c->crowd_keys, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->crowd_sorted, cap * sizeof(uint32_t)));
        c->crowd_keys_cap = cap;
    }
    p.nlarge = nlarge;
    p.crowd_count = c->crowd_mem; p.crowd_cursor = c->crowd_mem + (size_t)c->crowd_capacity * 256; p.crowd_start = p.crowd_cursor + (size_t)c->crowd_capacity * 256;
    p.crowd_long = p.crowd_start + (size_t)c->crowd_capacity * 257; p.crowd_giant = p.crowd_long + (size_t)c->crowd_capacity * 256;
    p.crowd_keys = c->crowd_keys; p.crowd_sorted = c->crowd_sorted;
    if (nlarge) {
        // The crowded bins on two streams of their own, beside the ordinary bins' blend (disjoint texels, kernels that wait on
        // chains and loads rather than fill the chip): their fragments regrouped by texel, then the long runs on one stream -
        // the walk of the longest run, one fragment after the other, overlaps with everything else instead of following it -
        // and the short runs on the other.
        TH_HIP(hipStreamWaitEvent(c->side2, c->forked, 0));       // (recorded behind the emitting pass and its plan: the ordinary bins' blend need not be waited for)
        th::launch_bins_regroup(p, c->side2);
        TH_HIP(hipEventRecord(c->regrouped, c->side2));
        TH_HIP(hipStreamWaitEvent(c->side, c->regrouped, 0));
        th::launch_bins_blend_long(p, c->side);
        TH_HIP(hipEventRecord(c->joined, c->side));
        th::launch_bins_blend_crowd(p, c->side2);
        TH_HIP(hipEventRecord(c->joined2, c->side2));
    }
    if (!early) th::launch_bins_blend(p, c->stream);
    if (nlarge) { TH_HIP(hipStreamWaitEvent(c->stream, c->joined, 0)); TH_HIP(hipStreamWaitEvent(c->stream, c->joined2, 0)); }
    TH_HIP(hipGetLastError());
    return TH_OK;
}

th_status th_flow_deposit(th_context *c, const th_deposit_uniforms *u, uint64_t *fragments)
{
    if (th_status s = use(c)) return s;
    if (c->cfg.height != c->cfg.global_height)
        return fail(TH_ERR_UNSUPPORTED, "flow deposit on a row-band shard (%d of %d rows): use th_deposit_emit / th_deposit_merge with the exchange of tendrils_amd/sharding.py", c->cfg.height, c->cfg.global_height);
    for (int pass = 0;; ++pass) {            // (a binned pass that gives up before blending is repeated in stream order)
        th::DepositParams p;
        bool bins = false;
        if (th_status s = deposit_prepare(c, u, p, pass == 0, &bins)) return s;
        if (!bins) return deposit_run(c, p, fragments);
        const th_status s = deposit_run_bins(c, p, fragments);
        if (s != kRetryInStreamOrder) return s;
    }
}

// ---- view pass ---------------------------------------------------------------------------------------------
static th_status view_storage(th_context *c)
{
    if (c->view && c->view_w == c->fw && c->view_h == c->fh) return TH_OK;
    TH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->view);
    c->view = nullptr; c->view_w = c->view_h = 0;
    TH_HIP(hipMalloc((void **)&c->view, (size_t)c->fw * c->fh * sizeof(uchar4)));
    TH_HIP(hipMemsetAsync(c->view, 0, (size_t)c->fw * c->fh * sizeof(uchar4), c->stream));     // a fresh drawing buffer is transparent black
    c->view_w = c->fw; c->view_h = c->fh;
    return TH_OK;
}

static void view_fields(th_context *c, const th_render_uniforms *u, th::DepositParams &p)
{
    p.flow_decay = u->flowDecay; p.speed_alpha = u->speedAlpha; p.colormap_alpha = u->colorMapAlpha; p.sin_term = u->sinTerm;
    for (int k = 0; k < 4; ++k) { p.base_color[k] = u->baseColor[k]; p.flow_color[k] = u->flowColor[k]; }
    p.colormap = c->colormap; p.cw = c->cmap_w; p.ch = c->cmap_h;
}

static th_status view_params(th_context *c, const th_render_uniforms *u, th::DepositParams &p, bool want_bins = false, bool *bins = nullptr)
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

// Both passes of Tendrils.draw() (src/index.js:278-337) in one: the lines are rasterised, scanned, emitted and sorted once,
// every fragment carries the flow pass's varying and the view pass's colour side by side, one gather brings both into
// the sorted order and each target is blended from its half.  The two passes must agree on what they draw: the same
// viewSize, time and speedLimit (what Tendrils.draw() hands to both); results are those of th_flow_deposit followed
// by th_view_draw.
th_status th_draw(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *fragments)
{
    if (th_status s = use(c)) return s;
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
    }
}

th_status th_view_draw(th_context *c, const th_render_uniforms *u, uint64_t *fragments)
{
    if (th_status s = use(c, true)) return s;
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

th_status th_deposit_set_owners(th_context *c, int32_t world)
{
    TH_REQUIRE(c, "null context");
    // (the owner's merge walks up to 32 source bands per texel - th_deposit.hip: kMaxBands: more ranks than that could only be
    // refused after the blend had begun)
    TH_REQUIRE(world >= 1 && world <= 32, "owner count %d outside [1, 32]", world);
    c->dep_owners = (uint32_t)world;
    return TH_OK;
}

// the (counted) fragments of this band's lines, keyed (owner, texel, global stream index) and parted by owner
static th_status emit_parted(th_context *c, th::DepositParams &p, uint32_t total, uint64_t *count, void **keys_dev, void **colors_dev)
{
    *count = total; *keys_dev = nullptr; *colors_dev = nullptr;
    if (total == 0) return TH_OK;
    const bool pairs = p.mode == 2;                  // th_draw_emit: two varyings per fragment, side by side
    if (th_status s = deposit_reserve(c, total, true, pairs)) return s;
    p.keys64 = c->dep_u64[0]; p.slots = c->dep_u32[1]; p.colors = c->dep_colors;
    p.owners = c->dep_owners;
    p.owner_chunk = (uint32_t)(((uint64_t)c->fw * c->fh + p.owners - 1u) / p.owners);
    th::launch_deposit_scatter(p, c->stream);
    *keys_dev = c->dep_u64[0]; *colors_dev = c->dep_colors;
    if (p.owners > 1u) {
        // the fragment array is in this band's stream order: ONE stable pass on the owner bits parts it by destination
        // (every part still in stream order); the owners sort by texel
        int owner_bits = 1;
        while ((1u << owner_bits) < p.owners) ++owner_bits;
        if (th_status s = deposit_temp(c, th::radix_sort_temp_bytes(total, th::kOwnerShift, th::kOwnerShift + owner_bits))) return s;
        const int in_b = th::launch_radix_sort_u64(c->dep_u64[0], c->dep_u32[1], c->dep_u64[1], c->dep_u32[3], total, th::kOwnerShift,
                                                   th::kOwnerShift + owner_bits, c->dep_temp, true, c->stream);
        if (pairs) th::launch_deposit_gather_pairs(c->dep_colors_sorted, c->dep_colors, in_b ? c->dep_u32[3] : c->dep_u32[1], total, c->stream);
        else th::launch_deposit_gather_colors(c->dep_colors_sorted, c->dep_colors, in_b ? c->dep_u32[3] : c->dep_u32[1], total, c->stream);
        *keys_dev = in_b ? c->dep_u64[1] : c->dep_u64[0]; *colors_dev = c->dep_colors_sorted;
    }
    TH_HIP(hipGetLastError());
    TH_HIP(hipStreamSynchronize(c->stream));               // the caller hands the buffers to a collective on its own stream
    return TH_OK;
}

th_status th_deposit_emit(th_context *c, const th_deposit_uniforms *u, uint64_t *count, void **keys_dev, void **colors_dev)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(count && keys_dev && colors_dev, "null outputs");
    TH_REQUIRE((uint64_t)c->fw * c->fh <= (uint64_t)th::kTexelMask + 1u, "the sharded deposit keys hold 24 texel bits: flow %dx%d is too large", c->fw, c->fh);
    th::DepositParams p;
    uint32_t total = 0;
    if (th_status s = deposit_count(c, u, p, &total)) return s;
    return emit_parted(c, p, total, count, keys_dev, colors_dev);
}

// the view pass of a row-band shard: the same lines with the render shader's colours
th_status th_view_emit(th_context *c, const th_render_uniforms *u, uint64_t *count, void **keys_dev, void **colors_dev)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(count && keys_dev && colors_dev, "null outputs");
    TH_REQUIRE((uint64_t)c->fw * c->fh <= (uint64_t)th::kTexelMask + 1u, "the sharded deposit keys hold 24 texel bits: view %dx%d is too large", c->fw, c->fh);
    th::DepositParams p;
    if (th_status s = view_params(c, u, p)) return s;
    th::launch_deposit_count(p, c->stream);
    uint32_t total = 0;
    if (th_status s = deposit_scan_total(c, p, &total)) return s;
    return emit_parted(c, p, total, count, keys_dev, colors_dev);
}

// both passes of a row-band shard's draw() in one: every fragment with the flow pass's varying and the view pass's colour
// side by side (32 bytes), rasterised, parted and - by the host or th_draw_sharded - exchanged once
th_status th_draw_emit(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *count, void **keys_dev, void **colors_dev)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(du && ru && count && keys_dev && colors_dev, "null argument");
    TH_REQUIRE((uint64_t)c->fw * c->fh <= (uint64_t)th::kTexelMask + 1u, "the sharded deposit keys hold 24 texel bits: target %dx%d is too large", c->fw, c->fh);
    TH_REQUIRE(memcmp(du->viewSize, ru->viewSize, sizeof du->viewSize) == 0 && memcmp(&du->time, &ru->time, sizeof du->time) == 0 &&
               memcmp(&du->speedLimit, &ru->speedLimit, sizeof du->speedLimit) == 0,
               "the two passes of one draw share viewSize, time and speedLimit");
    TH_REQUIRE(drawn_line_width(c, TH_PASS_FLOW) == drawn_line_width(c, TH_PASS_VIEW),
               "the two passes draw their lines %g and %g wide: th_deposit_emit and th_view_emit rasterise them apart",
               (double)drawn_line_width(c, TH_PASS_FLOW), (double)drawn_line_width(c, TH_PASS_VIEW));
    th::DepositParams p;
    if (th_status s = deposit_prepare(c, du, p)) return s;
    p.mode = 2;
    view_fields(c, ru, p);
    th::launch_deposit_count(p, c->stream);
    uint32_t total = 0;
    if (th_status s = deposit_scan_total(c, p, &total)) return s;
    return emit_parted(c, p, total, count, keys_dev, colors_dev);
}

th_status th_deposit_set_halo(th_context *c, const void *lo_dev, const void *hi_dev)
{
    TH_REQUIRE(c, "null context");
    c->halo_lo = static_cast<const float4 *>(lo_dev);
    c->halo_hi = static_cast<const float4 *>(hi_dev);
    return TH_OK;
}

// target: 0 = the flow texture, 1 = the view buffer, 2 = both (the fragments carry pairs of varyings: th_draw_emit)
static th_status merge_parted(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count, int target)
{
    if (count == 0) return TH_OK;
    TH_REQUIRE(keys_dev && colors_dev && count < (1ull << 31), "bad fragment buffers");
    const uint32_t total = (uint32_t)count;
    const bool into_view = target == 1;
    if (target == 2 && !c->mrg_pairs) {               // room for two varyings per fragment from now on
        (void)hipFree(c->mrg_colors); c->mrg_colors = nullptr;
        if (c->mrg_capacity) TH_HIP(hipMalloc((void **)&c->mrg_colors, 2 * c->mrg_capacity * sizeof(float4)));
        c->mrg_pairs = true;
    }
    if (c->mrg_capacity < total) {
        (void)hipFree(c->mrg_keys); (void)hipFree(c->mrg_keys2); (void)hipFree(c->mrg_vals[0]); (void)hipFree(c->mrg_vals[1]);
        (void)hipFree(c->mrg_colors);
        c->mrg_keys = c->mrg_keys2 = nullptr; c->mrg_vals[0] = c->mrg_vals[1] = nullptr; c->mrg_colors = nullptr; c->mrg_capacity = 0;
        const size_t cap = (size_t)total + (size_t)total / 4 + 1024;
        TH_HIP(hipMalloc((void **)&c->mrg_keys, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->mrg_keys2, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->mrg_vals[0], cap * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->mrg_vals[1], cap * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->mrg_colors, (c->mrg_pairs ? 2 : 1) * cap * sizeof(float4)));
        c->mrg_capacity = cap;
    }
    // what arrives is one part per source band, every part in that band's stream order: a stable sort by texel (the
    // owner bits above and the stream index below are left alone), then the blend merges the bands inside each texel
    const int bits = 32 + deposit_texel_bits(c);
    if (th_status s = deposit_temp(c, th::radix_sort_temp_bytes(total, 32, bits))) return s;
    if (!c->dep_total) TH_HIP(hipMalloc((void **)&c->dep_total, 8 * sizeof(uint32_t)));
    TH_HIP(hipMemsetAsync(c->dep_total, 0, 8 * sizeof(uint32_t), c->stream));
    // (the sort ping-pongs between its two buffer pairs: the caller's keys are copied, not sorted in place)
    TH_HIP(hipMemcpyAsync(c->mrg_keys, keys_dev, (size_t)total * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    const int in_b = th::launch_radix_sort_u64(c->mrg_keys, c->mrg_vals[0], c->mrg_keys2, c->mrg_vals[1], total, 32, bits, c->dep_temp, true, c->stream);
    if (target == 2)
        th::launch_draw_blend64(c->flow, c->view, in_b ? c->mrg_keys2 : c->mrg_keys, in_b ? c->mrg_vals[1] : c->mrg_vals[0],
                                static_cast<const float4 *>(colors_dev), c->mrg_colors, total, c->dep_total, c->stream);
    else if (into_view)
        th::launch_view_blend64(c->view, in_b ? c->mrg_keys2 : c->mrg_keys, in_b ? c->mrg_vals[1] : c->mrg_vals[0],
                                static_cast<const float4 *>(colors_dev), c->mrg_colors, total, c->dep_total, c->stream);
    else
        th::launch_deposit_blend64(c->flow, in_b ? c->mrg_keys2 : c->mrg_keys, in_b ? c->mrg_vals[1] : c->mrg_vals[0],
                                   static_cast<const float4 *>(colors_dev), c->mrg_colors, total, c->dep_total, c->stream);
    TH_HIP(hipGetLastError());
    uint32_t too_many = 0;
    if (th_status s = read_back(c, &too_many, c->dep_total, sizeof too_many)) return s;        // (a sync: the input buffers may be reused by the caller now)
    if (too_many) return fail(TH_ERR_UNSUPPORTED, "a texel received fragments of more than 32 source bands");
    return TH_OK;
}

th_status th_deposit_merge(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count)
{
    if (th_status s = use(c)) return s;
    return merge_parted(c, keys_dev, colors_dev, count, 0);
}

th_status th_view_merge(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    return merge_parted(c, keys_dev, colors_dev, count, 1);
}

th_status th_draw_merge(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    return merge_parted(c, keys_dev, colors_dev, count, 2);
}

// ---- draw() of a row-band shard, the exchange issued by the library over its own communicator ---------------------------------
// One pass: this band's fragments parted by owner -> all-to-all -> the owner's merge -> all-gather of the owned ranges.
// du alone: the flow pass; ru alone: the view pass; both: both passes over one rasterisation and one exchange (two all-gathers)
static th_status sharded_pass(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *fragments)
{
    const int world = c->comm_world, rank = c->comm_rank;
    const bool view = ru != nullptr, both = ru != nullptr && du != nullptr;
    const size_t color_bytes = both ? 2 * sizeof(float4) : sizeof(float4);
    uint64_t count = 0;
    void *keys = nullptr, *colors = nullptr;
    if (th_status s = both ? th_draw_emit(c, du, ru, &count, &keys, &colors)
                           : (view ? th_view_emit(c, ru, &count, &keys, &colors) : th_deposit_emit(c, du, &count, &keys, &colors))) return s;
    if (fragments) *fragments = count;
    // every owner's share of what this rank emitted, and of what it will receive
    unsigned long long *bounds = c->x_counts, *sendc = c->x_counts + 33, *recvc = c->x_counts + 65;
    std::vector<unsigned long long> hb((size_t)world + 1, 0ull);
    if (count) {
        th::launch_owner_bounds(static_cast<const unsigned long long *>(keys), (uint32_t)count, (uint32_t)world, bounds, c->stream);
        if (th_status s = read_back(c, hb.data(), bounds, ((size_t)world + 1) * sizeof(unsigned long long))) return s;
    }
    std::vector<size_t> scount((size_t)world), soff((size_t)world), rcount((size_t)world), roff((size_t)world), one((size_t)world, 1), idx((size_t)world);
    std::vector<unsigned long long> hs((size_t)world), hr((size_t)world);
    for (int r = 0; r < world; ++r) { scount[(size_t)r] = (size_t)(hb[(size_t)r + 1] - hb[(size_t)r]); soff[(size_t)r] = (size_t)hb[(size_t)r]; hs[(size_t)r] = scount[(size_t)r]; idx[(size_t)r] = (size_t)r; }
    TH_HIP(hipMemcpyAsync(sendc, hs.data(), (size_t)world * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    if (th::comm_alltoallv(c->comm, sendc, one.data(), idx.data(), recvc, one.data(), idx.data(), sizeof(unsigned long long), world, c->stream))
        return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    if (th_status s = read_back(c, hr.data(), recvc, (size_t)world * sizeof(unsigned long long))) return s;
    size_t total = 0;
    for (int r = 0; r < world; ++r) { rcount[(size_t)r] = (size_t)hr[(size_t)r]; roff[(size_t)r] = total; total += rcount[(size_t)r]; }
    TH_REQUIRE(total < ((size_t)1 << 31), "too many fragments for one owner");
    if (c->x_capacity < total || (both && !c->x_pairs)) {
        (void)hipFree(c->x_keys); (void)hipFree(c->x_colors);
        c->x_keys = nullptr; c->x_colors = nullptr; c->x_capacity = 0;
        c->x_pairs = c->x_pairs || both;
        const size_t cap = std::max(total, c->x_capacity) + total / 4 + 1024;
        TH_HIP(hipMalloc((void **)&c->x_keys, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->x_colors, (c->x_pairs ? 2 : 1) * cap * sizeof(float4)));
        c->x_capacity = cap;
    }
    if (th::comm_alltoallv(c->comm, keys, scount.data(), soff.data(), c->x_keys, rcount.data(), roff.data(), sizeof(unsigned long long), world, c->stream) ||
        th::comm_alltoallv(c->comm, colors, scount.data(), soff.data(), c->x_colors, rcount.data(), roff.data(), color_bytes, world, c->stream))
        return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    if (th_status s = both ? th_draw_merge(c, c->x_keys, c->x_colors, total)
                           : (view ? th_view_merge(c, c->x_keys, c->x_colors, total) : th_deposit_merge(c, c->x_keys, c->x_colors, total))) return s;
    // the owners' texel ranges of the target(s) to every rank, in place
    const size_t texels = (size_t)c->fw * c->fh, chunk = (texels + (size_t)world - 1) / (size_t)world;
    for (int plane_of = 0; plane_of < 2; ++plane_of) {          // 0: the flow texture, 1: the view buffer
        if (plane_of == 0 ? (view && !both) : !view) continue;
        const size_t elem = plane_of ? sizeof(uchar4) : sizeof(float4);
        std::vector<size_t> gb((size_t)world), go((size_t)world);
        for (int r = 0; r < world; ++r) {
            const size_t lo = std::min(texels, (size_t)r * chunk), hi = std::min(texels, ((size_t)r + 1) * chunk);
            gb[(size_t)r] = (hi - lo) * elem; go[(size_t)r] = lo * elem;
        }
        char *plane = plane_of ? reinterpret_cast<char *>(c->view) : reinterpret_cast<char *>(c->flow);
        if (th::comm_allgather_bytes(c->comm, plane + go[(size_t)rank], plane, gb.data(), go.data(), rank, world, c->stream))
            return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    }
    return TH_OK;
}

th_status th_draw_sharded(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *fragments)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(du, "null uniforms");
    TH_REQUIRE(c->comm, "th_draw_sharded needs the job's communicator (th_comm_init)");
    TH_REQUIRE(c->comm_world <= 32, "the owners' merge handles up to 32 ranks");
    const int world = c->comm_world, rank = c->comm_rank, W = c->cfg.width;
    if (!c->x_counts) TH_HIP(hipMalloc((void **)&c->x_counts, 97 * sizeof(unsigned long long)));
    // the neighbouring bands' edge rows of both state buffers (the fp32 row lookup of the vertex stream can land one row
    // beside a line's own row for some texture heights): my first row to the rank below, my last row to the rank above
    c->halo_lo = c->halo_hi = nullptr;
    if (world > 1 && !c->packed) {
        if (th_status s = ensure_identity(c)) return s;
        if (!c->x_halo) TH_HIP(hipMalloc((void **)&c->x_halo, (size_t)4 * W * sizeof(float4)));
        std::vector<size_t> sc((size_t)world, 0), so((size_t)world, 0), rc((size_t)world, 0), ro((size_t)world, 0);
        for (int b = 0; b < 2; ++b) {           // ring buffer b: one exchange each (the rows lie in different allocations)
            const float4 *state = c->ring[(size_t)b];
            std::fill(sc.begin(), sc.end(), 0); std::fill(rc.begin(), rc.end(), 0);
            // to rank - 1: my first row (its `hi`); to rank + 1: my last row (its `lo`).  Offsets are in rows of W texels from `state`.
            if (rank > 0) { sc[(size_t)rank - 1] = 1; so[(size_t)rank - 1] = 0; rc[(size_t)rank - 1] = 1; ro[(size_t)rank - 1] = (size_t)b; }
            if (rank + 1 < world) { sc[(size_t)rank + 1] = 1; so[(size_t)rank + 1] = (size_t)c->cfg.height - 1; rc[(size_t)rank + 1] = 1; ro[(size_t)rank + 1] = 2 + (size_t)b; }
            if (th::comm_alltoallv(c->comm, state, sc.data(), so.data(), c->x_halo, rc.data(), ro.data(), (size_t)W * sizeof(float4), world, c->stream))
                return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
        }
        c->halo_lo = rank > 0 ? c->x_halo : nullptr;
        c->halo_hi = rank + 1 < world ? c->x_halo + (size_t)2 * W : nullptr;
    }
    c->dep_owners = (uint32_t)world;
    if (ru) {
        TH_REQUIRE(memcmp(du->viewSize, ru->viewSize, sizeof du->viewSize) == 0 && memcmp(&du->time, &ru->time, sizeof du->time) == 0 &&
                   memcmp(&du->speedLimit, &ru->speedLimit, sizeof du->speedLimit) == 0,
                   "the two passes of one draw share viewSize, time and speedLimit");
        if (th_status s = view_storage(c)) return s;
    }
    if (ru && drawn_line_width(c, TH_PASS_FLOW) == drawn_line_width(c, TH_PASS_VIEW)) {
        // both passes draw the same lines: one rasterisation, one sort, one exchange of fragments carrying both varyings
        if (th_status s = sharded_pass(c, du, ru, fragments)) return s;
    } else {
        if (th_status s = sharded_pass(c, du, nullptr, fragments)) return s;
        if (ru) if (th_status s = sharded_pass(c, nullptr, ru, nullptr)) return s;
    }
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_view_device_ptr(th_context *c, void **dptr)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(dptr, "null output");
    if (th_status s = view_storage(c)) return s;
    *dptr = c->view;
    return TH_OK;
}

th_status th_flow_device_ptr(th_context *c, void **dptr)
{
    TH_REQUIRE(c && dptr, "null argument");
    *dptr = c->flow;
    return TH_OK;
}

th_status th_frames_resize(th_context *c, int32_t w, int32_t h)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(w > 0 && h > 0 && (uint64_t)w * h < (1ull << 28), "bad frame shape %dx%d", w, h);
    if (w == c->frw && h == c->frh) return TH_OK;
    TH_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k) {
        TH_HIP(hipFree(c->frames[k]));
        c->frames[k] = nullptr;
        TH_HIP(hipMalloc((void **)&c->frames[k], (size_t)w * h * sizeof(uchar4)));
        TH_HIP(hipMemsetAsync(c->frames[k], 0, (size_t)w * h * sizeof(uchar4), c->stream));
    }
    c->frw = w; c->frh = h;
    return TH_OK;
}

th_status th_frames_upload(th_context *c, const uint8_t *rgba8)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(rgba8 && c->frames[0], "no frame buffers (call th_frames_resize) or null pixels");
    TH_HIP(hipMemcpyAsync(c->frames[0], rgba8, (size_t)c->frw * c->frh * sizeof(uchar4), hipMemcpyHostToDevice, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_frames_rotate(th_context *c)
{
    TH_REQUIRE(c, "null context");
    uchar4 *t = c->frames[1]; c->frames[1] = c->frames[0]; c->frames[0] = t;   // utils.step on 2 buffers
    return TH_OK;
}

th_status th_optical_flow(th_context *c, const th_optical_flow_uniforms *u)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(u, "null uniforms");
    TH_REQUIRE(c->frames[0] && c->frames[1], "no frame buffers (call th_frames_resize)");
    th::OpticalFlowParams p{};
    p.view = c->frames[0]; p.last = c->frames[1];      // OpticalFlow.update: view = buffers[0], last = buffers[1]
    p.flow = c->flow;
    p.fr_w = c->frw; p.fr_h = c->frh;
    p.out_w = c->fw; p.out_h = c->fh;
    p.grad_x = 2.0f / (float)c->fw; p.grad_y = 2.0f / (float)c->fh;
    p.u = *u;
    th::launch_optical_flow(p, c->stream);
    TH_HIP(hipGetLastError());
    return TH_OK;
}

th_status th_stats_async(th_context *c, float speed_limit, void **device_counters)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(!c->ring.empty(), "no state buffers");
    float4 *view = nullptr;
    if (th_status s = unpacked_view(c, c->ring[0], 0, &view)) return s;
    th::launch_stats(view, c->texels(), speed_limit, c->partials, c->d_respawned, c->d_counters, c->stream);
    TH_HIP(hipGetLastError());
    if (device_counters) *device_counters = c->d_counters;
    return TH_OK;
}

th_status th_stats(th_context *c, float speed_limit, th_counters *out)
{
    TH_REQUIRE(out, "null output");
    if (th_status s = th_stats_async(c, speed_limit, nullptr)) return s;
    return read_back(c, out, c->d_counters, sizeof *out);
}

// ---- one process per GPU: the communicator of the job's ranks and the path's collective (th_comm.hip) --------------------
th_status th_comm_unique_id(void *id_out)
{
    TH_REQUIRE(id_out, "null output");
    if (th::comm_unique_id(id_out, TH_COMM_ID_BYTES)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    return TH_OK;
}

th_status th_comm_init(th_context *c, const void *id, int32_t rank, int32_t world)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(id, "null communicator id");
    TH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d outside world %d", rank, world);
    TH_REQUIRE(!c->comm, "the context already holds a communicator (th_comm_destroy first)");
    if (th::comm_init(&c->comm, id, TH_COMM_ID_BYTES, rank, world)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    c->comm_rank = rank; c->comm_world = world;
    return TH_OK;
}

th_status th_comm_destroy(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    if (!c->comm) return TH_OK;
    TH_HIP(hipStreamSynchronize(c->stream));
    const int bad = th::comm_destroy(c->comm);
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    if (bad) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    return TH_OK;
}

th_status th_comm_query(th_context *c, th_comm_info *out)
{
    TH_REQUIRE(c && out, "null argument");
    *out = th_comm_info{};
    out->rank = c->comm_rank; out->world = c->comm_world; out->active = c->comm ? 1 : 0;
    int v = 0;
    if (th::comm_available(&v) == 0) out->rccl_version = v;
    return TH_OK;
}

// ---- row-band shards: the whole particle texture on every rank, for the spawners that sample arbitrary particles ----------
static th_status gather_storage(th_context *c, int32_t buffer)
{
    TH_REQUIRE(buffer >= 0 && buffer < (int32_t)c->ring.size(), "bad buffer %d (ring has %zu)", buffer, c->ring.size());
    if (!c->gathered) TH_HIP(hipMalloc((void **)&c->gathered, (size_t)c->cfg.width * c->cfg.global_height * sizeof(float4)));
    c->gathered_of = nullptr;
    return TH_OK;
}

th_status th_state_gather_ptr(th_context *c, int32_t buffer, void **dptr)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(dptr, "null output");
    if (th_status s = gather_storage(c, buffer)) return s;
    c->gathered_of = c->ring[(size_t)buffer];
    *dptr = c->gathered;
    return TH_OK;
}

th_status th_state_gather(th_context *c, int32_t buffer)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(c->comm, "th_state_gather needs the job's communicator (th_comm_init)");
    if (th_status s = gather_storage(c, buffer)) return s;
    if (th_status s = ensure_identity(c)) return s;          // bands travel in texel order
    // the bands of sharding.shard_rows: contiguous, balanced (the first H % world ranks hold one row more)
    const int world = c->comm_world, H = c->cfg.global_height, base = H / world, extra = H % world;
    std::vector<size_t> bytes((size_t)world), offset((size_t)world);
    for (int r = 0; r < world; ++r) {
        const int rows = base + (r < extra ? 1 : 0), row0 = r * base + (r < extra ? r : extra);
        bytes[(size_t)r] = (size_t)rows * c->cfg.width * sizeof(float4);
        offset[(size_t)r] = (size_t)row0 * c->cfg.width * sizeof(float4);
        if (r == c->comm_rank)
            TH_REQUIRE(rows == c->cfg.height && row0 == c->cfg.row0, "this context holds rows %d..%d, rank %d of %d balanced bands holds %d..%d",
                       c->cfg.row0, c->cfg.row0 + c->cfg.height, r, world, row0, row0 + rows);
    }
    float4 *data = nullptr;
    if (th_status s = unpacked_view(c, c->ring[(size_t)buffer], 2, &data)) return s;
    if (th::comm_allgather_bytes(c->comm, data, c->gathered, bytes.data(), offset.data(), c->comm_rank, world, c->stream))
        return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    c->gathered_of = c->ring[(size_t)buffer];
    return TH_OK;
}

th_status th_stats_allreduce(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    if (!c->comm) return TH_OK;                    // a single-rank job: the local block is the global one
    if (th::comm_allreduce_counters(c->comm, c->d_counters, c->stream)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    return TH_OK;
}

th_status th_stats_global(th_context *c, float speed_limit, th_counters *out)
{
    TH_REQUIRE(out, "null output");
    if (th_status s = th_stats_async(c, speed_limit, nullptr)) return s;
    if (th_status s = th_stats_allreduce(c)) return s;
    return read_back(c, out, c->d_counters, sizeof *out);
}

th_status th_sync(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
}

th_status th_stream(th_context *c, void **hip_stream)
{
    TH_REQUIRE(c && hip_stream, "null argument");
    *hip_stream = (void *)c->stream;
    return TH_OK;
}

th_status th_state_device_ptr(th_context *c, int32_t buffer, void **dptr)
{
    TH_REQUIRE(c && dptr, "null argument");
    if (th_status s = use(c)) return s;
    bool moved = false;
    if (th_status s = ensure_identity(c, &moved)) return s;      // the pointer is only meaningful in texel order
    // the caller reads the buffer on a stream of its own: what was just enqueued on the context's stream must be done
    if (moved) TH_HIP(hipStreamSynchronize(c->stream));
    TH_REQUIRE(buffer >= 0 && buffer < (int32_t)c->ring.size(), "bad buffer index %d", buffer);
    *dptr = c->ring[buffer];
    // (use() above dropped the geometry a view pass would reuse from the last flow pass: whoever holds this address may write
    // the state behind the library's back - after such a write, call any state entry point, or th_sync, before th_view_draw)
    return TH_OK;
}

th_status th_timer_start(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    TH_HIP(hipEventRecord(c->ev0, c->stream));
    return TH_OK;
}

th_status th_timer_stop(th_context *c, float *elapsed_ms)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(elapsed_ms, "null output");
    TH_HIP(hipEventRecord(c->ev1, c->stream));
    TH_HIP(hipEventSynchronize(c->ev1));
    TH_HIP(hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return TH_OK;
}

th_status th_kernel_timing(th_context *c, int32_t enable)
{
    if (th_status s = use(c, true)) return s;
    c->kernel_timing = enable != 0;
    if (!enable) c->kt_used = 0;
    return TH_OK;
}

th_status th_kernel_timing_read(th_context *c, float *mean_ms, int32_t *launches)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(mean_ms && launches, "null output");
    TH_HIP(hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (size_t k = 0; k + 1 < c->kt_used; k += 2) {
        float ms = 0.0f;
        TH_HIP(hipEventElapsedTime(&ms, c->kt_events[k], c->kt_events[k + 1]));
        sum += ms;
    }
    *launches = (int32_t)(c->kt_used / 2);
    *mean_ms = *launches ? (float)(sum / *launches) : 0.0f;
    c->kt_used = 0;
    return TH_OK;
}

th_status th_shapes(th_context *c, th_shapes_info *out)
{
    TH_REQUIRE(c && out, "null argument");
    out->state_w = c->cfg.width; out->state_h = c->cfg.height;
    out->flow_w = c->fw; out->flow_h = c->fh;
    out->frames_w = c->frw; out->frames_h = c->frh;
    return TH_OK;
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
