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
#include "th_ctx.hpp"

using namespace thi;

namespace {

thread_local std::string g_error;

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

// the switches a context starts with
void options_from_environment(th_options &o)
{
    auto number = [](const char *name, long long otherwise) { const char *e = getenv(name); return e && *e ? atoll(e) : otherwise; };
    o.bucket = (int)number("TH_BUCKET", -1);
    o.resort_steps = (int)number("TH_RESORT_STEPS", 64); if (o.resort_steps <= 0) o.resort_steps = 64;
    o.rebucket_steps = (int)number("TH_REBUCKET_STEPS", 256); if (o.rebucket_steps <= 0) o.rebucket_steps = 256;
    o.fuse = number("TH_FUSE", 1) != 0;
    o.graph = number("TH_GRAPH", 1) != 0;
    o.force_generic = getenv("TH_FORCE_GENERIC") != nullptr && number("TH_FORCE_GENERIC", 1) != 0;
    const char *d = getenv("TH_DRAW");
    o.draw = !d ? -1 : (!strcmp(d, "bins") ? 1 : (!strcmp(d, "stream") ? 0 : -1));
    o.draw_reuse = number("TH_DRAW_REUSE", 1) != 0;
    o.async_sort = number("TH_ASYNC_SORT", 1) != 0;
    o.skip_unseen = number("TH_SKIP_UNSEEN", 1) != 0;
    o.bins_pool = (uint32_t)number("TH_BINS_POOL", 0);
    o.bins_pages = (int)number("TH_BINS_PAGES", 0); if (o.bins_pages > (int)th::kBinPagesLimit || o.bins_pages < -(int)th::kBinPagesLimit) o.bins_pages = 0;
}

}  // namespace

namespace thi {

std::string &last_error() { return g_error; }

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

// keeps_lines: the entry point leaves the particle state and the per-line / per-fragment buffers of the last draw pass
// alone, so that a view pass can still reuse the flow pass's geometry (deposit_run)
th_status use(th_context *c, bool keeps_lines)
{
    if (!c) return fail(TH_ERR_INVALID, "null context");
    TH_HIP(hipSetDevice(c->cfg.device));
    if (!keeps_lines) { c->drawn.valid = false; c->fused_stats.valid = false; }
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
        state_written(c, *out);
    } else if (target == TH_TARGET_TARGETS) {
        *out = c->targets;
        c->targets_checked = false;
    } else if (target >= 0 && target < (int32_t)c->ring.size()) {
        *out = c->ring[target];
        state_written(c, *out);
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

// a few words from the device to the host, through pinned memory, with the stream's work before them finished
th_status read_back(th_context *c, void *host, const void *dev, size_t bytes)
{
    TH_REQUIRE(bytes <= kPinnedBytes, "read_back of %zu bytes", bytes);
    if (!c->pinned) TH_HIP(hipHostMalloc(&c->pinned, kPinnedBytes, hipHostMallocDefault));
    TH_HIP(hipMemcpyAsync(c->pinned, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    memcpy(host, c->pinned, bytes);
    return TH_OK;
}

}  // namespace thi

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
    options_from_environment(c->opt);
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
    if (c->side) (void)hipStreamSynchronize(c->side);          // (a re-sort beside the last draw may still be running there)
    if (c->side2) (void)hipStreamSynchronize(c->side2);
    if (c->comm) { (void)c->transport->destroy(c->comm); c->comm = nullptr; }
    (void)hipFree(c->d_status); (void)hipFree(c->own_mem);
    for (float4 *b : c->ring) (void)hipFree(b);
    (void)hipFree(c->flow); (void)hipFree(c->flow_dec); (void)hipFree(c->flow3); (void)hipFree(c->targets); (void)hipFree(c->lut_block);
    (void)hipFree(c->frames[0]); (void)hipFree(c->frames[1]);
    (void)hipFree(c->dep_count); (void)hipFree(c->dep_offset); (void)hipFree(c->dep_blocks); (void)hipFree(c->dep_total);
    (void)hipFree(c->dep_record); (void)hipFree(c->dep_lists); (void)hipFree(c->mrg_keys2); (void)hipFree(c->mrg_colors);
    (void)hipFree(c->bin_mem); (void)hipFree(c->draw_blocks); (void)hipFree(c->draw_block_flags); (void)hipFree(c->d_row_draws); (void)hipFree(c->src_row_index); (void)hipFree(c->src_col_index); (void)hipFree(c->src_slots); (void)hipFree(c->edge_rows); (void)hipFree(c->crowd_mem); (void)hipFree(c->chunk_table);
    (void)hipFree(c->bins_keys); (void)hipFree(c->bins_colors); (void)hipFree(c->crowd_keys); (void)hipFree(c->crowd_sorted); (void)hipFree(c->crowd_parted); (void)hipFree(c->crowd_windows); (void)hipFree(c->gathered);
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
    (void)hipFree(c->image); (void)hipFree(c->view_screen); (void)hipFree(c->colormap);
    for (uchar4 *b : c->view_ring) (void)hipFree(b);
    (void)hipFree(c->d_flag); (void)hipFree(c->partials); (void)hipFree(c->fused_parts); (void)hipFree(c->d_counters); (void)hipFree(c->d_respawned);
    clear_graphs(c);
    for (float4 *t : c->tmp) (void)hipFree(t);
    for (th_context::SlotOrder &o : c->orders) { (void)hipFree(o.perm); (void)hipFree(o.chunks); (void)hipFree(o.records); (void)hipFree(o.nchunks); }
    (void)hipFree(c->spare); (void)hipFree(c->tile_mem); (void)hipFree(c->block_records); (void)hipFree(c->asort.dst); (void)hipFree(c->seen.bytes);
    if (c->asort.ready) (void)hipEventDestroy(c->asort.ready);
    if (c->asort.done) (void)hipEventDestroy(c->asort.done);
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
        state_written(c, c->ring[b]);
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
    if (c->fused_stats.valid && c->fused_stats.buf == c->ring[0] && memcmp(&c->fused_stats.limit, &speed_limit, sizeof speed_limit) == 0) {
        // the launch that wrote this state took its statistics on the way (th_step_n): only the fold is left
        const uint32_t n = c->fused_stats.nparts;
        th::launch_stats_fold(c->fused_parts, n, c->fused_parts + n, c->texels(), c->d_respawned, c->d_counters, c->stream);
        TH_HIP(hipGetLastError());
        if (device_counters) *device_counters = c->d_counters;
        return TH_OK;
    }
    float4 *view = nullptr;                      // (a packed ring: the f32 copy of what its texels decode to)
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
    state_written(c, c->ring[buffer]);        // (whoever holds the address may write through it)
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

th_status th_option_set(th_context *c, int32_t option, int64_t value)
{
    if (th_status s = use(c)) return s;
    th_options &o = c->opt;
    switch (option) {
    case TH_OPT_BUCKET: TH_REQUIRE(value >= -1 && value <= 1, "TH_OPT_BUCKET takes -1, 0 or 1"); o.bucket = (int)value; break;
    case TH_OPT_RESORT_STEPS: TH_REQUIRE(value > 0 && value < (1 << 30), "TH_OPT_RESORT_STEPS must be positive"); o.resort_steps = (int)value; break;
    case TH_OPT_REBUCKET_STEPS: TH_REQUIRE(value > 0 && value < (1 << 30), "TH_OPT_REBUCKET_STEPS must be positive"); o.rebucket_steps = (int)value; break;
    case TH_OPT_FUSE: o.fuse = value != 0; break;
    case TH_OPT_GRAPH: o.graph = value != 0; break;
    case TH_OPT_FORCE_GENERIC: o.force_generic = value != 0; break;
    case TH_OPT_DRAW_REUSE: o.draw_reuse = value != 0; break;
    case TH_OPT_SKIP_UNSEEN: o.skip_unseen = value != 0; break;
    case TH_OPT_ASYNC_SORT: o.async_sort = value != 0; if (!o.async_sort) { if (th_status s = asort_drop(c)) return s; } break;
    case TH_OPT_BINS_POOL: TH_REQUIRE(value >= 0 && value < (1ll << 32), "TH_OPT_BINS_POOL out of range"); o.bins_pool = (uint32_t)value; break;
#ifdef TH_TESTING
    case TH_OPT_INJECT_FAILURE: TH_REQUIRE(value >= 0 && value <= 4, "TH_OPT_INJECT_FAILURE takes 0..4"); o.inject_failure = (int)value; break;
#endif
    case TH_OPT_BINS_PAGES:
        TH_REQUIRE(value >= -(int64_t)th::kBinPagesLimit && value <= (int64_t)th::kBinPagesLimit && value != 1 && value != -1, "TH_OPT_BINS_PAGES takes 0, or 2..%u (negative: never widened)", th::kBinPagesLimit);
        o.bins_pages = (int)value;
        if (c->chunk_table) {           // (takes effect at once: the table is laid out again by the next binned pass)
            TH_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->chunk_table); c->chunk_table = nullptr; (void)hipFree(c->bin_mem); c->bin_mem = nullptr; c->bin_capacity = 0; c->bin_max_pages = 0;
        }
        break;
    default: return fail(TH_ERR_INVALID, "unknown option %d", option);
    }
    clear_graphs(c);                 // (captured sequences were planned under the old switches)
    return TH_OK;
}

th_status th_option_get(th_context *c, int32_t option, int64_t *value)
{
    TH_REQUIRE(c && value, "null argument");
    const th_options &o = c->opt;
    switch (option) {
    case TH_OPT_BUCKET: *value = o.bucket; break;
    case TH_OPT_RESORT_STEPS: *value = o.resort_steps; break;
    case TH_OPT_REBUCKET_STEPS: *value = o.rebucket_steps; break;
    case TH_OPT_FUSE: *value = o.fuse; break;
    case TH_OPT_GRAPH: *value = o.graph; break;
    case TH_OPT_FORCE_GENERIC: *value = o.force_generic; break;
    case TH_OPT_DRAW_REUSE: *value = o.draw_reuse; break;
    case TH_OPT_ASYNC_SORT: *value = o.async_sort; break;
    case TH_OPT_SKIP_UNSEEN: *value = o.skip_unseen; break;
    case TH_OPT_BINS_POOL: *value = o.bins_pool; break;
#ifdef TH_TESTING
    case TH_OPT_INJECT_FAILURE: *value = o.inject_failure; break;
#endif
    case TH_OPT_BINS_PAGES: *value = o.bins_pages; break;
    default: return fail(TH_ERR_INVALID, "unknown option %d", option);
    }
    return TH_OK;
}

}  // extern "C"
