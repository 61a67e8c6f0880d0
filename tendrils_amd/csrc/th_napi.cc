// th_napi.cc - thin N-API shim over the C ABI in include/tendrils_hip.h.
//
// The reference's host is JavaScript: `Particles` (src/particles.js) drives WebGL through
// stackgl (gl-fbo / gl-shader / gl-texture2d).  This addon is the replacement for that GL
// layer: tendrils_amd/js/*.js keep the reference's object model and call these functions
// where the reference called gl.*.  Every export maps 1:1 to one th_* entry point; a
// non-zero status becomes a thrown JS Error carrying th_last_error() (gl-fbo / gl-shader
// throw JS Errors the same way: docs/js/index.js:42).
//
// Plain N-API (node_api.h, ABI-stable C interface); built with g++, no node-gyp:
//   g++ -shared -fPIC -I/usr/include/node th_napi.cc -ltendrils_hip
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "tendrils_hip.h"

namespace {

#define NAPI_OK(call)                                                  \
    do {                                                               \
        if ((call) != napi_ok) {                                       \
            napi_throw_error(env, nullptr, "N-API call failed: " #call); \
            return nullptr;                                            \
        }                                                              \
    } while (0)

napi_value throw_status(napi_env env, th_status st, const char *what)
{
    char msg[640];
    snprintf(msg, sizeof msg, "tendrils_hip %s: status %d: %s", what, (int)st, th_last_error());
    napi_throw_error(env, nullptr, msg);
    return nullptr;
}

#define TH_CALL(what, expr)                                   \
    do {                                                      \
        th_status st_ = (expr);                               \
        if (st_ != TH_OK) return throw_status(env, st_, what); \
    } while (0)

struct Args {
    napi_env env;
    size_t argc = 12;
    napi_value argv[12];
    bool ok = true;
    Args(napi_env e, napi_callback_info info) : env(e)
    {
        ok = napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) == napi_ok;
    }
    th_context *ctx(size_t i)
    {
        void *p = nullptr;
        if (i >= argc || napi_get_value_external(env, argv[i], &p) != napi_ok || !p) { ok = false; return nullptr; }
        return *static_cast<th_context **>(p);
    }
    int32_t i32(size_t i)
    {
        int32_t v = 0;
        if (i >= argc || napi_get_value_int32(env, argv[i], &v) != napi_ok) ok = false;
        return v;
    }
    double f64(size_t i)
    {
        double v = 0;
        if (i >= argc || napi_get_value_double(env, argv[i], &v) != napi_ok) ok = false;
        return v;
    }
    // typed array -> raw pointer + element count
    void *typed(size_t i, napi_typedarray_type want, size_t *len)
    {
        napi_typedarray_type t;
        void *data = nullptr;
        size_t n = 0;
        if (i >= argc || napi_get_typedarray_info(env, argv[i], &t, &n, &data, nullptr, nullptr) != napi_ok || t != want) {
            ok = false;
            return nullptr;
        }
        if (len) *len = n;
        return data;
    }
    // Float32Array whose layout is the named uniform struct
    template <typename T>
    bool uniforms(size_t i, T *out)
    {
        size_t n = 0;
        float *f = static_cast<float *>(typed(i, napi_float32_array, &n));
        if (!f || n * sizeof(float) != sizeof(T)) { ok = false; return false; }
        memcpy(out, f, sizeof(T));
        return true;
    }
};

#define BAD_ARGS(name)                                              \
    do {                                                            \
        napi_throw_type_error(env, nullptr, name ": bad arguments"); \
        return nullptr;                                             \
    } while (0)

napi_value undefined(napi_env env)
{
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

void finalize_ctx(napi_env, void *data, void *)
{
    th_context **slot = static_cast<th_context **>(data);
    if (*slot) th_destroy(*slot);
    delete slot;
}

// create(device, width, height, globalHeight, row0, numBuffers, mode[, stateFormat]) -> handle
napi_value Create(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_config cfg{};
    cfg.device = a.i32(0); cfg.width = a.i32(1); cfg.height = a.i32(2); cfg.global_height = a.i32(3);
    cfg.row0 = a.i32(4); cfg.num_buffers = a.i32(5); cfg.mode = a.i32(6);
    cfg.state_format = a.argc > 7 ? a.i32(7) : TH_STATE_F32;
    if (!a.ok) BAD_ARGS("create");
    th_context *c = nullptr;
    TH_CALL("th_create", th_create(&cfg, &c));
    th_context **slot = new th_context *(c);
    napi_value ext;
    NAPI_OK(napi_create_external(env, slot, finalize_ctx, nullptr, &ext));
    return ext;
}

napi_value Destroy(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    void *p = nullptr;
    if (a.argc < 1 || napi_get_value_external(env, a.argv[0], &p) != napi_ok || !p) BAD_ARGS("destroy");
    th_context **slot = static_cast<th_context **>(p);
    if (*slot) { th_destroy(*slot); *slot = nullptr; }
    return undefined(env);
}

napi_value AbiVersion(napi_env env, napi_callback_info)
{
    napi_value v;
    NAPI_OK(napi_create_int32(env, th_abi_version(), &v));
    return v;
}

napi_value DeviceCount(napi_env env, napi_callback_info)
{
    int32_t n = 0;
    TH_CALL("th_device_count", th_device_count(&n));
    napi_value v;
    NAPI_OK(napi_create_int32(env, n, &v));
    return v;
}

#define CTX_ONLY(fn_name, th_fn)                                   \
    napi_value fn_name(napi_env env, napi_callback_info info)      \
    {                                                              \
        Args a(env, info);                                         \
        th_context *c = a.ctx(0);                                  \
        if (!a.ok) BAD_ARGS(#th_fn);                               \
        TH_CALL(#th_fn, th_fn(c));                                 \
        return undefined(env);                                     \
    }

#define CTX_I32(fn_name, th_fn)                                    \
    napi_value fn_name(napi_env env, napi_callback_info info)      \
    {                                                              \
        Args a(env, info);                                         \
        th_context *c = a.ctx(0);                                  \
        int32_t v = a.i32(1);                                      \
        if (!a.ok) BAD_ARGS(#th_fn);                               \
        TH_CALL(#th_fn, th_fn(c, v));                              \
        return undefined(env);                                     \
    }

CTX_I32(SetMode, th_set_mode)
CTX_I32(Setup, th_setup)
CTX_ONLY(FlowClear, th_flow_clear)
CTX_ONLY(TargetsClear, th_targets_clear)
CTX_I32(SpawnInit, th_spawn_init)
CTX_ONLY(FramesRotate, th_frames_rotate)
CTX_ONLY(Sync, th_sync)

napi_value NumBuffers(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_num_buffers");
    int32_t n = 0;
    TH_CALL("th_num_buffers", th_num_buffers(c, &n));
    napi_value v;
    NAPI_OK(napi_create_int32(env, n, &v));
    return v;
}

// uploadState(ctx, buffer, Float32Array, x0, y0, w, h) / downloadState(...)
template <bool UP>
napi_value StateXfer(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t buffer = a.i32(1);
    size_t n = 0;
    float *px = static_cast<float *>(a.typed(2, napi_float32_array, &n));
    int32_t x0 = a.i32(3), y0 = a.i32(4), w = a.i32(5), h = a.i32(6);
    if (!a.ok || w <= 0 || h <= 0 || n < (size_t)w * h * 4) BAD_ARGS("state transfer");
    if (UP) TH_CALL("th_upload_state", th_upload_state(c, buffer, px, x0, y0, w, h));
    else TH_CALL("th_download_state", th_download_state(c, buffer, px, x0, y0, w, h));
    return undefined(env);
}

napi_value FlowResize(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t w = a.i32(1), h = a.i32(2);
    if (!a.ok) BAD_ARGS("th_flow_resize");
    TH_CALL("th_flow_resize", th_flow_resize(c, w, h));
    return undefined(env);
}

napi_value FramesResize(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t w = a.i32(1), h = a.i32(2);
    if (!a.ok) BAD_ARGS("th_frames_resize");
    TH_CALL("th_frames_resize", th_frames_resize(c, w, h));
    return undefined(env);
}

// (ctx, Float32Array) transfers: the array must hold what the library will read or write - the size comes from the
// context's own shapes (th_shapes), not from the caller
static size_t texture_floats(th_context *c, int which)
{
    th_shapes_info s{};
    if (th_shapes(c, &s) != TH_OK) return (size_t)-1;
    if (which == 0) return (size_t)s.flow_w * s.flow_h * 4;
    if (which == 1) return (size_t)s.state_w * s.state_h * 4;
    return (size_t)s.frames_w * s.frames_h * 4;          // (bytes for the RGBA8 frames)
}
#define CTX_F32(fn_name, th_fn, which)                                           \
    napi_value fn_name(napi_env env, napi_callback_info info)                    \
    {                                                                            \
        Args a(env, info);                                                       \
        th_context *c = a.ctx(0);                                                \
        size_t n = 0;                                                            \
        float *px = static_cast<float *>(a.typed(1, napi_float32_array, &n));    \
        if (!a.ok || n < texture_floats(c, which)) BAD_ARGS(#th_fn);             \
        TH_CALL(#th_fn, th_fn(c, px));                                           \
        return undefined(env);                                                   \
    }
CTX_F32(FlowUpload, th_flow_upload, 0)
CTX_F32(FlowDownload, th_flow_download, 0)
CTX_F32(TargetsUpload, th_targets_upload, 1)
CTX_F32(TargetsDownload, th_targets_download, 1)

napi_value FramesUpload(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0;
    uint8_t *px = static_cast<uint8_t *>(a.typed(1, napi_uint8_array, &n));
    if (!a.ok || n < texture_floats(c, 2)) BAD_ARGS("th_frames_upload");
    TH_CALL("th_frames_upload", th_frames_upload(c, px));
    return undefined(env);
}

// ---- view pass --------------------------------------------------------------------------------------------
// viewDraw(ctx, Float32Array(16) th_render_uniforms) -> fragments
napi_value ViewDraw(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_render_uniforms u;
    a.uniforms(1, &u);
    if (!a.ok) BAD_ARGS("th_view_draw");
    uint64_t fragments = 0;
    TH_CALL("th_view_draw", th_view_draw(c, &u, &fragments));
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)fragments, &v));
    return v;
}

// draw(ctx, Float32Array(4) th_deposit_uniforms, Float32Array(16) th_render_uniforms) -> fragments (both passes in one)
napi_value Draw(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_deposit_uniforms d;
    th_render_uniforms u;
    a.uniforms(1, &d);
    a.uniforms(2, &u);
    if (!a.ok) BAD_ARGS("th_draw");
    uint64_t fragments = 0;
    TH_CALL("th_draw", th_draw(c, &d, &u, &fragments));
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)fragments, &v));
    return v;
}

// viewFill(ctx, Float32Array(4) rgba)
napi_value ViewFill(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0;
    float *rgba = static_cast<float *>(a.typed(1, napi_float32_array, &n));
    if (!a.ok || n < 4) BAD_ARGS("th_view_fill");
    TH_CALL("th_view_fill", th_view_fill(c, rgba));
    return undefined(env);
}

napi_value ViewClear(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_view_clear");
    TH_CALL("th_view_clear", th_view_clear(c));
    return undefined(env);
}

// Tendrils.buffers: viewBuffers(ctx, count), viewBind(ctx, index | -1), viewCopy(ctx, index), viewStepBuffers(ctx)
template <th_status (*FN)(th_context *, int32_t)>
napi_value ViewIndexed(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t k = a.i32(1);
    if (!a.ok) BAD_ARGS("th_view_buffers / th_view_bind / th_view_copy");
    TH_CALL("th_view_buffers / th_view_bind / th_view_copy", FN(c, k));
    return undefined(env);
}

napi_value ViewStepBuffers(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_view_step_buffers");
    TH_CALL("th_view_step_buffers", th_view_step_buffers(c));
    return undefined(env);
}

// viewDownload(ctx) -> Uint8Array (flow shape, RGBA8, row-major)
napi_value ViewDownload(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_view_download");
    const size_t bytes = texture_floats(c, 0);       // flow texels x 4 channels, one byte each here
    napi_value buf, arr;
    void *data = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, bytes, &data, &buf));
    TH_CALL("th_view_download", th_view_download(c, static_cast<uint8_t *>(data)));
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, bytes, buf, 0, &arr));
    return arr;
}

// colormapUpload(ctx, Float32Array rgba, w, h)
napi_value ColormapUpload(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0;
    float *px = static_cast<float *>(a.typed(1, napi_float32_array, &n));
    int32_t w = a.i32(2), h = a.i32(3);
    if (!a.ok || w <= 0 || h <= 0 || n < (size_t)w * (size_t)h * 4) BAD_ARGS("th_colormap_upload");
    TH_CALL("th_colormap_upload", th_colormap_upload(c, px, w, h));
    return undefined(env);
}

// exportViewLines(ctx, Float32Array(16) th_render_uniforms) -> Float32Array (12 floats per line)
napi_value ExportViewLines(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_render_uniforms u;
    a.uniforms(1, &u);
    if (!a.ok) BAD_ARGS("th_export_view_lines");
    uint64_t n = 0;
    TH_CALL("th_export_view_lines", th_export_view_lines(c, &u, nullptr, 0, &n));
    napi_value buf, arr;
    void *data = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 12 * sizeof(float), &data, &buf));
    if (n) TH_CALL("th_export_view_lines", th_export_view_lines(c, &u, static_cast<float *>(data), n, &n));
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, (size_t)n * 12, buf, 0, &arr));
    return arr;
}

// step(ctx, Float32Array(19) uniforms, target)
napi_value Step(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_logic_uniforms u;
    a.uniforms(1, &u);
    int32_t target = a.i32(2);
    if (!a.ok) BAD_ARGS("th_step");
    TH_CALL("th_step", th_step(c, &u, target));
    return undefined(env);
}

// stepN(ctx, uniforms, time0, dtMs, n)
napi_value StepN(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_logic_uniforms u;
    a.uniforms(1, &u);
    double t0 = a.f64(2), dt = a.f64(3);
    int32_t n = a.i32(4);
    if (!a.ok) BAD_ARGS("th_step_n");
    TH_CALL("th_step_n", th_step_n(c, &u, t0, dt, n));
    return undefined(env);
}

napi_value SpawnBall(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_spawn_ball_uniforms u;
    a.uniforms(1, &u);
    int32_t target = a.i32(2);
    if (!a.ok) BAD_ARGS("th_spawn_ball");
    TH_CALL("th_spawn_ball", th_spawn_ball(c, &u, target));
    return undefined(env);
}

// spawnSample(ctx, Float32Array(17) float uniforms, samples, apply, source, target)
napi_value SpawnSample(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0;
    float *f = static_cast<float *>(a.typed(1, napi_float32_array, &n));
    th_spawn_sample_uniforms u{};
    if (!f || n != 17) a.ok = false;
    else memcpy(&u, f, 17 * sizeof(float));
    u.samples = a.i32(2); u.apply = a.i32(3);
    int32_t source = a.i32(4), target = a.i32(5);
    if (!a.ok) BAD_ARGS("th_spawn_sample");
    TH_CALL("th_spawn_sample", th_spawn_sample(c, &u, source, target));
    return undefined(env);
}

// spawnDirect(ctx, Float32Array(17) float uniforms, source, target)
napi_value SpawnDirect(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0;
    float *f = static_cast<float *>(a.typed(1, napi_float32_array, &n));
    th_spawn_sample_uniforms u{};
    if (!f || n != 17) a.ok = false;
    else memcpy(&u, f, 17 * sizeof(float));
    u.samples = 0; u.apply = 2;
    int32_t source = a.i32(2), target = a.i32(3);
    if (!a.ok) BAD_ARGS("th_spawn_direct");
    TH_CALL("th_spawn_direct", th_spawn_direct(c, &u, source, target));
    return undefined(env);
}

// spawnImageUpload(ctx, Float32Array rgba, w, h)
napi_value SpawnImageUpload(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0;
    float *px = static_cast<float *>(a.typed(1, napi_float32_array, &n));
    int32_t w = a.i32(2), h = a.i32(3);
    if (!a.ok || w <= 0 || h <= 0 || n < (size_t)w * (size_t)h * 4) BAD_ARGS("th_spawn_image_upload");
    TH_CALL("th_spawn_image_upload", th_spawn_image_upload(c, px, w, h));
    return undefined(env);
}

// spawnImageTriangles(ctx, Float32Array positions, Float32Array [viewSize.x, viewSize.y, r, g, b, a], w, h)
napi_value SpawnImageTriangles(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0, m = 0;
    float *pos = static_cast<float *>(a.typed(1, napi_float32_array, &n));
    float *vc = static_cast<float *>(a.typed(2, napi_float32_array, &m));
    int32_t w = a.i32(3), h = a.i32(4);
    if (!a.ok || m != 6 || n % 6 != 0) BAD_ARGS("th_spawn_image_triangles");
    TH_CALL("th_spawn_image_triangles", th_spawn_image_triangles(c, pos, (int32_t)(n / 6), vc, vc + 2, w, h));
    return undefined(env);
}

napi_value OpticalFlow(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_optical_flow_uniforms u;
    a.uniforms(1, &u);
    if (!a.ok) BAD_ARGS("th_optical_flow");
    TH_CALL("th_optical_flow", th_optical_flow(c, &u));
    return undefined(env);
}

// flowDeposit(ctx, Float32Array [viewSize.x, viewSize.y, time, speedLimit]) -> fragments
napi_value FlowDeposit(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_deposit_uniforms u;
    a.uniforms(1, &u);
    if (!a.ok) BAD_ARGS("th_flow_deposit");
    uint64_t fragments = 0;
    TH_CALL("th_flow_deposit", th_flow_deposit(c, &u, &fragments));
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)fragments, &v));
    return v;
}

// exportLines(ctx, Float32Array [viewSize.x, viewSize.y, time, speedLimit]) -> Float32Array (12 floats per line)
napi_value ExportLines(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_deposit_uniforms u;
    a.uniforms(1, &u);
    if (!a.ok) BAD_ARGS("th_export_lines");
    uint64_t n = 0;
    TH_CALL("th_export_lines", th_export_lines(c, &u, nullptr, 0, &n));
    napi_value buf, arr;
    void *data = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 12 * sizeof(float), &data, &buf));
    if (n) TH_CALL("th_export_lines", th_export_lines(c, &u, static_cast<float *>(data), n, &n));
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, (size_t)n * 12, buf, 0, &arr));
    return arr;
}

// stats(ctx, speedLimit) -> {particles, live, nan, capped, sumSpeed, maxSpeed}
napi_value Stats(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    double limit = a.f64(1);
    if (!a.ok) BAD_ARGS("th_stats");
    th_counters k{};
    TH_CALL("th_stats", th_stats(c, (float)limit, &k));
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    const char *names[] = {"particles", "live", "nan", "capped", "respawned", "sumSpeed", "maxSpeed"};
    double vals[] = {(double)k.particles, (double)k.live, (double)k.nan, (double)k.capped, (double)k.respawned,
                     k.sum_speed, k.max_speed};
    for (int i = 0; i < 7; ++i) {
        NAPI_OK(napi_create_double(env, vals[i], &v));
        NAPI_OK(napi_set_named_property(env, o, names[i], v));
    }
    return o;
}

napi_value counters_object(napi_env env, const th_counters &k)
{
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    const char *names[] = {"particles", "live", "nan", "capped", "respawned", "sumSpeed", "maxSpeed"};
    double vals[] = {(double)k.particles, (double)k.live, (double)k.nan, (double)k.capped, (double)k.respawned,
                     k.sum_speed, k.max_speed};
    for (int i = 0; i < 7; ++i) {
        NAPI_OK(napi_create_double(env, vals[i], &v));
        NAPI_OK(napi_set_named_property(env, o, names[i], v));
    }
    return o;
}

// ---- one Node process per GPU (row-band shards): the communicator of the job's ranks.  Rank 0 makes the id, the host
// application hands its 128 bytes to the other ranks (a file, a socket, an environment variable), every rank joins ----
// commUniqueId() -> Uint8Array(128)
napi_value CommUniqueId(napi_env env, napi_callback_info)
{
    napi_value buf, arr;
    void *data = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, TH_COMM_ID_BYTES, &data, &buf));
    TH_CALL("th_comm_unique_id", th_comm_unique_id(data));
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, TH_COMM_ID_BYTES, buf, 0, &arr));
    return arr;
}

#ifdef TH_TESTING
// commLoopbackId() -> Uint8Array(128): the id of an in-process world (th_comm_loopback_id)
napi_value CommLoopbackId(napi_env env, napi_callback_info)
{
    napi_value buf, arr;
    void *data = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, TH_COMM_ID_BYTES, &data, &buf));
    TH_CALL("th_comm_loopback_id", th_comm_loopback_id(data));
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, TH_COMM_ID_BYTES, buf, 0, &arr));
    return arr;
}
#endif

// commInit(ctx, id: Uint8Array(128), rank, world)  (collective: returns when every rank has joined)
napi_value CommInit(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    size_t n = 0;
    void *id = a.typed(1, napi_uint8_array, &n);
    int32_t rank = a.i32(2), world = a.i32(3);
    if (!a.ok || n != TH_COMM_ID_BYTES) BAD_ARGS("th_comm_init");
    TH_CALL("th_comm_init", th_comm_init(c, id, rank, world));
    return undefined(env);
}

napi_value CommDestroy(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_comm_destroy");
    TH_CALL("th_comm_destroy", th_comm_destroy(c));
    return undefined(env);
}

// commQuery(ctx) -> {active, rank, world, rcclVersion}
napi_value CommQuery(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_comm_query");
    th_comm_info q{};
    TH_CALL("th_comm_query", th_comm_query(c, &q));
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    const char *names[] = {"active", "rank", "world", "rcclVersion"};
    int32_t vals[] = {q.active, q.rank, q.world, q.rccl_version};
    for (int i = 0; i < 4; ++i) {
        NAPI_OK(napi_create_int32(env, vals[i], &v));
        NAPI_OK(napi_set_named_property(env, o, names[i], v));
    }
    return o;
}

// statsAllreduce(ctx): the device block of the last statistics pass reduced over the ranks, on the context's stream
napi_value StatsAllreduce(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_stats_allreduce");
    TH_CALL("th_stats_allreduce", th_stats_allreduce(c));
    return undefined(env);
}

// statsGlobal(ctx, speedLimit) -> the job's counters (every rank gets the same object)
napi_value StatsGlobal(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    double limit = a.f64(1);
    if (!a.ok) BAD_ARGS("th_stats_global");
    th_counters k{};
    TH_CALL("th_stats_global", th_stats_global(c, (float)limit, &k));
    return counters_object(env, k);
}

// ---- multi-GPU exchange primitives (row-band shards): device addresses travel as BigInt; the transport between the
// ranks' processes is the host application's (the Python host uses torch.distributed / RCCL, tendrils_amd/sharding.py) ----
bool bigint_ptr(napi_env env, napi_value v, void **out)        // BigInt | null | undefined -> address
{
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return false;
    if (t == napi_null || t == napi_undefined) { *out = nullptr; return true; }
    uint64_t u = 0;
    bool lossless = false;
    if (t != napi_bigint || napi_get_value_bigint_uint64(env, v, &u, &lossless) != napi_ok || !lossless) return false;
    *out = reinterpret_cast<void *>((uintptr_t)u);
    return true;
}
napi_value make_bigint(napi_env env, const void *p)
{
    napi_value v = nullptr;
    napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)p, &v);
    return v;
}

// depositSetOwners(ctx, world)
napi_value DepositSetOwners(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    const int32_t world = a.i32(1);
    if (!a.ok) BAD_ARGS("th_deposit_set_owners");
    TH_CALL("th_deposit_set_owners", th_deposit_set_owners(c, world));
    return undefined(env);
}

// depositSetHalo(ctx, loAddress | null, hiAddress | null)
napi_value DepositSetHalo(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    void *lo = nullptr, *hi = nullptr;
    if (!a.ok || a.argc < 3 || !bigint_ptr(env, a.argv[1], &lo) || !bigint_ptr(env, a.argv[2], &hi)) BAD_ARGS("th_deposit_set_halo");
    TH_CALL("th_deposit_set_halo", th_deposit_set_halo(c, lo, hi));
    return undefined(env);
}

// depositEmit(ctx, Float32Array(4) th_deposit_uniforms) -> { count, keys: address, colors: address }
napi_value DepositEmit(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_deposit_uniforms u;
    a.uniforms(1, &u);
    if (!a.ok) BAD_ARGS("th_deposit_emit");
    uint64_t count = 0;
    void *keys = nullptr, *colors = nullptr;
    TH_CALL("th_deposit_emit", th_deposit_emit(c, &u, &count, &keys, &colors));
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_double(env, (double)count, &v));
    NAPI_OK(napi_set_named_property(env, o, "count", v));
    NAPI_OK(napi_set_named_property(env, o, "keys", make_bigint(env, keys)));
    NAPI_OK(napi_set_named_property(env, o, "colors", make_bigint(env, colors)));
    return o;
}

// depositMerge(ctx, keysAddress, colorsAddress, count)
napi_value DepositMerge(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    void *keys = nullptr, *colors = nullptr;
    const double count = a.f64(3);
    if (!a.ok || !bigint_ptr(env, a.argv[1], &keys) || !bigint_ptr(env, a.argv[2], &colors) || count < 0) BAD_ARGS("th_deposit_merge");
    TH_CALL("th_deposit_merge", th_deposit_merge(c, keys, colors, (uint64_t)count));
    return undefined(env);
}

// flowDevicePtr(ctx) / stateDevicePtr(ctx, buffer) -> address
napi_value FlowDevicePtr(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_flow_device_ptr");
    void *p = nullptr;
    TH_CALL("th_flow_device_ptr", th_flow_device_ptr(c, &p));
    return make_bigint(env, p);
}
napi_value StateDevicePtr(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    const int32_t buffer = a.i32(1);
    if (!a.ok) BAD_ARGS("th_state_device_ptr");
    void *p = nullptr;
    TH_CALL("th_state_device_ptr", th_state_device_ptr(c, buffer, &p));
    return make_bigint(env, p);
}

// viewEmit(ctx, Float32Array th_render_uniforms) -> { count, keys: address, colors: address }  (row-band shard: the view pass's fragments)
napi_value ViewEmit(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_render_uniforms u;
    a.uniforms(1, &u);
    if (!a.ok) BAD_ARGS("th_view_emit");
    uint64_t count = 0;
    void *keys = nullptr, *colors = nullptr;
    TH_CALL("th_view_emit", th_view_emit(c, &u, &count, &keys, &colors));
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_double(env, (double)count, &v));
    NAPI_OK(napi_set_named_property(env, o, "count", v));
    NAPI_OK(napi_set_named_property(env, o, "keys", make_bigint(env, keys)));
    NAPI_OK(napi_set_named_property(env, o, "colors", make_bigint(env, colors)));
    return o;
}

// viewMerge(ctx, keysAddress, colorsAddress, count)
napi_value ViewMerge(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    void *keys = nullptr, *colors = nullptr;
    const double count = a.f64(3);
    if (!a.ok || !bigint_ptr(env, a.argv[1], &keys) || !bigint_ptr(env, a.argv[2], &colors) || count < 0) BAD_ARGS("th_view_merge");
    TH_CALL("th_view_merge", th_view_merge(c, keys, colors, (uint64_t)count));
    return undefined(env);
}

// drawEmit(ctx, Float32Array th_deposit_uniforms, Float32Array th_render_uniforms) -> { count, keys: address, colors: address }
// (row-band shard: both passes' fragments in one - colors holds 2 x float4 per fragment)
napi_value DrawEmit(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_deposit_uniforms d;
    th_render_uniforms u;
    a.uniforms(1, &d);
    a.uniforms(2, &u);
    if (!a.ok) BAD_ARGS("th_draw_emit");
    uint64_t count = 0;
    void *keys = nullptr, *colors = nullptr;
    TH_CALL("th_draw_emit", th_draw_emit(c, &d, &u, &count, &keys, &colors));
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_double(env, (double)count, &v));
    NAPI_OK(napi_set_named_property(env, o, "count", v));
    NAPI_OK(napi_set_named_property(env, o, "keys", make_bigint(env, keys)));
    NAPI_OK(napi_set_named_property(env, o, "colors", make_bigint(env, colors)));
    return o;
}

// drawMerge(ctx, keysAddress, colorsAddress, count)
napi_value DrawMerge(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    void *keys = nullptr, *colors = nullptr;
    const double count = a.f64(3);
    if (!a.ok || !bigint_ptr(env, a.argv[1], &keys) || !bigint_ptr(env, a.argv[2], &colors) || count < 0) BAD_ARGS("th_draw_merge");
    TH_CALL("th_draw_merge", th_draw_merge(c, keys, colors, (uint64_t)count));
    return undefined(env);
}

// viewDevicePtr(ctx) -> address of the RGBA8 view buffer
napi_value ViewDevicePtr(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_view_device_ptr");
    void *p = nullptr;
    TH_CALL("th_view_device_ptr", th_view_device_ptr(c, &p));
    return make_bigint(env, p);
}

// drawSharded(ctx, Float32Array th_deposit_uniforms, Float32Array th_render_uniforms | null) -> fragments of this rank
// (Tendrils.draw() of a row-band shard, the exchange issued by the library over its communicator: collective)
napi_value DrawSharded(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    th_deposit_uniforms du;
    th_render_uniforms ru;
    a.uniforms(1, &du);
    napi_valuetype t = napi_undefined;
    const bool view = a.argc > 2 && napi_typeof(env, a.argv[2], &t) == napi_ok && t != napi_null && t != napi_undefined;
    if (view) a.uniforms(2, &ru);
    if (!a.ok) BAD_ARGS("th_draw_sharded");
    uint64_t n = 0;
    TH_CALL("th_draw_sharded", th_draw_sharded(c, &du, view ? &ru : nullptr, &n));
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)n, &v));
    return v;
}

// stateGather(ctx, buffer): the whole particle texture of ring buffer `buffer` on every rank (RCCL all-gather; needs commInit)
napi_value StateGather(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    const int32_t buffer = a.i32(1);
    if (!a.ok) BAD_ARGS("th_state_gather");
    TH_CALL("th_state_gather", th_state_gather(c, buffer));
    return undefined(env);
}

// stateGatherPtr(ctx, buffer) -> address of that copy, for a host with its own transport
napi_value StateGatherPtr(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    const int32_t buffer = a.i32(1);
    if (!a.ok) BAD_ARGS("th_state_gather_ptr");
    void *p = nullptr;
    TH_CALL("th_state_gather_ptr", th_state_gather_ptr(c, buffer, &p));
    return make_bigint(env, p);
}

napi_value TimerStart(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_timer_start");
    TH_CALL("th_timer_start", th_timer_start(c));
    return undefined(env);
}

napi_value TimerStop(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_timer_stop");
    float ms = 0;
    TH_CALL("th_timer_stop", th_timer_stop(c, &ms));
    napi_value v;
    NAPI_OK(napi_create_double(env, ms, &v));
    return v;
}

napi_value KernelTiming(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t on = a.i32(1);
    if (!a.ok) BAD_ARGS("th_kernel_timing");
    TH_CALL("th_kernel_timing", th_kernel_timing(c, on));
    return undefined(env);
}

// drawPipeline(ctx, which): TH_DRAW_AUTO / _STREAM / _BINS
napi_value DrawPipeline(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t which = a.i32(1);
    if (!a.ok) BAD_ARGS("th_draw_pipeline");
    TH_CALL("th_draw_pipeline", th_draw_pipeline(c, which));
    return undefined(env);
}

// option(ctx, which[, value]) -> the switch's value (set first when `value` is given): th_option_set / th_option_get
napi_value Option(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t which = a.i32(1);
    if (!a.ok) BAD_ARGS("th_option_get");
    if (a.argc > 2) {
        double value = a.f64(2);
        if (!a.ok) BAD_ARGS("th_option_set");
        TH_CALL("th_option_set", th_option_set(c, which, (int64_t)value));
    }
    int64_t out = 0;
    TH_CALL("th_option_get", th_option_get(c, which, &out));
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)out, &v));
    return v;
}

// drawQuery(ctx) -> {pipeline, fragments, crowdedFragments} of the last draw pass
napi_value DrawQuery(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_draw_query");
    th_draw_info q{};
    TH_CALL("th_draw_query", th_draw_query(c, &q));
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_int32(env, q.pipeline, &v));
    NAPI_OK(napi_set_named_property(env, o, "pipeline", v));
    NAPI_OK(napi_create_double(env, (double)q.fragments, &v));
    NAPI_OK(napi_set_named_property(env, o, "fragments", v));
    NAPI_OK(napi_create_double(env, (double)q.crowded_fragments, &v));
    NAPI_OK(napi_set_named_property(env, o, "crowdedFragments", v));
    NAPI_OK(napi_create_double(env, (double)q.sent_bytes, &v));
    NAPI_OK(napi_set_named_property(env, o, "sentBytes", v));
    NAPI_OK(napi_create_double(env, (double)q.received_bytes, &v));
    NAPI_OK(napi_set_named_property(env, o, "receivedBytes", v));
    return o;
}

// lineWidth(ctx, pass, width): gl.lineWidth before the flow (0) / view (1) pass; lineWidthRange(ctx, lo, hi): what
// ALIASED_LINE_WIDTH_RANGE reports on this context
napi_value LineWidth(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t pass = a.i32(1);
    double w = a.f64(2);
    if (!a.ok) BAD_ARGS("th_line_width");
    TH_CALL("th_line_width", th_line_width(c, pass, (float)w));
    return undefined(env);
}

napi_value LineWidthRange(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    double lo = a.f64(1), hi = a.f64(2);
    if (!a.ok) BAD_ARGS("th_line_width_range");
    TH_CALL("th_line_width_range", th_line_width_range(c, (float)lo, (float)hi));
    return undefined(env);
}

// lineWidthQuery(ctx, pass) -> {width, drawn, range: [lo, hi]}
napi_value LineWidthQuery(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    int32_t pass = a.i32(1);
    if (!a.ok) BAD_ARGS("th_line_width_query");
    float w = 0, d = 0, r[2] = {0, 0};
    TH_CALL("th_line_width_query", th_line_width_query(c, pass, &w, &d, r));
    napi_value o, v, arr;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_double(env, w, &v));
    NAPI_OK(napi_set_named_property(env, o, "width", v));
    NAPI_OK(napi_create_double(env, d, &v));
    NAPI_OK(napi_set_named_property(env, o, "drawn", v));
    NAPI_OK(napi_create_array_with_length(env, 2, &arr));
    for (uint32_t k = 0; k < 2; ++k) {
        NAPI_OK(napi_create_double(env, r[k], &v));
        NAPI_OK(napi_set_element(env, arr, k, v));
    }
    NAPI_OK(napi_set_named_property(env, o, "range", arr));
    return o;
}

// kernelTimingRead(ctx) -> {meanMs, launches}
napi_value KernelTimingRead(napi_env env, napi_callback_info info)
{
    Args a(env, info);
    th_context *c = a.ctx(0);
    if (!a.ok) BAD_ARGS("th_kernel_timing_read");
    float ms = 0;
    int32_t n = 0;
    TH_CALL("th_kernel_timing_read", th_kernel_timing_read(c, &ms, &n));
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_double(env, ms, &v));
    NAPI_OK(napi_set_named_property(env, o, "meanMs", v));
    NAPI_OK(napi_create_int32(env, n, &v));
    NAPI_OK(napi_set_named_property(env, o, "launches", v));
    return o;
}

napi_value Init(napi_env env, napi_value exports)
{
    struct { const char *name; napi_callback fn; } table[] = {
        {"abiVersion", AbiVersion}, {"deviceCount", DeviceCount},
        {"create", Create}, {"destroy", Destroy}, {"setMode", SetMode}, {"setup", Setup}, {"numBuffers", NumBuffers},
        {"uploadState", StateXfer<true>}, {"downloadState", StateXfer<false>},
        {"flowResize", FlowResize}, {"flowUpload", FlowUpload}, {"flowDownload", FlowDownload}, {"flowClear", FlowClear},
        {"targetsUpload", TargetsUpload}, {"targetsDownload", TargetsDownload}, {"targetsClear", TargetsClear},
        {"step", Step}, {"stepN", StepN},
        {"spawnInit", SpawnInit}, {"spawnBall", SpawnBall}, {"spawnSample", SpawnSample},
        {"spawnDirect", SpawnDirect}, {"spawnImageUpload", SpawnImageUpload}, {"spawnImageTriangles", SpawnImageTriangles},
        {"framesResize", FramesResize}, {"framesUpload", FramesUpload}, {"framesRotate", FramesRotate},
        {"opticalFlow", OpticalFlow},
        {"flowDeposit", FlowDeposit}, {"exportLines", ExportLines},
        {"viewDraw", ViewDraw}, {"draw", Draw}, {"viewFill", ViewFill}, {"viewClear", ViewClear}, {"viewDownload", ViewDownload},
        {"viewBuffers", ViewIndexed<th_view_buffers>}, {"viewBind", ViewIndexed<th_view_bind>}, {"viewCopy", ViewIndexed<th_view_copy>}, {"viewStepBuffers", ViewStepBuffers},
        {"colormapUpload", ColormapUpload}, {"exportViewLines", ExportViewLines},
        {"depositSetOwners", DepositSetOwners}, {"depositSetHalo", DepositSetHalo}, {"depositEmit", DepositEmit},
        {"depositMerge", DepositMerge}, {"flowDevicePtr", FlowDevicePtr}, {"stateDevicePtr", StateDevicePtr},
        {"stats", Stats}, {"sync", Sync}, {"timerStart", TimerStart}, {"timerStop", TimerStop},
        {"kernelTiming", KernelTiming}, {"kernelTimingRead", KernelTimingRead}, {"drawPipeline", DrawPipeline}, {"drawQuery", DrawQuery}, {"option", Option},
        {"lineWidth", LineWidth}, {"lineWidthRange", LineWidthRange}, {"lineWidthQuery", LineWidthQuery},
        {"commUniqueId", CommUniqueId},
#ifdef TH_TESTING
        {"commLoopbackId", CommLoopbackId},
#endif
        {"commInit", CommInit}, {"commDestroy", CommDestroy}, {"commQuery", CommQuery},
        {"statsAllreduce", StatsAllreduce}, {"statsGlobal", StatsGlobal},
        {"viewEmit", ViewEmit}, {"viewMerge", ViewMerge}, {"drawEmit", DrawEmit}, {"drawMerge", DrawMerge}, {"viewDevicePtr", ViewDevicePtr},
        {"stateGather", StateGather}, {"stateGatherPtr", StateGatherPtr}, {"drawSharded", DrawSharded},
    };
    for (auto &e : table) {
        napi_value fn;
        if (napi_create_function(env, e.name, NAPI_AUTO_LENGTH, e.fn, nullptr, &fn) != napi_ok) return nullptr;
        if (napi_set_named_property(env, exports, e.name, fn) != napi_ok) return nullptr;
    }
    napi_value v;
    struct { const char *name; int32_t val; } consts[] = {
        {"MODE_EXACT", TH_MODE_EXACT}, {"MODE_FAST", TH_MODE_FAST}, {"STATE_F32", TH_STATE_F32}, {"STATE_F16", TH_STATE_F16},
        {"TARGET_RING", TH_TARGET_RING}, {"TARGET_TARGETS", TH_TARGET_TARGETS}, {"SOURCE_FLOW", TH_SOURCE_FLOW},
        {"SOURCE_IMAGE", TH_SOURCE_IMAGE}, {"DRAW_AUTO", TH_DRAW_AUTO}, {"DRAW_STREAM", TH_DRAW_STREAM}, {"DRAW_BINS", TH_DRAW_BINS},
        {"OPT_BUCKET", TH_OPT_BUCKET}, {"OPT_RESORT_STEPS", TH_OPT_RESORT_STEPS}, {"OPT_REBUCKET_STEPS", TH_OPT_REBUCKET_STEPS},
        {"OPT_FUSE", TH_OPT_FUSE}, {"OPT_GRAPH", TH_OPT_GRAPH}, {"OPT_FORCE_GENERIC", TH_OPT_FORCE_GENERIC},
        {"OPT_DRAW_REUSE", TH_OPT_DRAW_REUSE}, {"OPT_BINS_POOL", TH_OPT_BINS_POOL}, {"OPT_BINS_PAGES", TH_OPT_BINS_PAGES}, {"OPT_ASYNC_SORT", TH_OPT_ASYNC_SORT}, {"OPT_SKIP_UNSEEN", TH_OPT_SKIP_UNSEEN},
#ifdef TH_TESTING
        {"OPT_INJECT_FAILURE", TH_OPT_INJECT_FAILURE},
#endif
    };
    for (auto &e : consts) {
        if (napi_create_int32(env, e.val, &v) != napi_ok) return nullptr;
        if (napi_set_named_property(env, exports, e.name, v) != napi_ok) return nullptr;
    }
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
