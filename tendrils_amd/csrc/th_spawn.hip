// th_spawn.hip - the spawners (src/spawn/**, src/demo.main.js:433-515): init, ball, best-sample from the flow / the particle
// texture / an image, direct image spawn, GeometrySpawner's triangle raster.
#include "th_ctx.hpp"

using namespace thi;

extern "C" {

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

}  // extern "C"
