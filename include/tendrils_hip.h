/*
 * tendrils_hip.h - C ABI of the MI355X-native Tendrils particle integrator.
 *
 * This is the drop-in boundary for the reference's GPGPU update path
 * (keeffEoghan/tendrils).  The reference has no native FFI: the seam it offers
 * is the `Particles` object contract (src/particles.js:43-196) with a swappable
 * "logic" shader, driven by `Tendrils.step()/spawnShader()` (src/index.js:248-272,
 * 432-457) over WebGL FBO ping-pong.  Each entry point below replaces one of
 * those GL-backed operations; the ctypes binding (tendrils_amd/_capi.py) binds exactly these
 * symbols and the N-API shim (tendrils_amd/csrc/th_napi.cc) all of them except th_stream, th_stats_async,
 * th_spawn_image_download and th_slot_order.  Both hosts run row-band-sharded jobs, one process per GPU: the ranks join the
 * library's own communicator (th_comm_unique_id -> th_comm_init; JS: Particles.commUniqueId / commInit) and from then on
 * the path's collectives - the counter all-reduce, the draw()'s exchange, the state gather - are issued by the library over
 * RCCL (th_stats_allreduce, th_draw_sharded, th_state_gather); the host only carries the 128-byte id between its processes.
 * The exchange primitives underneath (th_deposit_emit / _merge / _set_halo / _set_owners, th_flow_device_ptr,
 * th_state_device_ptr; device addresses reach JS as BigInt) stay exported for hosts with a transport of their own - the
 * Python host's torch.distributed path, tendrils_amd/sharding.py:draw_sharded - INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no exceptions across the boundary; every call returns th_status
 *    (0 = TH_OK) and th_last_error() gives the message of the last failure on
 *    the calling thread (gl-shader/gl-fbo throw JS Errors: docs/js/index.js:42 -
 *    the shim turns a non-zero status into a thrown Error).
 *  - host pointers are caller-owned; device memory is library-owned.
 *  - texel layout on the host side is the reference's: row-major RGBA32F,
 *    texel (x, y) at float offset 4*(y*W + x)  (what gl.readPixels returns and
 *    what Particles.spawn uploads, src/particles.js:94-117).
 *  - one HIP stream per context; calls enqueue and return (GL semantics, no
 *    host sync in th_step); th_sync() / downloads synchronise.
 *  - a context is not thread-safe; different contexts are independent.
 */
#ifndef TENDRILS_HIP_H
#define TENDRILS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TH_ABI_VERSION 14

typedef int32_t th_status;
enum {
    TH_OK = 0,
    TH_ERR_INVALID = 1,      /* bad argument / bad state (e.g. step with < 2 buffers) */
    TH_ERR_HIP = 2,          /* a HIP runtime call failed */
    TH_ERR_NO_DEVICE = 3,    /* no usable gfx950 device */
    TH_ERR_UNSUPPORTED = 4
};

/* src/const/inert.glsl:1, src/const/inert.js:2 */
#define TH_INERT (-1000000.0f)

typedef struct th_context th_context;

/* Arithmetic mode of the integrator kernel. */
enum {
    TH_MODE_EXACT = 0,   /* bit-identical to the reference shader's fp32 arithmetic */
    TH_MODE_FAST = 1     /* FMA contraction + approximate rcp/sqrt; tolerance in DESIGN.md */
};

/* Storage format of the state ring.  The host side of the ABI always speaks RGBA32F texels. */
enum {
    TH_STATE_F32 = 0,    /* RGBA32F, the reference's texture format (gl-fbo {float:true}, src/particles.js:84-85) */
    TH_STATE_F16 = 1     /* 8 B per particle: SNORM16 position over [-2,2) + fp16 velocity (config C5; build-defined,
                            the reference has no half path).  Arithmetic stays fp32; only the storage is quantised. */
};

/* Render targets / texture sources, replacing the FBO arguments of
 * Particles.step(update, buffer) (src/particles.js:123-130) and
 * PixelSpawner.buffer (src/spawn/pixels/index.js:38-40). */
enum {
    TH_TARGET_RING = -1,     /* rotate the ring, write buffers[0] (the default path) */
    TH_TARGET_TARGETS = -2,  /* tendrils.targets (src/index.js:105) */
    TH_SOURCE_FLOW = -3,     /* tendrils.flow as spawnData (src/demo.main.js:403-406) */
    TH_SOURCE_IMAGE = -4     /* the spawner's own buffer: an RGBA image in a float texture (th_spawn_image_upload) */
    /* values >= 0 name ring buffer k in its CURRENT order (buffers[k]) */
};

/* new Particles(gl, {shape}) + setup(numBuffers)  (src/particles.js:44-92).
 * A context may hold one row band of a larger state texture (multi-GPU
 * sharding): kernels then use the global coordinates so that `gl_FragCoord`,
 * `uv` and the index `i` (src/logic.frag:46,57-58) match the unsharded run. */
typedef struct th_config {
    int32_t device;         /* HIP device ordinal */
    int32_t width;          /* state texture width  (dataRes.x) */
    int32_t height;         /* rows held by THIS context */
    int32_t global_height;  /* dataRes.y of the whole texture (0 = height) */
    int32_t row0;           /* global row of local row 0 */
    int32_t num_buffers;    /* ring length; Tendrils uses 2 (src/index.js:186) */
    int32_t mode;           /* TH_MODE_* */
    int32_t state_format;   /* TH_STATE_* : storage of the state ring */
} th_config;

/* Uniforms of src/logic.frag:3-34; field names = shader uniform names =
 * Tendrils.state keys (src/index.js:28-66).  time/dt in milliseconds
 * (src/timer.js, src/index.js:67). */
typedef struct th_logic_uniforms {
    float viewSize[2];
    float time, dt;
    float speedLimit, damping;
    float forceWeight, flowWeight, noiseWeight;
    float flowDecay;
    float noiseSpeed, noiseScale;
    float target;
    float varyForce, varyFlow, varyNoise, varyNoiseScale, varyNoiseSpeed, varyTarget;
} th_logic_uniforms;

/* Uniforms of src/optical-flow/index.frag:12-24 (defaults src/optical-flow/index.js:21-29). */
typedef struct th_optical_flow_uniforms {
    float viewSize[2];
    float scaleUV[2];
    float offset, lambda;
    float time, speed, speedLimit;
} th_optical_flow_uniforms;

/* Uniforms of src/spawn/ball/index.frag:3-4 (defaults src/spawn/ball/index.js:7-10). */
typedef struct th_spawn_ball_uniforms {
    float radius, speed;
} th_spawn_ball_uniforms;

/* Uniforms of src/spawn/pixels/frag/head.frag:6-17, best-sample-main.frag:12,
 * flow-sample.frag:3 (host values: src/spawn/pixels/index.js:47-56). */
typedef struct th_spawn_sample_uniforms {
    float spawnSize[2];
    float jitter[2];
    float time, speed, bias;
    float flowDecay;
    float spawnMatrix[9];    /* column-major mat3 */
    int32_t samples;         /* flow-sample 5, data-sample 2 */
    int32_t apply;           /* 0: apply/flow.glsl; 1: apply/identity.glsl over the vignette pass (data-sample.frag);
                                2: apply/color.glsl over the vignette pass (best-sample.frag, index.frag);
                                3: apply/brightest.glsl (bright-sample.frag, GeometrySpawner) */
} th_spawn_sample_uniforms;

/* Uniforms of the flow pass of Tendrils.draw(): src/flow/vert/head.vert:8-12 (viewSize, time, speedLimit). */
typedef struct th_deposit_uniforms {
    float viewSize[2];
    float time;
    float speedLimit;
} th_deposit_uniforms;

/* Build-defined statistics (the reference has none). */
typedef struct th_counters {
    uint64_t particles;      /* texels examined */
    uint64_t live;           /* pos != inert */
    uint64_t nan;            /* any component NaN */
    uint64_t capped;         /* live and |vel| >= speedLimit*(1 - 2^-20) */
    uint64_t respawned;      /* particles (re)spawned into the state ring on this context since th_create: every texel
                                of a th_spawn_ball pass, the accepted candidates of a th_spawn_sample pass */
    double sum_speed;        /* sum of |vel| over live, finite particles */
    double max_speed;        /* max |vel| over live, finite particles */
} th_counters;

/* -- library ------------------------------------------------------------- */
int32_t th_abi_version(void);
const char *th_last_error(void);
th_status th_device_count(int32_t *count);

/* -- lifecycle: new Particles(...).setup(n) / dispose() --------------------- */
th_status th_create(const th_config *cfg, th_context **out);
th_status th_destroy(th_context *ctx);
th_status th_set_mode(th_context *ctx, int32_t mode);
/* Particles.setup(numBuffers) src/particles.js:81-92: grow/shrink the ring. */
th_status th_setup(th_context *ctx, int32_t num_buffers);
th_status th_num_buffers(th_context *ctx, int32_t *out);

/* -- state ring: buffer.color[0].setPixels / readPixels --------------------- */
/* buffer = ring index in current order, or -1 = every buffer (what
 * Particles.spawn does, src/particles.js:115-116).  Sub-rectangle in LOCAL rows. */
th_status th_upload_state(th_context *ctx, int32_t buffer, const float *rgba,
                          int32_t x0, int32_t y0, int32_t w, int32_t h);
th_status th_download_state(th_context *ctx, int32_t buffer, float *rgba,
                            int32_t x0, int32_t y0, int32_t w, int32_t h);

/* -- flow / targets textures (src/index.js:102-105, 207, 231-236, 405) ------ */
th_status th_flow_resize(th_context *ctx, int32_t w, int32_t h);      /* flow.shape = [w,h]; zero-filled */
th_status th_flow_upload(th_context *ctx, const float *rgba);
th_status th_flow_download(th_context *ctx, float *rgba);
th_status th_flow_clear(th_context *ctx);                              /* Tendrils.clearFlow */
th_status th_targets_upload(th_context *ctx, const float *rgba);      /* local rows */
th_status th_targets_download(th_context *ctx, float *rgba);
th_status th_targets_clear(th_context *ctx);

/* -- the hot path: Particles.step(update, buffer) with logic.frag ----------- */
/* target = TH_TARGET_RING: utils.step(buffers) then write buffers[0] from
 * buffers[1] (src/particles.js:128-139); otherwise write `target` from
 * buffers[1] without rotating (src/particles.js:124-126). */
th_status th_step(th_context *ctx, const th_logic_uniforms *u, int32_t target);
/* n consecutive Tendrils.step() calls with a fixed-step timer
 * (time_k = time0 + (k+1)*dt_ms in double, as src/timer.js:28-31 accumulates).
 * Result and ring order are identical to n th_step calls; internally the steps of a
 * particle are fused into one pass per <= 32 steps (2-buffer f32 ring), else replayed
 * from a captured hipGraph.  u->time is ignored, u->dt = (float)dt_ms. */
th_status th_step_n(th_context *ctx, const th_logic_uniforms *u, double time0, double dt_ms, int32_t n);

/* -- respawn passes: Tendrils.spawnShader (src/index.js:432-457) ------------
 * Parity of the hash-driven spawners (th_spawn_ball, th_spawn_sample): the reference draws its random numbers from
 * fract(sin(dot(..)) * 43758.5453) (glsl-random), and GLSL leaves sin() to the implementation - the captured GL's values
 * are not reproducible by arithmetic.  This library fixes ONE sequence (sin evaluated in fp64, rounded once); the HIP path
 * equals its restatement bit for bit and the reference's captures in DISTRIBUTION only (radius, speed, acceptance
 * statistics: the spawn tests).  th_spawn_init, th_spawn_direct and the geometry raster have no such
 * freedom and are bit-exact against the captures. */
th_status th_spawn_init(th_context *ctx, int32_t target);
th_status th_spawn_ball(th_context *ctx, const th_spawn_ball_uniforms *u, int32_t target);
th_status th_spawn_sample(th_context *ctx, const th_spawn_sample_uniforms *u, int32_t source, int32_t target);
/* src/spawn/pixels/index.frag (frag/direct-main.frag:10-21, colour apply over the vignette pass): every particle
 * from its own texel of the spawn data (the image spawner of src/demo.main.js:455-512); u->samples is ignored. */
th_status th_spawn_direct(th_context *ctx, const th_spawn_sample_uniforms *u, int32_t source, int32_t target);
/* PixelSpawner.buffer (src/spawn/pixels/index.js:17,34-36: a float FBO) + setPixels: w x h RGBA float texels. */
th_status th_spawn_image_upload(th_context *ctx, const float *rgba, int32_t w, int32_t h);
/* GeometrySpawner.spawn's draw (src/spawn/geometry/index.js:97-115): resize the spawner's buffer to w x h, clear it,
 * and blend `triangles` triangles (6 floats each: xy of three vertices, clip space before the viewSize scale of
 * src/geom/vert/index.vert) in `color` (src/geom/frag/index.frag). */
th_status th_spawn_image_triangles(th_context *ctx, const float *positions, int32_t triangles, const float viewSize[2],
                                   const float color[4], int32_t w, int32_t h);
th_status th_spawn_image_download(th_context *ctx, float *rgba);           /* read-back (tests) */

/* -- optical flow producer: OpticalFlow (src/optical-flow/index.js:32-71) ---- */
th_status th_frames_resize(th_context *ctx, int32_t w, int32_t h);     /* OpticalFlow.resize */
th_status th_frames_upload(th_context *ctx, const uint8_t *rgba8);     /* setPixels -> buffers[0] */
th_status th_frames_rotate(th_context *ctx);                           /* OpticalFlow.step */
/* one full-screen pass of optical-flow/index.frag alpha-blended into flow
 * (src/demo.main.js:1107-1159; blend func src/index.js:267-268). */
th_status th_optical_flow(th_context *ctx, const th_optical_flow_uniforms *u);

/* -- flow deposit: the flow pass of Tendrils.draw() (src/index.js:278-303) ----- */
/* Renders every particle's (previous -> current) line (buffers[1] -> buffers[0]) into the flow texture as
 * (vel, time, min(|vel|/speedLimit, 1)), alpha-blended in the reference's primitive order, with the context's line width
 * (th_line_width below; 1 unless the host widened the range).  fragments (optional) receives the number of fragments blended;
 * the call synchronises once (the fragment lists are sized from a device count).
 * Needs the whole particle texture on this context (a row-band shard: th_deposit_emit / th_deposit_merge below). */
th_status th_flow_deposit(th_context *ctx, const th_deposit_uniforms *u, uint64_t *fragments);
/* gl.lineWidth (src/index.js:302: flowWidth before the flow pass, :336: lineWidth before the view pass): context state
 * like the GL's, kept per pass because th_draw / th_draw_sharded run both passes in one call (the reference sets the width
 * between them); every later pass of that kind draws its lines with clamp(width, range).  th_line_width_range is what
 * gl.getParameter(ALIASED_LINE_WIDTH_RANGE) reports, chosen by the host: [1, 1] by default - the range of the GL the
 * reference was captured on, so that flowWidth = 5 draws width-1 lines as it does there and every capture of the
 * reference keeps pinning the result - up to [1, TH_MAX_LINE_WIDTH] for the picture of a GL that honours the width (a wide line = the width-1
 * construction with its endpoint diamonds scaled: `width` texels across in the minor direction, OpenGL ES 2.0 3.4.2.1;
 * unpinned - no GL within reach draws one).  width <= 0 or NaN: TH_ERR_INVALID, state unchanged (GL: INVALID_VALUE). */
#define TH_MAX_LINE_WIDTH 64.0f
#define TH_PASS_FLOW 0      /* th_flow_deposit, th_deposit_emit, the flow pass of th_draw / th_draw_sharded */
#define TH_PASS_VIEW 1      /* th_view_draw, th_view_emit, the view pass of th_draw / th_draw_sharded */
th_status th_line_width(th_context *ctx, int32_t pass, float width);
th_status th_line_width_range(th_context *ctx, float lo, float hi);         /* 0 < lo <= 1 <= hi <= TH_MAX_LINE_WIDTH */
/* width = as set, drawn = after the clamp, range[2]; any pointer may be NULL */
th_status th_line_width_query(th_context *ctx, int32_t pass, float *width, float *drawn, float *range);

/* Trail export (build-defined; the reference never reads its lines back): the line list draw() hands to GL - same
 * vertex stream, pairing and frame choice as the flow pass (src/state/state-at-frame.glsl:12-22, read by both
 * src/flow/vert/main.vert and src/render/index.vert) - 12 floats per line in stream order: p0.xy, p1.xy (clip space
 * = state.xy*viewSize), then the two vertices' (vel.x, vel.y, time, min(|vel|/speedLimit, 1)).  Lines with an inert
 * vertex or zero length are left out.  lines == NULL queries the count; capacity is in lines. */
th_status th_export_lines(th_context *ctx, const th_deposit_uniforms *u, float *lines, uint64_t capacity, uint64_t *count);

/* Row-band shards (multi-GPU): the deposit in two steps around one exchange (tendrils_amd/sharding.py).
 *  th_deposit_set_owners: the number of ranks that own flow texels (contiguous ranges of ceil(texels / world) texels,
 *    rank by rank; default 1).
 *  th_deposit_emit: rasterise THIS context's lines; key = owner rank << 56 | flow texel << 32 | global stream index
 *    of the line; the fragments are parted by owner (one stable radix pass), every part in the band's stream order;
 *    *keys_dev = uint64[count], *colors_dev = float4[count] (device, owned by the context, valid until the next
 *    deposit call on it).  The caller sends every part to its owner (all-to-all).
 *  th_deposit_merge: blend the fragments received for the texels this rank owns - the parts of the source bands one
 *    after the other, each still in its band's stream order - into this context's flow texture, in (texel, stream
 *    index) order: a stable sort by texel, then the bands of a texel are merged by stream index as they are blended.
 *    For the owned texels the result is the unsharded deposit's, bit for bit.  The owners' texel ranges are then
 *    all-gathered into every rank's flow (th_flow_device_ptr). */
th_status th_deposit_set_owners(th_context *ctx, int32_t world);
th_status th_deposit_emit(th_context *ctx, const th_deposit_uniforms *u, uint64_t *count, void **keys_dev, void **colors_dev);
th_status th_deposit_merge(th_context *ctx, const void *keys_dev, const void *colors_dev, uint64_t count);
/* For texture heights where the fp32 row lookup of the vertex stream (src/state/state-at-frame.glsl:12-22) lands one
 * row beside the line's own row, a band needs its neighbours' edge rows: lo_dev = row row0-1, hi_dev = row
 * row0+rows, each `width` RGBA32F texels of buffers[0] followed by `width` texels of buffers[1] (device memory owned
 * by the caller, read by the next th_deposit_emit; NULL = none).  Without them such a lookup fails the emit. */
th_status th_deposit_set_halo(th_context *ctx, const void *lo_dev, const void *hi_dev);
th_status th_flow_device_ptr(th_context *ctx, void **dptr);
/* Best-sample spawning from the PARTICLE texture (spawnData = particles.buffers[k], src/demo.main.js:433-441) reads arbitrary
 * particles: on a row-band shard th_spawn_sample / th_spawn_direct read them from a copy of the WHOLE texture
 * (width x global_height RGBA32F) of ring buffer `buffer`, valid until that buffer is written again:
 *  th_state_gather    : every rank, collectively (needs th_comm_init; balanced bands): all-gather over RCCL on the context's stream
 *  th_state_gather_ptr: the copy's device address, for a host that fills it by its own means (another transport, several
 *                       contexts on one device). */
th_status th_state_gather(th_context *ctx, int32_t buffer);
th_status th_state_gather_ptr(th_context *ctx, int32_t buffer, void **dptr);

/* -- statistics, sync, interop ---------------------------------------------- */
th_status th_stats(th_context *ctx, float speed_limit, th_counters *out);   /* of buffers[0]; synchronises */
/* enqueue the reduction only; result lands (as th_counters) at the returned device pointer */
th_status th_stats_async(th_context *ctx, float speed_limit, void **device_counters);
th_status th_sync(th_context *ctx);

/* -- one process per GPU (BASELINE.json config 4): row-band shards, the flow texture replicated, and ONE
 * collective on the integrator's path - the statistics block reduced over the ranks (counts and sum_speed added,
 * max_speed maximised).  The reference has no counterpart (one WebGL context); this is the RCCL all-reduce over xGMI
 * the north star names, issued by the library on the context's own stream so that any host - a Node process per GPU
 * through th_napi.cc, the Python host - needs no transport of its own beyond handing rank 0's id to the other ranks
 * (a file, a socket, an environment variable: 128 bytes).  librccl is bound at run time (the copy the process already
 * holds, else the system's; TH_RCCL_LIB names another); a single-GPU host never loads it.
 *  th_comm_unique_id : rank 0 - a fresh id (ncclGetUniqueId)
 *  th_comm_init      : every rank, collectively - the context joins the communicator as `rank` of `world`
 *  th_stats_allreduce: the block th_stats_async filled, reduced in place on the context's stream (no host sync; a
 *                      context without a communicator is a world of one: nothing to do)
 *  th_stats_global   : th_stats_async + th_stats_allreduce + download; synchronises */
#define TH_COMM_ID_BYTES 128
typedef struct th_comm_info {
    int32_t active;          /* the context holds a communicator: 1 = RCCL, 2 = in-process (th_comm_loopback_id); 0 = none */
    int32_t rank, world;
    int32_t rccl_version;    /* ncclGetVersion (0: librccl not loadable) */
} th_comm_info;
th_status th_comm_unique_id(void *id_out /* TH_COMM_ID_BYTES */);
#ifdef TH_TESTING
/* TEST BUILDS ONLY (make TESTING=1, the default of tendrils_amd/csrc/Makefile and what __graft_entry__.build() makes: the
 * suites need it; `make release` leaves it - th_loopback.hip, this entry point, TH_OPT_INJECT_FAILURE - out of the library).
 * An id of an IN-PROCESS world instead: the ranks are contexts of one process (on one device or several), each driven by a
 * host thread of its own, and the exchanges are device-to-device copies around a host-side rendezvous (th_loopback.hip).
 * Everything above the byte transport - which fragments go to which owner, bands, owned ranges, the agreement on failures
 * - is the code an RCCL job runs; this is how it is exercised with more ranks than a box has GPUs.  th_comm_init tells the
 * two kinds of id apart by itself.  A collective that the other ranks do not join within TH_LOOPBACK_TIMEOUT_MS
 * (default 120 000) fails instead of hanging. */
th_status th_comm_loopback_id(void *id_out /* TH_COMM_ID_BYTES */);
#endif
th_status th_comm_init(th_context *ctx, const void *id /* TH_COMM_ID_BYTES */, int32_t rank, int32_t world);
th_status th_comm_destroy(th_context *ctx);
th_status th_comm_query(th_context *ctx, th_comm_info *out);
th_status th_stats_allreduce(th_context *ctx);
th_status th_stats_global(th_context *ctx, float speed_limit, th_counters *out);
th_status th_stream(th_context *ctx, void **hip_stream);               /* hipStream_t of the context */
/* Device address of ring buffer `buffer` (RGBA32F / packed texels in texel order; valid until the ring rotates or re-sorts).
 * The call itself invalidates the line geometry th_view_draw would reuse from the last th_flow_deposit; a host that WRITES
 * through the address later must hand out the address again (or call any state entry point) before the next th_view_draw. */
th_status th_state_device_ptr(th_context *ctx, int32_t buffer, void **dptr);
/* HIP-event timing on the context's own stream (for bench.py / profilers). */
th_status th_timer_start(th_context *ctx);
th_status th_timer_stop(th_context *ctx, float *elapsed_ms);           /* synchronises */
/* Per-launch timing of the integrator kernel alone: while enabled, every th_step records a HIP
 * event pair around its logic-kernel launch; _read synchronises, returns the mean duration and the
 * number of launches since the last read, and resets. */
th_status th_kernel_timing(th_context *ctx, int32_t enable);
th_status th_kernel_timing_read(th_context *ctx, float *mean_ms, int32_t *launches);
/* The context's current texture shapes, for bindings that must size host arrays: state (this context's band),
 * flow / view, frames (0 x 0 before th_frames_resize). */
typedef struct th_shapes_info {
    int32_t state_w, state_h, flow_w, flow_h, frames_w, frames_h;
} th_shapes_info;
th_status th_shapes(th_context *ctx, th_shapes_info *out);

/* ---- view pass of Tendrils.draw() (src/index.js:315-337): the particles' lines with the colours of
 * src/render/index.vert:58-100, blended in primitive order into an RGBA8 image of the drawing buffer's size
 * (= the flow texture's shape, viewRes).  Same rasteriser as the flow pass (the reference draws both with gl.LINES;
 * captures: a context without multisampling, lineWidth 1).  sinTerm = sin(time*flowDecay), evaluated by the host: a
 * uniform-only expression whose value GLSL leaves to the implementation. */
typedef struct th_render_uniforms {
    float viewSize[2];
    float time, speedLimit, flowDecay, speedAlpha, colorMapAlpha, sinTerm;
    float baseColor[4], flowColor[4];
} th_render_uniforms;
th_status th_view_draw(th_context *ctx, const th_render_uniforms *u, uint64_t *fragments);
/* Both passes of Tendrils.draw() in one call (src/index.js:278-337: the flow pass, then the view pass): same results as
 * th_flow_deposit(u) followed by th_view_draw(r), with the lines rasterised and the fragments sorted once when both passes
 * draw with the same width.  u and r must agree in viewSize, time and speedLimit. */
th_status th_draw(th_context *ctx, const th_deposit_uniforms *u, const th_render_uniforms *r, uint64_t *fragments);
/* Tendrils.drawFill / drawFade (src/index.js:342-356): a full-screen colour blended SRC_ALPHA / ONE_MINUS_SRC_ALPHA */
th_status th_view_fill(th_context *ctx, const float rgba[4]);
th_status th_view_clear(th_context *ctx);                              /* gl.clear(COLOR_BUFFER_BIT), clear colour 0 */
th_status th_view_download(th_context *ctx, uint8_t *rgba8);           /* flow-shape RGBA8, row-major */
/* Tendrils.buffers (src/index.js:66-68 `numBuffers`, 172-184 setupBuffers, 359-391 drawBuffer / copyBuffer / stepBuffers): a
 * ring of off-screen RGBA8 view images of the drawing buffer's shape beside the screen image.  Every view entry point above
 * and below (th_view_draw, the view pass of th_draw / th_draw_sharded, th_view_fill / _clear / _download / _device_ptr) works
 * on the BOUND image: the screen unless th_view_bind chose a buffer - Tendrils.draw() binds buffers[0] when there is one
 * (src/index.js:318-325), drawBuffer() the screen.  A buffer keeps being bound when the ring rotates under it (GL binds the
 * object, not the position); a bound buffer that th_view_buffers removes leaves the screen bound.  A resize empties them all.
 *   th_view_buffers       setupBuffers(count): buffers added (transparent black) or removed at the ring's end
 *   th_view_bind          gl.bindFramebuffer: index -1 = the screen, k = buffers[k] as the ring stands now
 *   th_view_copy          copyBuffer(index): buffers[index] through copy.frag - texel for texel - blended SRC_ALPHA /
 *                         ONE_MINUS_SRC_ALPHA into the bound image; an index beyond the ring does nothing, as there
 *   th_view_step_buffers  stepBuffers(): buffers.unshift(buffers.pop()) when there are two or more */
th_status th_view_buffers(th_context *ctx, int32_t count);
th_status th_view_bind(th_context *ctx, int32_t index);
th_status th_view_copy(th_context *ctx, int32_t index);
th_status th_view_step_buffers(th_context *ctx);
/* tendrils.colorMap (src/index.js:94-96: a 1x1 float FBO unless given): RGBA32F, NEAREST, CLAMP_TO_EDGE */
th_status th_colormap_upload(th_context *ctx, const float *rgba, int32_t w, int32_t h);
/* th_export_lines with the view pass's vertex colours in place of the flow varyings */
th_status th_export_view_lines(th_context *ctx, const th_render_uniforms *u, float *lines, uint64_t capacity, uint64_t *count);
/* The view pass of a row-band shard (src/index.js:315-337), the same way: th_view_emit = this band's fragments with the render
 * shader's colours (src/render/index.vert:58-100), parted by owner; th_view_merge = the owner's fragments blended into
 * this context's view buffer in (texel, stream index) order - for the owned texels the unsharded view pass byte for byte;
 * th_view_device_ptr = the RGBA8 view buffer (flow shape) whose owned ranges the ranks then gather. */
/* Tendrils.draw() of a row-band shard with the exchange issued by the LIBRARY over its own communicator (every rank,
 * collectively; needs th_comm_init; balanced bands, <= 32 ranks): edge rows to the neighbours, then emit -> fragment
 * all-to-all by texel owner -> merge -> all-gather of the owned ranges, all on the context's stream (ncclSend / ncclRecv
 * groups, ncclAllGather) - once for both passes (th_draw_emit / th_draw_merge below) when they draw with the same line
 * width, else pass by pass.  r = NULL: the flow pass only.  Same results as the primitives driven by a host. */
th_status th_draw_sharded(th_context *ctx, const th_deposit_uniforms *u, const th_render_uniforms *r, uint64_t *fragments);
th_status th_view_emit(th_context *ctx, const th_render_uniforms *u, uint64_t *count, void **keys_dev, void **colors_dev);
th_status th_view_merge(th_context *ctx, const void *keys_dev, const void *colors_dev, uint64_t count);
th_status th_view_device_ptr(th_context *ctx, void **dptr);
/* Both passes of a row-band shard's draw() over ONE rasterisation and ONE exchange (src/index.js:278-337; what th_draw is to
 * th_flow_deposit + th_view_draw): th_draw_emit = this band's fragments, every one with the flow pass's varying and the view
 * pass's colour side by side (colors: 2 x float4 per fragment), parted by owner; th_draw_merge = the owner's fragments
 * blended into the flow texture and the view buffer.  Needs both passes to draw with the same line width (else:
 * TH_ERR_INVALID, use the per-pass primitives); u and r must agree in viewSize, time and speedLimit.  th_draw_sharded goes
 * this way whenever it can. */
th_status th_draw_emit(th_context *ctx, const th_deposit_uniforms *u, const th_render_uniforms *r, uint64_t *count, void **keys_dev, void **colors_dev);
th_status th_draw_merge(th_context *ctx, const void *keys_dev, const void *colors_dev, uint64_t count);

/* Slot layout of the ring (build-defined, invisible in every result): how many ring buffers are held in a
 * tile-sorted slot order, integrator passes since the last sort, and the flow taps since then that left the
 * LDS-staged window of their workgroup (served by the global gather instead).  Synchronises. */
typedef struct th_slot_order_info {
    int32_t sorted_buffers;
    int32_t steps_since_sort;
    uint64_t window_misses;
    uint64_t sorts;              /* sorts so far in this context's life */
} th_slot_order_info;
th_status th_slot_order(th_context *ctx, th_slot_order_info *out);

/* Which of the two draw() pipelines th_flow_deposit / th_view_draw / th_draw run through (build-defined, invisible in every
 * result: both reproduce GL's primitive order bit for bit).  TH_DRAW_STREAM: fragments produced in stream order from
 * particles in texel order, stable radix sort by texel (th_deposit.hip).  TH_DRAW_BINS: particles walked in whatever slot
 * order the ring is held in, fragments bucketed by 16 x 16-texel bin of the target and put in order inside each bin
 * (th_bins.hip) - what lets a step() + draw() frame loop (src/demo.main.js:1082) stay on tile-sorted slots.
 * TH_DRAW_AUTO (default): bins wherever the integrator steps over sorted slots. */
enum { TH_DRAW_AUTO = -1, TH_DRAW_STREAM = 0, TH_DRAW_BINS = 1 };
th_status th_draw_pipeline(th_context *ctx, int32_t which);
/* What the last draw pass did: the pipeline it took, its fragments, and (binned pipeline) how many of them fell into bins of
 * more than 4096 fragments - how crowded the target is (most fragments in texels of hundreds and thousands once the wake of a
 * long-running loop has drawn the particles together). */
typedef struct th_draw_info {
    int32_t pipeline;            /* TH_DRAW_STREAM / TH_DRAW_BINS */
    int32_t reserved;
    uint64_t fragments, crowded_fragments;
    uint64_t sent_bytes, received_bytes;   /* th_draw_sharded: the payload this rank gave to / took from the other ranks in the whole
                                            * call - edge rows, counts, fragments (or bins) for other owners, the owned ranges of the
                                            * target(s), each counted once; 0 after a local draw and in a world of one */
} th_draw_info;
th_status th_draw_query(th_context *ctx, th_draw_info *out);

/* Per-context switches between equivalent paths (build-defined; no switch changes a result - the parity suites rerun under
 * each, tests/conftest.py).  A context starts from the environment variables of the same names, read by th_create
 * (TH_BUCKET, TH_RESORT_STEPS, TH_REBUCKET_STEPS, TH_FUSE, TH_GRAPH, TH_FORCE_GENERIC, TH_DRAW_REUSE, TH_BINS_POOL, TH_BINS_PAGES, TH_ASYNC_SORT, TH_SKIP_UNSEEN;
 * TH_DRAW=stream|bins sets what TH_DRAW_AUTO means).
 *   TH_OPT_BUCKET          tile-sorted slot order never (0) / always (1) / when it pays (-1, default)
 *   TH_OPT_RESORT_STEPS    re-sort period of single-step launches (default 64)
 *   TH_OPT_REBUCKET_STEPS  ... of fused th_step_n launches (default 256)
 *   TH_OPT_FUSE            temporal fusion in th_step_n (default 1)
 *   TH_OPT_GRAPH           captured hipGraphs in th_step_n where it does not fuse (default 1)
 *   TH_OPT_FORCE_GENERIC   every step through the reference-order kernel (default 0)
 *   TH_OPT_DRAW_REUSE      the stream-ordered view pass reuses the flow pass's rasterisation and sort (default 1)
 *   TH_OPT_SKIP_UNSEEN     (TH_SKIP_UNSEEN) a single step that a draw() follows notes per 64 slots whether any of their lines may touch
 *                          the view, and the binned draw() skips the blocks of slots of which none may (default 1)
 *   TH_OPT_ASYNC_SORT      (TH_ASYNC_SORT) while draws are going on, the single steps' re-sort runs beside the draw that follows
 *                          a step - a plain move on the draw's side stream, taken up by the next step - instead of inside two
 *                          steps of every TH_OPT_RESORT_STEPS (default 1)
 *   TH_OPT_BINS_POOL       first size, in pages, of the binned pipeline's page pool (default 0: by the target's size)
 *   TH_OPT_BINS_PAGES      pages one list of a bin of the binned pipeline can grow to at first (default 0: 128 - half a million
 *                          fragments per 16 x 16-texel bin; a bin that outgrows its lists gets a table four times as wide and the
 *                          pass is repeated, up to 4096); negative: that many and never more (the draw then goes to the
 *                          stream-ordered pipeline - tests)
 *   TH_OPT_INJECT_FAILURE  (TH_TESTING builds only - a release build answers "unknown option"; the one switch that DOES change what a
 *                          call returns) the next th_draw_sharded of THIS context fails on its own at stage 1 (its edge rows; packed rings), 2
 *                          (rasterising its lines) or 3 (making room for what it owns); 4: its binned pass gives up (every rank takes the stream-ordered pass, the draw succeeds);
 *                          the switch resets itself.  What is
 *                          tested: every other rank of the job returns an error too instead of waiting in a collective */
enum { TH_OPT_BUCKET = 0, TH_OPT_RESORT_STEPS = 1, TH_OPT_REBUCKET_STEPS = 2, TH_OPT_FUSE = 3, TH_OPT_GRAPH = 4,
       TH_OPT_FORCE_GENERIC = 5, TH_OPT_DRAW_REUSE = 6, TH_OPT_BINS_POOL = 7,
#ifdef TH_TESTING
       TH_OPT_INJECT_FAILURE = 8,
#endif
       TH_OPT_BINS_PAGES = 9, TH_OPT_ASYNC_SORT = 10, TH_OPT_SKIP_UNSEEN = 11 };
th_status th_option_set(th_context *ctx, int32_t option, int64_t value);
th_status th_option_get(th_context *ctx, int32_t option, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* TENDRILS_HIP_H */
