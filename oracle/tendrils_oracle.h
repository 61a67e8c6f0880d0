/*
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Interface of the CPU restatement
 * (oracle/tendrils_oracle.c) of the reference's particle path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef TENDRILS_ORACLE_H
#define TENDRILS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/const/inert.glsl:1, src/const/inert.js:2 */
#define TO_INERT (-1000000.0f)

/* Uniform set of src/logic.frag:3-34 (names as in the shader / Tendrils.state,
 * src/index.js:28-66).  data_w/data_h = `dataRes` of the WHOLE state texture. */
typedef struct to_logic_uniforms {
    int32_t data_w, data_h;
    float viewSize[2];
    float time, dt;
    float speedLimit, damping;
    float forceWeight, flowWeight, noiseWeight;
    float flowDecay;
    float noiseSpeed, noiseScale;
    float target;
    float varyForce, varyFlow, varyNoise, varyNoiseScale, varyNoiseSpeed, varyTarget;
} to_logic_uniforms;

float to_snoise3(float x, float y, float z);

void to_logic_step(const to_logic_uniforms *u, const float *in, float *out, int y0, int rows,
                   const float *flow, int fw, int fh, const float *targets);

void to_spawn_init(float *out, size_t texels);

/* Uniform set of src/optical-flow/index.frag:12-24 (defaults src/optical-flow/index.js:21-29). */
typedef struct to_optical_flow_uniforms {
    float viewSize[2];
    float scaleUV[2];
    float offset, lambda;
    float time, speed, speedLimit;
} to_optical_flow_uniforms;

/* One blended full-screen pass of src/optical-flow/index.frag into `flow`
 * (out_w x out_h RGBA32F, updated in place: SRC_ALPHA / ONE_MINUS_SRC_ALPHA,
 * src/index.js:267-268).  view/last: RGBA8 frames fr_w x fr_h. blend=0 writes
 * the shader output unblended. */
void to_optical_flow(const to_optical_flow_uniforms *u, const uint8_t *view, const uint8_t *last,
                     int fr_w, int fr_h, float *flow, int out_w, int out_h, int blend);

/* glsl-random 0.0.5 hash with an explicit sine so that it is reproducible:
 * sin evaluated in double precision and rounded to fp32. */
float to_random(float cox, float coy);

typedef struct to_spawn_ball_uniforms { float radius, speed; } to_spawn_ball_uniforms;

/* src/spawn/ball/index.frag:11-19 over rows [y0, y0+rows) of a w-wide texture. */
void to_spawn_ball(const to_spawn_ball_uniforms *u, float *out, int w, int y0, int rows);

/* Uniform set of src/spawn/pixels/frag/head.frag:6-17 + best-sample-main.frag:12
 * (+ flowDecay for flow-sample.frag:3). */
typedef struct to_spawn_sample_uniforms {
    int32_t data_w, data_h;
    float spawnSize[2];
    float jitter[2];
    float time, speed, bias;
    float flowDecay;
    float spawnMatrix[9];   /* column-major mat3 as uploaded by uniformMatrix3fv */
    int32_t samples;        /* 5 = flow-sample.frag:8, 2 = data-sample.frag:10 */
    int32_t apply;          /* 0 = apply/flow.glsl, 1 = identity over the vignette pass (data-sample.frag),
                               2 = apply/color.glsl over the vignette pass (best-sample.frag, index.frag),
                               3 = apply/brightest.glsl (bright-sample.frag: GeometrySpawner) */
} to_spawn_sample_uniforms;

/* src/spawn/pixels/frag/best-sample-main.frag:21-46 over rows [y0,y0+rows). */
void to_spawn_sample(const to_spawn_sample_uniforms *u, const float *particles, float *out,
                     int y0, int rows, const float *spawn_data, int sw, int sh);

/* src/spawn/pixels/index.frag (frag/direct-main.frag:10-21): every particle from its own texel of spawn_data. */
void to_spawn_direct(const to_spawn_sample_uniforms *u, float *out, int y0, int rows,
                     const float *spawn_data, int sw, int sh);

/* GeometrySpawner's draw (src/spawn/geometry/index.js:97-115): ntri triangles (6 floats each) blended into img. */
void to_triangles(const float *positions, int ntri, const float *view_size, const float *color,
                  float *img, int w, int h);

/* Flow deposit: the particle lines of Tendrils.draw() (src/index.js:278-303) rendered into the flow FBO with the
 * flow shader (src/flow/index.vert -> vert/main.vert:10-17, apply/state.glsl:5-16; index.frag).
 * data_w/data_h = particle texture shape; the vertex stream is Particles.generateLUT(geomShape = [w, 2h])
 * (src/particles.js:171-190, src/index.js:195-197) drawn as gl.LINES.  lineWidth = the width the GL draws with, i.e.
 * gl.lineWidth(flowWidth / lineWidth) (src/index.js:302,336) AFTER the clamp to ALIASED_LINE_WIDTH_RANGE - [1, 1] on the GL
 * every fixture was captured on, so 1 wherever a capture pins the result (0 is taken as 1). */
typedef struct to_deposit_uniforms {
    int32_t data_w, data_h;
    float viewSize[2];
    float time, speedLimit;
    float lineWidth;
} to_deposit_uniforms;

/* Blends every line in stream order into flow (fw x fh RGBA32F); returns the number of fragments.
 * coverage (optional, fw*fh int32) receives the fragment count per texel. */
long to_flow_deposit(const to_deposit_uniforms *u, const float *current, const float *previous,
                     float *flow, int fw, int fh, int32_t *coverage);

/* Uniforms of the view pass (src/render/index.vert:10-24; defaults src/index.js:58-65); sinTerm = sin(time*flowDecay)
 * evaluated by the caller (implementation-defined in GLSL). */
typedef struct to_render_uniforms {
    float speedLimit, flowDecay, speedAlpha, colorMapAlpha, sinTerm;
    float baseColor[4], flowColor[4];
} to_render_uniforms;

/* The view pass of draw(): the same lines with the render shader's colours, blended in stream order into an RGBA8
 * image (vw x vh, the drawing buffer); colormap: cw x ch RGBA32F or NULL (= the 1x1 zero texture).  Returns the
 * number of fragments. */
long to_view_render(const to_deposit_uniforms *u, const to_render_uniforms *r, const float *colormap, int cw, int ch,
                    const float *current, const float *previous, uint8_t *view, int vw, int vh);
void to_view_fill(uint8_t *view, int vw, int vh, const float *color);
long to_export_view_lines(const to_deposit_uniforms *u, const to_render_uniforms *r, const float *colormap, int cw, int ch,
                          const float *current, const float *previous, float *out, long capacity);

/* Trail export: the (previous -> current) line list of draw(), 12 floats per line (see the .c file). */
long to_export_lines(const to_deposit_uniforms *u, const float *current, const float *previous, float *out, long capacity);

#ifdef __cplusplus
}
#endif
#endif
