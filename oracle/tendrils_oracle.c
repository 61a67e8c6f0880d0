/*
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, strict IEEE fp32, no FMA contraction) of the
 * reference's GPGPU particle path, used only as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (tendrils_amd/) never links, loads or calls anything in this file.
 *
 * Parity status: PINNED.  Every function below is checked against golden
 * vectors captured from the reference's own compiled shaders executed in this
 * build container (oracle/gen_fixtures.py -> tests/golden/), see
 * tests/test_oracle_golden.py.  to_logic_step is bit-exact against those
 * captures; tolerances for the other passes are stated in the tests.
 *
 * Build: oracle/Makefile (-O2 -ffp-contract=off, no -ffast-math: every
 * `a*b+c` below is two correctly rounded fp32 operations, which is what the
 * reference's shader compiler emitted in the captures).
 *
 * Each function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "tendrils_oracle.h"

/* ------------------------------------------------------------------------- */
/* glsl-noise 0.0.0 simplex/3d (Ashima/McEwan `snoise(vec3)`), required by    */
/* src/logic.frag:36; source is not vendored under src/, the compiled text    */
/* is inlined in docs/js/index.js:56 (shader lines 48-138).  Algorithm        */
/* restated per component; association order as in the GLSL expressions.      */
/* ------------------------------------------------------------------------- */

static inline float mod289f(float x)
{
    /* mod289: x - floor(x * (1.0 / 289.0)) * 289.0 */
    const float inv289 = 1.0f / 289.0f;
    return x - floorf(x * inv289) * 289.0f;
}

static inline float permutef(float x)
{
    /* permute: mod289(((x*34.0)+1.0)*x) */
    return mod289f(((x * 34.0f) + 1.0f) * x);
}

static inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

static inline float dot3f(float ax, float ay, float az, float bx, float by, float bz)
{
    return ax * bx + ay * by + az * bz;
}

float to_snoise3(float vx, float vy, float vz)
{
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;
    const float n_ = 0.142857142857f;
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;

    /* first corner */
    float s = dot3f(vx, vy, vz, Cy, Cy, Cy);
    float ix = floorf(vx + s), iy = floorf(vy + s), iz = floorf(vz + s);
    float t = dot3f(ix, iy, iz, Cx, Cx, Cx);
    float x0x = vx - ix + t, x0y = vy - iy + t, x0z = vz - iz + t;

    /* other corners */
    float gx = stepf(x0y, x0x), gy = stepf(x0z, x0y), gz = stepf(x0x, x0z);
    float lx = 1.0f - gx, ly = 1.0f - gy, lz = 1.0f - gz;
    float i1x = fminf(gx, lz), i1y = fminf(gy, lx), i1z = fminf(gz, ly);
    float i2x = fmaxf(gx, lz), i2y = fmaxf(gy, lx), i2z = fmaxf(gz, ly);

    float cx[4][3]; /* offsets from each simplex corner */
    cx[0][0] = x0x;             cx[0][1] = x0y;             cx[0][2] = x0z;
    cx[1][0] = x0x - i1x + Cx;  cx[1][1] = x0y - i1y + Cx;  cx[1][2] = x0z - i1z + Cx;
    cx[2][0] = x0x - i2x + Cy;  cx[2][1] = x0y - i2y + Cy;  cx[2][2] = x0z - i2z + Cy;
    cx[3][0] = x0x - 0.5f;      cx[3][1] = x0y - 0.5f;      cx[3][2] = x0z - 0.5f;

    /* permutations */
    ix = mod289f(ix); iy = mod289f(iy); iz = mod289f(iz);
    const float oz[4] = {0.0f, i1z, i2z, 1.0f};
    const float oy[4] = {0.0f, i1y, i2y, 1.0f};
    const float ox[4] = {0.0f, i1x, i2x, 1.0f};

    float acc = 0.0f, mm[4], gd[4];
    for (int k = 0; k < 4; ++k) {
        float p = permutef(permutef(permutef(iz + oz[k]) + iy + oy[k]) + ix + ox[k]);

        /* gradient: 7x7 points over a square, mapped onto an octahedron */
        float j = p - 49.0f * floorf(p * nsz * nsz);
        float x_ = floorf(j * nsz);
        float y_ = floorf(j - 7.0f * x_);
        float x = x_ * nsx + nsy;
        float y = y_ * nsx + nsy;
        float h = 1.0f - fabsf(x) - fabsf(y);
        float sx = floorf(x) * 2.0f + 1.0f;
        float sy = floorf(y) * 2.0f + 1.0f;
        float sh = -stepf(h, 0.0f);
        float px = x + sx * sh, py = y + sy * sh, pz = h;

        /* normalise (taylorInvSqrt) */
        float norm = 1.79284291400159f - 0.85373472095314f * dot3f(px, py, pz, px, py, pz);
        px *= norm; py *= norm; pz *= norm;

        float m = fmaxf(0.6f - dot3f(cx[k][0], cx[k][1], cx[k][2], cx[k][0], cx[k][1], cx[k][2]), 0.0f);
        m = m * m;
        mm[k] = m * m;
        gd[k] = dot3f(px, py, pz, cx[k][0], cx[k][1], cx[k][2]);
    }
    acc = mm[0] * gd[0] + mm[1] * gd[1] + mm[2] * gd[2] + mm[3] * gd[3];
    return 42.0f * acc;
}

/* ------------------------------------------------------------------------- */
/* src/logic.frag                                                             */
/* ------------------------------------------------------------------------- */

/* src/logic.frag:41-43 */
static inline float varyf(float base, float offset, float variance)
{
    return base + (offset * variance * base);
}

/* NEAREST + CLAMP_TO_EDGE texel fetch (gl-fbo colour textures: docs/js/index.js:42) */
static inline int nearest_texel(float u, int n)
{
    float f = floorf(u * (float)n);
    if (!(f > 0.0f)) return 0;          /* also catches NaN */
    if (f > (float)(n - 1)) return n - 1;
    return (int)f;
}

/* src/flow/get.glsl:3-5 */
static inline void flow_get(const float *texel, float time, float decay, float *fx, float *fy)
{
    float k = fmaxf(0.0f, 1.0f - ((time - texel[2]) * decay));
    *fx = texel[0] * k;
    *fy = texel[1] * k;
}

/* One fragment of src/logic.frag:45-101 at global texel (x, y). */
static void logic_texel(const to_logic_uniforms *u, int x, int y, const float *in,
                        const float *flow, int fw, int fh, const float *target_texel, float *out)
{
    const float W = (float)u->data_w, H = (float)u->data_h;
    float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;    /* gl_FragCoord.xy */
    float uvx = fcx / W, uvy = fcy / H;                      /* :46 */
    float posx = in[0], posy = in[1], velx = in[2], vely = in[3];
    float npx = posx, npy = posy, nvx = velx, nvy = vely;

    /* :52  `pos != inert` on a vec2 is true when ANY component differs */
    if (posx != TO_INERT || posy != TO_INERT) {
        float i = (fcx + (fcy * W)) / (W * H);               /* :57-58 */

        float nscale = varyf(u->noiseScale, i, u->varyNoiseScale);
        float noisex = posx * nscale, noisey = posy * nscale;    /* :62 */
        float noiseTime = u->time * varyf(u->noiseSpeed, i, u->varyNoiseSpeed); /* :65 */

        float wx = to_snoise3(noisex, noisey, uvx + noiseTime);               /* :67 */
        float wy = to_snoise3(noisex, noisey, uvy + noiseTime + 1234.5678f);  /* :68 */

        /* :75 flowAtScreenPos(pos*viewSize, ...): src/flow/flow-at-screen-pos.glsl:13-27
         * with levels = stride = 1 (single tap, factor 1, flowMax 1);
         * posToUV = glsl-map map(v,-1,1,0,1) = 0 + (1-0)*(v-(-1))/(1-(-1)) */
        float sx = posx * u->viewSize[0], sy = posy * u->viewSize[1];
        float fu = 0.0f + (1.0f * (sx + 1.0f)) / 2.0f;
        float fv = 0.0f + (1.0f * (sy + 1.0f)) / 2.0f;
        const float *ft = flow + 4 * ((size_t)nearest_texel(fv, fh) * fw + nearest_texel(fu, fw));
        float gxf, gyf;
        flow_get(ft, u->time, u->flowDecay, &gxf, &gyf);
        float ffx = (0.0f + gxf * 1.0f) / 1.0f, ffy = (0.0f + gyf * 1.0f) / 1.0f;

        /* :79-82 */
        float vfw = varyf(u->forceWeight, i, u->varyForce);
        float vflw = varyf(u->flowWeight, i, u->varyFlow);
        float vnw = varyf(u->noiseWeight, i, u->varyNoise);
        nvx = (velx * u->damping * u->dt) + (vfw * ((ffx * u->dt * vflw) + (wx * u->dt * vnw)));
        nvy = (vely * u->damping * u->dt) + (vfw * ((ffy * u->dt * vflw) + (wy * u->dt * vnw)));

        /* :85 */
        float vtg = varyf(u->target, i, u->varyTarget);
        nvx += (target_texel[0] - posx) * vtg;
        nvy += (target_texel[1] - posy) * vtg;

        /* :92-94 (speed == 0 gives 0/0 = NaN, as in the reference) */
        float speed = sqrtf(nvx * nvx + nvy * nvy);
        float r = fminf(speed, u->speedLimit) / speed;
        nvx *= r; nvy *= r;

        /* :97 */
        npx = posx + nvx; npy = posy + nvy;
    }
    out[0] = npx; out[1] = npy; out[2] = nvx; out[3] = nvy;   /* :100 */
}

/*
 * One `Particles.step` pass (src/particles.js:123-145) over rows
 * [y0, y0+rows) of a data_w x data_h state texture.  `in`/`out`/`targets`
 * point at the first of those rows (row-major RGBA32F texels, texel (x,y) at
 * [(y-y0)*data_w + x]).  `targets` may be NULL = a texture of zeros (a fresh
 * gl-fbo colour attachment).
 */
void to_logic_step(const to_logic_uniforms *u, const float *in, float *out, int y0, int rows,
                   const float *flow, int fw, int fh, const float *targets)
{
    static const float zero4[4] = {0, 0, 0, 0};
    const int W = u->data_w;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        for (int x = 0; x < W; ++x) {
            size_t o = 4 * ((size_t)r * W + x);
            logic_texel(u, x, y0 + r, in + o, flow, fw, fh, targets ? targets + o : zero4, out + o);
        }
    }
}

/* src/spawn/init/index.frag:5-10, src/spawn/init/cpu.js:3-8 */
void to_spawn_init(float *out, size_t texels)
{
    for (size_t k = 0; k < texels; ++k) {
        out[4 * k + 0] = TO_INERT; out[4 * k + 1] = TO_INERT;
        out[4 * k + 2] = 0.0f;     out[4 * k + 3] = 0.0f;
    }
}

/* ------------------------------------------------------------------------- */
/* src/optical-flow/index.frag:55-81 (compiled text: docs/js/demo.js:73)       */
/* ------------------------------------------------------------------------- */

/* NEAREST + CLAMP_TO_EDGE texel index of an 8-bit-per-channel texture as the captured
 * reference run computed it: the coordinate is clamped to [0, 1), truncated to 16 fractional
 * bits, and the texel is (coord16 * size) >> 16.  Differs from floor(u*size) only for
 * coordinates within 2^-16 of a texel boundary (GL leaves that precision to the implementation;
 * float textures - flow, state - took the plain floor path, see nearest_texel). */
static inline int nearest_texel_fx16(float u, int n)
{
    float c = u;
    if (!(c > 0.0f)) c = 0.0f;
    if (c > 65535.0f / 65536.0f) c = 65535.0f / 65536.0f;
    uint32_t fx = (uint32_t)(c * 65536.0f);
    return (int)((fx * (uint32_t)n) >> 16);
}

/* texture2D on an RGBA8 NEAREST/CLAMP texture, then grayScale()
 * (src/utils/gray-scale.glsl:2): dot(rgb, (0.3, 0.59, 0.11)) */
static inline float gray_tap(const uint8_t *img, int w, int h, float u, float v)
{
    const uint8_t *t = img + 4 * ((size_t)nearest_texel_fx16(v, h) * w + nearest_texel_fx16(u, w));
    /* UNORM8 -> float: GL leaves the conversion's rounding to the implementation; the captured
     * reference run widened to UNORM16 (c*257) and scaled by 1/65535 (bit-exact match), which
     * equals c/255 to within 1 ulp. */
    const float k = 1.0f / 65535.0f;
    float r = (float)(t[0] * 257) * k, g = (float)(t[1] * 257) * k, b = (float)(t[2] * 257) * k;
    return r * 0.3f + g * 0.59f + b * 0.11f;
}

void to_optical_flow(const to_optical_flow_uniforms *u, const uint8_t *view, const uint8_t *last,
                     int fr_w, int fr_h, float *flow, int out_w, int out_h, int blend)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < out_h; ++y) {
        for (int x = 0; x < out_w; ++x) {
            /* varying uv = position.xy (src/screen/index.vert:6-10): NDC of the pixel centre,
             * interpolated by the rasteriser as (x+0.5)*A - 1 with gradient A = d(uv)/dx.
             * A == fl(2/W) in every captured reference run kept as a fixture (64, 96, 240,
             * 1920 wide; 48, 64, 135, 1080 high); the reference rasteriser derives A through an
             * approximate reciprocal and can be 1 ulp off at other sizes (e.g. W = 100), where
             * GL leaves the interpolation precision implementation-defined. */
            float ux = ((float)x + 0.5f) * (2.0f / (float)out_w) - 1.0f;
            float uy = ((float)y + 0.5f) * (2.0f / (float)out_h) - 1.0f;
            /* :56 st = posToUV(uv*scaleUV/viewSize) */
            float px = ux * u->scaleUV[0] / u->viewSize[0], py = uy * u->scaleUV[1] / u->viewSize[1];
            float sx = 0.0f + (1.0f * (px + 1.0f)) / 2.0f, sy = 0.0f + (1.0f * (py + 1.0f)) / 2.0f;
            float o = u->offset;
            /* :63-67 */
            float gx = (gray_tap(view, fr_w, fr_h, sx + o, sy + 0.0f) - gray_tap(view, fr_w, fr_h, sx - o, sy - 0.0f)) +
                       (gray_tap(last, fr_w, fr_h, sx + o, sy + 0.0f) - gray_tap(last, fr_w, fr_h, sx - o, sy - 0.0f));
            float gy = (gray_tap(view, fr_w, fr_h, sx + 0.0f, sy + o) - gray_tap(view, fr_w, fr_h, sx - 0.0f, sy - o)) +
                       (gray_tap(last, fr_w, fr_h, sx + 0.0f, sy + o) - gray_tap(last, fr_w, fr_h, sx - 0.0f, sy - o));
            float gm = sqrtf((gx * gx) + (gy * gy) + u->lambda);                     /* :69 */
            float diff = gray_tap(view, fr_w, fr_h, sx, sy) - gray_tap(last, fr_w, fr_h, sx, sy);   /* :72 */
            float vx = (diff * (gx / gm)) * u->speed, vy = (diff * (gy / gm)) * u->speed;            /* :78 */
            /* :80 bezier(vec3(0,0,1), t) = (0*ut+0*t)*ut + (0*ut+1*t)*t  (src/utils/bezier.glsl:9-13) */
            float t = sqrtf(vx * vx + vy * vy) / u->speedLimit;
            float ut = 1.0f - t;
            float bz = (0.0f * ut + 0.0f * t) * ut + (0.0f * ut + 1.0f * t) * t;
            float fx = bz * vx, fy = bz * vy;
            /* flow(vel, speedLimit): src/flow/apply/state.glsl:5-16 */
            float a = fminf(sqrtf(fx * fx + fy * fy) / u->speedLimit, 1.0f);
            float src[4] = {fx, fy, u->time, a};
            float *d = flow + 4 * ((size_t)y * out_w + x);
            if (blend) {
                /* gl.blendFunc(SRC_ALPHA, ONE_MINUS_SRC_ALPHA) on a float target (src/index.js:267-268) */
                float ia = 1.0f - a;
                for (int c = 0; c < 4; ++c) d[c] = src[c] * a + d[c] * ia;
            } else {
                for (int c = 0; c < 4; ++c) d[c] = src[c];
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Respawn passes.  The GLSL hash `fract(sin(x)*43758.5453)` (glsl-random      */
/* 0.0.5) and angleToVec()'s cos/sin amplify the last bits of the platform's   */
/* sin/cos, so no two GL implementations agree on them.  This   */
/* build pins them: sin/cos are evaluated by the fixed fp64 sequence below     */
/* (quadrant reduction + Taylor polynomials, every step an IEEE fp64 fma/mul/  */
/* add) and rounded once to fp32 - reproducible bit-for-bit on any IEEE        */
/* machine, and within one fp32 rounding of the true value.  Against the       */
/* reference captures these passes are therefore checked statistically.        */
/* ------------------------------------------------------------------------- */

static void sincos_pinned(float xf, float *s_out, float *c_out)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632679489655800e+00;   /* pi/2 rounded to fp64 */
    const double PIO2_LO = 6.12323399573676603587e-17;   /* pi/2 - PIO2_HI */
    double x = (double)xf;
    double k = __builtin_rint(x * TWO_OVER_PI);
    double r = __builtin_fma(-k, PIO2_HI, x);
    r = __builtin_fma(-k, PIO2_LO, r);
    double r2 = r * r;
    /* sin(r) = r + r^3*(S1 + r^2*(S2 + ...)), |r| <= pi/4 */
    double ps = -1.0 / 1307674368000.0;                       /* -1/15! */
    ps = __builtin_fma(ps, r2, 1.0 / 6227020800.0);           /*  1/13! */
    ps = __builtin_fma(ps, r2, -1.0 / 39916800.0);            /* -1/11! */
    ps = __builtin_fma(ps, r2, 1.0 / 362880.0);               /*  1/9!  */
    ps = __builtin_fma(ps, r2, -1.0 / 5040.0);                /* -1/7!  */
    ps = __builtin_fma(ps, r2, 1.0 / 120.0);                  /*  1/5!  */
    ps = __builtin_fma(ps, r2, -1.0 / 6.0);                   /* -1/3!  */
    double sr = __builtin_fma(ps * r2, r, r);
    /* cos(r) = 1 + r^2*(C1 + r^2*(C2 + ...)) */
    double pc = 1.0 / 20922789888000.0;                       /*  1/16! */
    pc = __builtin_fma(pc, r2, -1.0 / 87178291200.0);         /* -1/14! */
    pc = __builtin_fma(pc, r2, 1.0 / 479001600.0);            /*  1/12! */
    pc = __builtin_fma(pc, r2, -1.0 / 3628800.0);             /* -1/10! */
    pc = __builtin_fma(pc, r2, 1.0 / 40320.0);                /*  1/8!  */
    pc = __builtin_fma(pc, r2, -1.0 / 720.0);                 /* -1/6!  */
    pc = __builtin_fma(pc, r2, 1.0 / 24.0);                   /*  1/4!  */
    pc = __builtin_fma(pc, r2, -0.5);                         /* -1/2!  */
    double cr = __builtin_fma(pc, r2, 1.0);
    long q = (long)k & 3;                                     /* two's complement: valid for k < 0 */
    double s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    double c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
    *s_out = (float)s;
    *c_out = (float)c;
}

static inline float modf_glsl(float x, float y) { return x - y * floorf(x / y); }   /* GLSL mod() */
static inline float fractf_glsl(float x) { return x - floorf(x); }                 /* GLSL fract() */

/* glsl-random 0.0.5 (required at src/spawn/ball/index.frag:6, src/spawn/pixels/frag/head.frag:21;
 * compiled text docs/js/demo.js:71): fract(sin(mod(dot(co,(12.9898,78.233)),3.14))*43758.5453) */
float to_random(float cox, float coy)
{
    float dt = cox * 12.9898f + coy * 78.233f;
    float sn = modf_glsl(dt, 3.14f);
    float s, c;
    sincos_pinned(sn, &s, &c);
    return fractf_glsl(s * 43758.5453f);
}

/* src/spawn/ball/index.frag:11-19 */
void to_spawn_ball(const to_spawn_ball_uniforms *u, float *out, int w, int y0, int rows)
{
    const float tau = 6.28318530717958647692f;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        for (int x = 0; x < w; ++x) {
            float fx = (float)x + 0.5f, fy = (float)(y0 + r) + 0.5f;
            float r0 = to_random(fx * 1.7654f + 2.3675f, fy * 1.7654f + 2.3675f);
            float r1 = to_random(fx * 1.23494f + 0.36434f, fy * 1.23494f + 0.36434f);
            float r2 = to_random(fx * 0.327789f + 3.498787f, fy * 0.327789f + 3.498787f);
            float r3 = to_random(fx * 9.0374f + 0.2773f, fy * 9.0374f + 0.2773f);
            float s0, c0, s1, c1;
            sincos_pinned(r0 * tau, &s0, &c0);        /* angleToVec: vec2(cos, sin), src/utils/angle-to-vec.glsl */
            sincos_pinned(r2 * tau, &s1, &c1);
            float *o = out + 4 * ((size_t)r * w + x);
            o[0] = c0 * r1 * u->radius; o[1] = s0 * r1 * u->radius;
            o[2] = c1 * r3 * u->speed;  o[3] = s1 * r3 * u->speed;
        }
    }
}

/* float-texture NEAREST/CLAMP fetch at uv */
static inline const float *tex_f32(const float *tex, int w, int h, float u, float v)
{
    return tex + 4 * ((size_t)nearest_texel(v, h) * w + nearest_texel(u, w));
}

/* src/spawn/pixels/frag/head.frag:28-34; mix(a,b,t) = a*(1-t) + b*t (GLSL ES 1.0 8.3) */
static void spawn_to_pos(const to_spawn_sample_uniforms *u, float uvx, float uvy, float *px, float *py)
{
    float tt = u->time * 0.001f;
    float ra = to_random(uvx - 1.2345f + tt, uvy - 1.2345f + tt);
    float rb = to_random(uvx + 1.2345f + tt, uvy + 1.2345f + tt);
    float ox = (-u->jitter[0]) * (1.0f - ra) + u->jitter[0] * ra;
    float oy = (-u->jitter[1]) * (1.0f - rb) + u->jitter[1] * rb;
    /* uvToPos = map(uv, 0,1, -1,1) = -1 + (1-(-1))*(v-0)/(1-0)   (glsl-map 1.0.1) */
    float qx = -1.0f + (2.0f * ((uvx + ox) - 0.0f)) / 1.0f;
    float qy = -1.0f + (2.0f * ((uvy + oy) - 0.0f)) / 1.0f;
    qx = qx * 1.0f * u->spawnSize[0];                       /* *flipUV*spawnSize, flipUV = (1,-1) */
    qy = qy * -1.0f * u->spawnSize[1];
    /* transform(mat3 m, vec2 v) = (m*vec3(v,1)).xy, column-major m */
    const float *m = u->spawnMatrix;
    *px = m[0] * qx + m[3] * qy + m[6] * 1.0f;
    *py = m[1] * qx + m[4] * qy + m[7] * 1.0f;
}

/* filter/pass/vignette.glsl:9-11 with curve (0.1,1,1), mid 0.5, limit 0.6 (spawn/pixels/vignette-head.glsl:4-6) */
static inline float spawn_vignette(float u, float v)
{
    float dx = u - 0.5f, dy = v - 0.5f;
    float amt = fminf(1.0f - (sqrtf(dx * dx + dy * dy) / 0.6f), 1.0f);
    float ut = 1.0f - amt;
    float bz = (0.1f * ut + 1.0f * amt) * ut + (1.0f * ut + 1.0f * amt) * amt;     /* utils/bezier.glsl:9-13 */
    return fmaxf(0.0f, bz);
}

/* spawn/pixels/apply/color.glsl:13-17 after the vignette pass (apply/compose-filter.glsl:10-12):
 * vec4(pos, angleToVec((hsv.r + time*0.00003)*tau)*hsv.g*hsv.b*pixel.a), hsv = rgb2hsv(pixel.rgb)
 * (libs/glsl-hsv/rgb-hsv.glsl:4-11).  cos/sin through the pinned evaluation (see sincos_pinned). */
static void spawn_apply_color(const float *texel, float vg, float time, float px, float py, float *out)
{
    const float r = texel[0] * vg, g = texel[1] * vg, b = texel[2] * vg, a = texel[3] * vg;
    float p0, p1, p2, p3;                                   /* vec4 p = (g < b) ? (b, g, -1, 2/3) : (g, b, 0, -1/3) */
    if (g < b) { p0 = b; p1 = g; p2 = -1.0f; p3 = 2.0f / 3.0f; } else { p0 = g; p1 = b; p2 = 0.0f; p3 = -1.0f / 3.0f; }
    float q0, q1, q2, q3;                                   /* vec4 q = (r < p.x) ? (p.xyw, r) : (r, p.yzx) */
    if (r < p0) { q0 = p0; q1 = p1; q2 = p3; q3 = r; } else { q0 = r; q1 = p1; q2 = p2; q3 = p0; }
    const float e = 1.0e-10f;
    const float d = q0 - fminf(q3, q1);
    const float h = fabsf(q2 + (q3 - q1) / (6.0f * d + e)), s = d / (q0 + e), v = q0;
    float sn, cs;
    sincos_pinned((h + (time * 0.00003f)) * 6.28318530717958647692f, &sn, &cs);
    out[0] = px; out[1] = py;
    out[2] = ((cs * s) * v) * a;
    out[3] = ((sn * s) * v) * a;
}

/* src/spawn/pixels/frag/direct-main.frag:10-21 (index.frag: colour apply over the vignette pass):
 * uv = (gl_FragCoord.xy/dataRes)*(geomRes/dataRes) with geomRes = [w, 2h] (src/index.js:195-197). */
void to_spawn_direct(const to_spawn_sample_uniforms *u, float *out, int y0, int rows,
                     const float *spawn_data, int sw, int sh)
{
    const int W = u->data_w;
    for (int r = 0; r < rows; ++r) {
        for (int x = 0; x < W; ++x) {
            size_t o = 4 * ((size_t)r * W + x);
            const float dw = (float)u->data_w, dh = (float)u->data_h;
            float uvx = (((float)x + 0.5f) / dw) * (dw / dw);
            float uvy = (((float)(y0 + r) + 0.5f) / dh) * ((2.0f * dh) / dh);
            float px, py;
            spawn_to_pos(u, uvx, uvy, &px, &py);
            float st[4];
            spawn_apply_color(tex_f32(spawn_data, sw, sh, uvx, uvy), spawn_vignette(uvx, uvy), u->time, px, py, st);
            out[o] = st[0]; out[o + 1] = st[1]; out[o + 2] = st[2] * u->speed; out[o + 3] = st[3] * u->speed;
        }
    }
}

/* src/spawn/pixels/frag/best-sample-main.frag:21-46 */
void to_spawn_sample(const to_spawn_sample_uniforms *u, const float *particles, float *out,
                     int y0, int rows, const float *spawn_data, int sw, int sh)
{
    const int W = u->data_w;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        for (int x = 0; x < W; ++x) {
            size_t o = 4 * ((size_t)r * W + x);
            float uvx = ((float)x + 0.5f) / (float)u->data_w, uvy = ((float)(y0 + r) + 0.5f) / (float)u->data_h;
            float st[4] = {particles[o], particles[o + 1], particles[o + 2], particles[o + 3]};
            float add = 1.2345f + (u->time * 0.001f);
            float base[4] = {st[0] + uvx + add, st[1] + uvy + add, st[2] + uvx + add, st[3] + uvy + add};
            for (int n = 0; n < u->samples; ++n) {
                float fn = (float)n;
                float su = modf_glsl(to_random(base[0] + fn, base[1] + fn), 1.0f);
                float sv = modf_glsl(to_random(base[2] + fn, base[3] + fn), 1.0f);
                float px, py;
                spawn_to_pos(u, su, sv, &px, &py);
                const float *t = tex_f32(spawn_data, sw, sh, su, sv);
                float other[4];
                if (u->apply == 3) {
                    /* bright-sample.frag -> apply/brightest.glsl:11-15 (no filter pass):
                     * vec4(pos, angleToVec(mod(random(uv*dot(pixel.rg, pixel.ba)), 1.0)*tau)*luma(pixel)*pixel.a) */
                    float sc = t[0] * t[2] + t[1] * t[3];
                    float ang = modf_glsl(to_random(su * sc, sv * sc), 1.0f) * 6.28318530717958647692f;
                    float sn, cs;
                    sincos_pinned(ang, &sn, &cs);
                    float lum = (t[0] * 0.299f + t[1] * 0.587f) + t[2] * 0.114f;       /* glsl-luma */
                    other[0] = px; other[1] = py;
                    other[2] = (cs * lum) * t[3];
                    other[3] = (sn * lum) * t[3];
                } else if (u->apply == 2) {
                    /* best-sample.frag: colour apply over the vignette pass */
                    spawn_apply_color(t, spawn_vignette(su, sv), u->time, px, py, other);
                } else if (u->apply == 0) {
                    /* apply/flow.glsl:11-13: vec4(pos, getFlow(pixel, time, decay)) */
                    float k = fmaxf(0.0f, 1.0f - ((u->time - t[2]) * u->flowDecay));
                    other[0] = px; other[1] = py; other[2] = t[0] * k; other[3] = t[1] * k;
                } else {
                    /* data-sample.frag: identity after filter/pass/vignette.glsl:9-11 with
                     * curve (0.1,1,1), mid 0.5, limit 0.6 (vignette-head.glsl:4-6) */
                    float vg = spawn_vignette(su, sv);
                    for (int c = 0; c < 4; ++c) other[c] = t[c] * vg;
                }
                float cand[4] = {other[0], other[1], other[2] * u->speed, other[3] * u->speed};
                float tc = st[2] * st[2] + st[3] * st[3], tn = cand[2] * cand[2] + cand[3] * cand[3];
                if (!(tc > u->bias * tn))       /* pick(): keep current only if strictly greater */
                    for (int c = 0; c < 4; ++c) st[c] = cand[c];
            }
            for (int c = 0; c < 4; ++c) out[o + c] = st[c];
        }
    }
}


/* ---------------------------------------------------------------------------------------------------------------
 * Flow deposit.  Restates what the reference's draw() does to the flow FBO on the GL it was
 * captured on (WebGL 1 over SwiftShader: ALIASED_LINE_WIDTH_RANGE = [1, 1]), pinned by tests/golden/deposit_*.npz:
 *   vertex stream   Particles.generateLUT([w, 2h]): uv = (i/(w-1), j/(2h-1)) as Float32, i outer, j inner
 *                   (src/particles.js:171-190); gl.LINES pairs stream vertices (2k, 2k+1) = (i, j = 2m), (i, 2m+1)
 *   vertex shader   stateAtFrame (src/state/state-at-frame.glsl:12-22): nearIndex = uv.y*dataRes.y; the vertex reads
 *                   `current` when fract(nearIndex) > 0.25, else `previous`, at (uv.x, floor(nearIndex)/dataRes.y);
 *                   gl_Position = (state.xy*viewSize, 1, 1), color = (vel, time, min(|vel|/speedLimit, 1))
 *                   (src/flow/vert/main.vert:10-17, apply/state.glsl:5-16) - only when state.xy != inert.
 *                   (About half of the pairs read `current` twice at the same texel - zero-length, no fragments -
 *                   as nearIndex drifts by j/(2h-1): that is the reference's behaviour and is kept.)
 *   DEVIATION       a pair with an inert vertex: the reference leaves gl_Position and the varying unwritten
 *                   (undefined by GLSL ES 1.00 10.x; the captured GL draws streaks towards clip-space (0,0,0,0)).
 *                   Here such a pair draws nothing.
 *   rasteriser      width-1 line = the hexagon spanned by the two endpoint diamonds (|dx|+|dy| <= 1/2 px), clipped
 *                   to the view volume in clip space, vertices snapped to 1/16 px, scan-converted with ceil() edges:
 *                   texel centres with left <= x < right (reproduces the captured coverage texel for texel: ties,
 *                   sub-texel lines and lines crossing the view's edge included)
 *   WIDE LINES      (UNPINNED: no GL within reach draws them - the captured one clamps every width to 1.)  A line of
 *                   width w is the same construction with the endpoint diamonds scaled by w (|dx|+|dy| <= w/2 px):
 *                   the hexagon then measures exactly w texels across in the line's minor direction, which is what
 *                   the GL specification asks of a non-antialiased wide line (OpenGL ES 2.0 3.4.2.1: a column of w
 *                   fragments per step in the major direction), and it is the captured rasteriser's own shape with its
 *                   one constant (the diamond's half-diagonal) multiplied - w = 1 is bit for bit the pinned case.  The
 *                   varying is projected on the line as before, so it is constant across the line's width.
 *   interpolation   the varying is linear along the SNAPPED endpoints (orthogonal projection, extrapolated beyond
 *                   them, unclamped); coinciding snapped endpoints give the first vertex's values
 *   blend           SRC_ALPHA, ONE_MINUS_SRC_ALPHA on all four channels, lines in stream order (src/index.js:267-268)
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { int live; float px, py; float c[4]; float uvx, uvy; const float *state; } deposit_vertex;

static void deposit_fetch(const to_deposit_uniforms *u, const float *current, const float *previous,
                          int i, int j, deposit_vertex *v)
{
    const int W = u->data_w, H = u->data_h;
    const int lw = W > 2 ? W : 2, lh = 2 * H > 2 ? 2 * H : 2;
    const double inv_x = 1.0 / (double)(lw - 1), inv_y = 1.0 / (double)(lh - 1);
    const float uvx = (float)((double)i * inv_x), uvy = (float)((double)j * inv_y);   /* Float32Array of JS doubles */
    const float near_index = uvy * (float)H;
    const float fl = floorf(near_index);
    const float offset = near_index - fl;                                             /* fract() */
    const float ly = fl / (float)H;
    const float *tex = offset > 0.25f ? current : previous;
    const float *t = tex + 4 * ((size_t)nearest_texel(ly, H) * W + nearest_texel(uvx, W));
    v->live = (t[0] != TO_INERT) || (t[1] != TO_INERT);
    v->px = t[0] * u->viewSize[0];
    v->py = t[1] * u->viewSize[1];
    v->c[0] = t[2]; v->c[1] = t[3]; v->c[2] = u->time;
    v->c[3] = fminf(sqrtf(t[2] * t[2] + t[3] * t[3]) / u->speedLimit, 1.0f);
    v->uvx = uvx; v->uvy = uvy; v->state = t;
}

/* The view pass of draw() (src/index.js:333-337) sends the same vertex stream through src/render/index.vert:58-100
 * instead: position as above, colour = base colour + colour map + flow-aligned colour, alpha scaled by the speed and
 * a vignette.  Operation order as in the shader; sin(time*flowDecay) is a uniform-only expression whose value is
 * implementation-defined - the caller evaluates it (sinTerm).  glsl-map: outMin + (outMax-outMin)*(v-inMin)/(inMax-inMin);
 * mix(a, b, t) = a*(1-t) + b*t; bezier(vec3) and vignette(): src/utils/bezier.glsl:9-13, src/filter/vignette.glsl:5-28. */
static void render_color(const to_render_uniforms *r, const float *state, float uvx, float uvy, int W, int H,
                         const float *colormap, int cw, int ch, float *c)
{
    const float posx = state[0], posy = state[1];
    const float velx = state[2] / r->speedLimit, vely = state[3] / r->speedLimit;
    const float speedRate = fminf((velx * velx + vely * vely) / r->speedAlpha, 1.0f);
    /* colour map at uv*geomRes/dataRes, geomRes = [W, 2H] */
    const float mu = uvx * (float)W / (float)W, mv = uvy * (float)(2 * H) / (float)H;
    const float *mt = colormap ? colormap + 4 * ((size_t)nearest_texel(mv, ch) * cw + nearest_texel(mu, cw)) : NULL;
    float mapped[4];
    for (int k = 0; k < 4; ++k) mapped[k] = (mt ? mt[k] : 0.0f) * r->colorMapAlpha;
    /* flowAxisR/G/B = angleToVec(0), (tau/3), (2 tau/3) as the shader's literals */
    const float axr[2] = {1.0f, 0.0f}, axg[2] = {-0.5000000000000004f, -0.8660254037844385f}, axb[2] = {-0.4999999999999998f, 0.8660254037844387f};
    const float al[3] = {velx * axr[0] + vely * axr[1], velx * axg[0] + vely * axg[1], velx * axb[0] + vely * axb[1]};
    const float gbr[3] = {al[1] * (1.0f - r->flowDecay), al[2] * (1.0f - r->flowDecay), al[0] * (1.0f - r->flowDecay)};
    float fa[3];
    for (int k = 0; k < 3; ++k) {
        const float m = al[k] * (1.0f - r->sinTerm) + gbr[k] * r->sinTerm;
        fa[k] = 0.0f + (1.0f - 0.0f) * (m - -1.0f) / (1.0f - -1.0f);
    }
    const float flw[4] = {r->flowColor[0] * fa[0], r->flowColor[1] * fa[1], r->flowColor[2] * fa[2], r->flowColor[3]};
    const float *terms[3] = {r->baseColor, mapped, flw};
    for (int k = 0; k < 4; ++k) c[k] = 0.0f;
    for (int t = 0; t < 3; ++t) {
        const float a = terms[t][3];
        const float pre[4] = {terms[t][0] * a, terms[t][1] * a, terms[t][2] * a, a};
        for (int k = 0; k < 4; ++k) c[k] = c[k] + fminf(fmaxf(pre[k], 0.0f), 1.0f);       /* clamp(x, min, max) = min(max(x, min), max) */
    }
    /* vignette(pos, center = 0, limit = 1, curve = falloff = (0.2, 1, 1)), clamped to fadeRange = (0.2, 1) */
    const float amount = fminf(1.0f - (sqrtf(posx * posx + posy * posy) / 1.0f), 1.0f);
    const float ut = 1.0f - amount;
    const float bz = (0.2f * ut + 1.0f * amount) * ut + (1.0f * ut + 1.0f * amount) * amount;
    const float vg = fmaxf(0.0f, bz);
    c[3] = c[3] * (speedRate * fminf(fmaxf(vg, 0.2f), 1.0f));
}

static inline long ceil_div(long long a, long long b)     /* b > 0 */
{
    long long q = a / b;
    if (a % b > 0) ++q;
    return (long)q;
}

static inline int snap16(float ndc, float scale, float offset) { return (int)lrintf(ndc * scale + offset); }

/* the rasteriser and the in-order blend; r == NULL: the flow pass into `flow` (RGBA32F), else the view pass into `view`
 * (RGBA8: the fragment colour is clamped to [0, 1], blended with the stored colour c/255 and stored as round(255 x)) */
static long deposit_core(const to_deposit_uniforms *u, const to_render_uniforms *r, const float *colormap, int cw, int ch,
                         const float *current, const float *previous, float *flow, uint8_t *view, int fw, int fh, int32_t *coverage)
{
    const int W = u->data_w, H = u->data_h;
    const float wx16 = 8.0f * (float)fw, wy16 = 8.0f * (float)fh;          /* 16 * viewport/2 */
    const float x0 = wx16 - 8.0f, y0 = wy16 - 8.0f;                         /* texel centres at integer*16 */
    const float lw = u->lineWidth > 0.0f ? u->lineWidth : 1.0f;
    const float hx = (0.5f * lw) / (0.5f * (float)fw), hy = (0.5f * lw) / (0.5f * (float)fh);   /* half the diamond (half a texel x width) in NDC */
    long fragments = 0;
    enum { SPAN = 64 };
    for (int i = 0; i < W; ++i) {
        for (int m = 0; m < H; ++m) {
            deposit_vertex a, b;
            deposit_fetch(u, current, previous, i, 2 * m, &a);
            deposit_fetch(u, current, previous, i, 2 * m + 1, &b);
            if (!a.live || !b.live) continue;                                /* DEVIATION, see above */
            if (r) {
                render_color(r, a.state, a.uvx, a.uvy, W, H, colormap, cw, ch, a.c);
                render_color(r, b.state, b.uvx, b.uvy, W, H, colormap, cw, ch, b.c);
            }
            const float dx = (0.5f * (float)fw) * (b.px - a.px), dy = (0.5f * (float)fh) * (b.py - a.py);
            if (dx == 0.0f && dy == 0.0f) continue;
            /* fixed-point range: endpoints beyond 1024 half-widths of the view (or non-finite) deposit nothing */
            if (!(fabsf(a.px) <= 1024.0f && fabsf(a.py) <= 1024.0f && fabsf(b.px) <= 1024.0f && fabsf(b.py) <= 1024.0f))
                continue;
            /* the hexagon in clip space (w = 1): left, top, right, bottom of both endpoint diamonds */
            const deposit_vertex *vv[2] = {&a, &b};
            float hxv[6], hyv[6];
            int sx[2], sy[2];
            for (int k = 0; k < 2; ++k) { sx[k] = snap16(vv[k]->px, wx16, x0); sy[k] = snap16(vv[k]->py, wy16, y0); }
#define TO_L(n, k) do { hxv[n] = vv[k]->px - hx; hyv[n] = vv[k]->py; } while (0)
#define TO_T(n, k) do { hxv[n] = vv[k]->px; hyv[n] = vv[k]->py + hy; } while (0)
#define TO_R(n, k) do { hxv[n] = vv[k]->px + hx; hyv[n] = vv[k]->py; } while (0)
#define TO_B(n, k) do { hxv[n] = vv[k]->px; hyv[n] = vv[k]->py - hy; } while (0)
            if (dx > dy) {
                if (dx > -dy) { TO_L(0, 0); TO_T(1, 0); TO_T(2, 1); TO_R(3, 1); TO_B(4, 1); TO_B(5, 0); }
                else          { TO_L(0, 1); TO_L(1, 0); TO_T(2, 0); TO_R(3, 0); TO_R(4, 1); TO_B(5, 1); }
            } else {
                if (dx > -dy) { TO_L(0, 0); TO_L(1, 1); TO_T(2, 1); TO_R(3, 1); TO_R(4, 0); TO_B(5, 0); }
                else          { TO_L(0, 1); TO_T(1, 1); TO_T(2, 0); TO_R(3, 0); TO_B(4, 0); TO_B(5, 1); }
            }
#undef TO_L
#undef TO_T
#undef TO_R
#undef TO_B
            /* clip against the view volume's side planes (Sutherland-Hodgman, intersection as
             * (dj*Vi - di*Vj) * (1/(dj - di)) with the inside vertex first), then snap to 1/16 texel */
            float cx[16], cy[16], tx_[16], ty_[16];
            int n = 6;
            for (int k = 0; k < 6; ++k) { cx[k] = hxv[k]; cy[k] = hyv[k]; }
            for (int plane = 0; plane < 4 && n >= 3; ++plane) {
                int t = 0;
                for (int k = 0; k < n; ++k) {
                    const int j = k == n - 1 ? 0 : k + 1;
                    float di, dj;
                    switch (plane) {
                    case 0: di = 1.0f + cx[k]; dj = 1.0f + cx[j]; break;     /* left   */
                    case 1: di = 1.0f - cx[k]; dj = 1.0f - cx[j]; break;     /* right  */
                    case 2: di = 1.0f - cy[k]; dj = 1.0f - cy[j]; break;     /* top    */
                    default: di = 1.0f + cy[k]; dj = 1.0f + cy[j]; break;    /* bottom */
                    }
                    if (di >= 0.0f) {
                        tx_[t] = cx[k]; ty_[t] = cy[k]; ++t;
                        if (dj < 0.0f) {
                            const float D = 1.0f / (dj - di);
                            tx_[t] = (dj * cx[k] - di * cx[j]) * D; ty_[t] = (dj * cy[k] - di * cy[j]) * D; ++t;
                        }
                    } else if (dj > 0.0f) {
                        const float D = 1.0f / (di - dj);
                        tx_[t] = (di * cx[j] - dj * cx[k]) * D; ty_[t] = (di * cy[j] - dj * cy[k]) * D; ++t;
                    }
                }
                n = t;
                for (int k = 0; k < n; ++k) { cx[k] = tx_[k]; cy[k] = ty_[k]; }
            }
            if (n < 3) continue;
            int PX[16], PY[16];
            for (int k = 0; k < n; ++k) { PX[k] = snap16(cx[k], wx16, x0); PY[k] = snap16(cy[k], wy16, y0); }
            int ymin = PY[0], ymax = PY[0];
            for (int k = 1; k < n; ++k) { if (PY[k] < ymin) ymin = PY[k]; if (PY[k] > ymax) ymax = PY[k]; }
            int r0 = (ymin + 15) >> 4, r1 = (ymax + 15) >> 4;               /* rows [r0, r1) */
            if (r0 < 0) r0 = 0;
            if (r1 > fh) r1 = fh;
            if (r0 >= r1) continue;
            /* long lines are walked in windows of SPAN rows */
            for (int base = r0; base < r1; base += SPAN) {
                const int top = base + SPAN < r1 ? base + SPAN : r1;
                int left[SPAN], right[SPAN];
                for (int k = 0; k < top - base; ++k) { left[k] = fw; right[k] = 0; }
                for (int k = 0; k < n; ++k) {
                    int Xa = PX[k], Ya = PY[k], Xb = PX[(k + 1) % n], Yb = PY[(k + 1) % n];
                    if (Ya == Yb) continue;
                    const int swap = Yb < Ya;
                    const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
                    int e0 = (Y1 + 15) >> 4, e1 = (Y2 + 15) >> 4;
                    if (e0 < base) e0 = base;
                    if (e1 > top) e1 = top;
                    const long long DX = X2 - X1, DY = Y2 - Y1;
                    for (int y = e0; y < e1; ++y) {
                        long x = ceil_div(DX * (((long long)y << 4) - Y1) + (long long)X1 * DY, 16 * DY);
                        if (x < 0) x = 0;
                        if (x > fw) x = fw;
                        if (swap) right[y - base] = (int)x; else left[y - base] = (int)x;
                    }
                }
                const long long ex = sx[1] - sx[0], ey = sy[1] - sy[0], den = ex * ex + ey * ey;
                for (int y = base; y < top; ++y) {
                    for (int x = left[y - base]; x < right[y - base]; ++x) {
                        float c[4];
                        if (den == 0) { for (int k = 0; k < 4; ++k) c[k] = a.c[k]; }
                        else {
                            const long long num = ((long long)(x << 4) - sx[0]) * ex + ((long long)(y << 4) - sy[0]) * ey;
                            const float t = (float)num / (float)den;
                            for (int k = 0; k < 4; ++k) c[k] = a.c[k] + t * (b.c[k] - a.c[k]);
                        }
                        if (view) {
                            uint8_t *q = view + 4 * ((size_t)y * fw + x);
                            for (int k = 0; k < 4; ++k) c[k] = fminf(fmaxf(c[k], 0.0f), 1.0f);
                            const float sa = c[3], da = 1.0f - sa;
                            for (int k = 0; k < 4; ++k) {
                                const float o = c[k] * sa + ((float)q[k] * (1.0f / 255.0f)) * da;
                                q[k] = (uint8_t)(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f + 0.5f);
                            }
                        } else {
                            float *d = flow + 4 * ((size_t)y * fw + x);
                            const float sa = c[3], da = 1.0f - sa;
                            for (int k = 0; k < 4; ++k) d[k] = c[k] * sa + d[k] * da;
                        }
                        if (coverage) ++coverage[(size_t)y * fw + x];
                        ++fragments;
                    }
                }
            }
        }
    }
    return fragments;
}


long to_flow_deposit(const to_deposit_uniforms *u, const float *current, const float *previous,
                     float *flow, int fw, int fh, int32_t *coverage)
{
    return deposit_core(u, NULL, NULL, 0, 0, current, previous, flow, NULL, fw, fh, coverage);
}

long to_view_render(const to_deposit_uniforms *u, const to_render_uniforms *r, const float *colormap, int cw, int ch,
                    const float *current, const float *previous, uint8_t *view, int vw, int vh)
{
    return deposit_core(u, r, colormap, cw, ch, current, previous, NULL, view, vw, vh, NULL);
}

/* Tendrils.drawFill (src/index.js:350-356): one full-screen quad of `color`, blended like everything else */
void to_view_fill(uint8_t *view, int vw, int vh, const float *color)
{
    float c[4];
    for (int k = 0; k < 4; ++k) c[k] = fminf(fmaxf(color[k], 0.0f), 1.0f);
    const float sa = c[3], da = 1.0f - sa;
    for (size_t i = 0; i < (size_t)vw * vh; ++i)
        for (int k = 0; k < 4; ++k) {
            const float o = c[k] * sa + ((float)view[4 * i + k] * (1.0f / 255.0f)) * da;
            view[4 * i + k] = (uint8_t)(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f + 0.5f);
        }
}


/* ---------------------------------------------------------------------------------------------------------------
 * GeometrySpawner's draw (src/spawn/geometry/index.js:97-115): triangles gl_Position = (position*viewSize, 0, 1)
 * (src/geom/vert/index.vert:3-5) in a constant colour (src/geom/frag/index.frag:3-5) into the spawner's float
 * buffer, blend SRC_ALPHA / ONE_MINUS_SRC_ALPHA as left by Tendrils.step().  Same rasteriser conventions as the
 * flow deposit above (clip-space clipping, 1/16-texel snapping, ceil() scan conversion, texels left <= x < right);
 * either winding is drawn (no culling).  Pinned by tests/golden/geometry_*.npz.
 * ------------------------------------------------------------------------------------------------------------- */
void to_triangles(const float *positions, int ntri, const float *view_size, const float *color,
                  float *img, int w, int h)
{
    const float wx16 = 8.0f * (float)w, wy16 = 8.0f * (float)h;
    const float x0 = wx16 - 8.0f, y0 = wy16 - 8.0f;
    for (int t = 0; t < ntri; ++t) {
        float cx[16], cy[16], tx_[16], ty_[16];
        int n = 3;
        for (int k = 0; k < 3; ++k) {
            cx[k] = positions[6 * t + 2 * k] * view_size[0];
            cy[k] = positions[6 * t + 2 * k + 1] * view_size[1];
        }
        for (int plane = 0; plane < 4 && n >= 3; ++plane) {
            int q = 0;
            for (int k = 0; k < n; ++k) {
                const int j = k == n - 1 ? 0 : k + 1;
                float di, dj;
                switch (plane) {
                case 0: di = 1.0f + cx[k]; dj = 1.0f + cx[j]; break;
                case 1: di = 1.0f - cx[k]; dj = 1.0f - cx[j]; break;
                case 2: di = 1.0f - cy[k]; dj = 1.0f - cy[j]; break;
                default: di = 1.0f + cy[k]; dj = 1.0f + cy[j]; break;
                }
                if (di >= 0.0f) {
                    tx_[q] = cx[k]; ty_[q] = cy[k]; ++q;
                    if (dj < 0.0f) {
                        const float D = 1.0f / (dj - di);
                        tx_[q] = (dj * cx[k] - di * cx[j]) * D; ty_[q] = (dj * cy[k] - di * cy[j]) * D; ++q;
                    }
                } else if (dj > 0.0f) {
                    const float D = 1.0f / (di - dj);
                    tx_[q] = (di * cx[j] - dj * cx[k]) * D; ty_[q] = (di * cy[j] - dj * cy[k]) * D; ++q;
                }
            }
            n = q;
            for (int k = 0; k < n; ++k) { cx[k] = tx_[k]; cy[k] = ty_[k]; }
        }
        if (n < 3) continue;
        int PX[16], PY[16];
        for (int k = 0; k < n; ++k) { PX[k] = snap16(cx[k], wx16, x0); PY[k] = snap16(cy[k], wy16, y0); }
        /* orientation of the snapped polygon: edges running up in y are left edges (as for the deposit's hexagons);
         * a polygon of the other winding is walked backwards */
        long long area2 = 0;
        for (int k = 0; k < n; ++k) {
            const int j = (k + 1) % n;
            area2 += (long long)PX[k] * PY[j] - (long long)PX[j] * PY[k];
        }
        if (area2 == 0) continue;
        if (area2 > 0)
            for (int a = 0, b = n - 1; a < b; ++a, --b) {
                int tmp = PX[a]; PX[a] = PX[b]; PX[b] = tmp;
                tmp = PY[a]; PY[a] = PY[b]; PY[b] = tmp;
            }
        for (int y = 0; y < h; ++y) {
            int left = w, right = 0;
            for (int k = 0; k < n; ++k) {
                int Xa = PX[k], Ya = PY[k], Xb = PX[(k + 1) % n], Yb = PY[(k + 1) % n];
                if (Ya == Yb) continue;
                const int swap = Yb < Ya;
                const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
                if (y < ((Y1 + 15) >> 4) || y >= ((Y2 + 15) >> 4)) continue;
                const long long DX = X2 - X1, DY = Y2 - Y1;
                long x = ceil_div(DX * (((long long)y << 4) - Y1) + (long long)X1 * DY, 16 * DY);
                if (x < 0) x = 0;
                if (x > w) x = w;
                if (swap) right = (int)x; else left = (int)x;
            }
            for (int x = left; x < right; ++x) {
                float *d = img + 4 * ((size_t)y * w + x);
                const float sa = color[3], da = 1.0f - sa;
                for (int k = 0; k < 4; ++k) d[k] = color[k] * sa + d[k] * da;
            }
        }
    }
}


/* Trail export (build-defined: the reference never reads its lines back): the line list that
 * draw() hands to GL - the same vertex stream, pairing and frame choice as to_flow_deposit (src/render/index.vert
 * and src/flow/vert/main.vert read the state through the same stateAtFrame) - as 12 floats per line in stream
 * order: p0.xy, p1.xy (clip space = state.xy*viewSize), then the two vertices' (vel.x, vel.y, time, alpha).
 * Lines with an inert vertex or zero length are left out.  Returns the number of lines. */
long to_export_lines(const to_deposit_uniforms *u, const float *current, const float *previous, float *out, long capacity)
{
    return to_export_view_lines(u, NULL, NULL, 0, 0, current, previous, out, capacity);
}

/* ... and with the view pass's vertex colours (src/render/index.vert:58-100) instead of the flow varyings when r != NULL */
long to_export_view_lines(const to_deposit_uniforms *u, const to_render_uniforms *r, const float *colormap, int cw, int ch,
                          const float *current, const float *previous, float *out, long capacity)
{
    const int W = u->data_w, H = u->data_h;
    long n = 0;
    for (int i = 0; i < W; ++i) {
        for (int m = 0; m < H; ++m) {
            deposit_vertex a, b;
            deposit_fetch(u, current, previous, i, 2 * m, &a);
            deposit_fetch(u, current, previous, i, 2 * m + 1, &b);
            if (!a.live || !b.live) continue;
            if (a.px == b.px && a.py == b.py) continue;
            if (r) {
                render_color(r, a.state, a.uvx, a.uvy, W, H, colormap, cw, ch, a.c);
                render_color(r, b.state, b.uvx, b.uvy, W, H, colormap, cw, ch, b.c);
            }
            if (n < capacity) {
                float *o = out + 12 * n;
                o[0] = a.px; o[1] = a.py; o[2] = b.px; o[3] = b.py;
                for (int k = 0; k < 4; ++k) { o[4 + k] = a.c[k]; o[8 + k] = b.c[k]; }
            }
            ++n;
        }
    }
    return n;
}
