// th_math.hpp - fp32 building blocks of the integrator, shared by host (table
// construction) and device (kernels).  The translation units that include this
// are compiled with -ffp-contract=off: `a * b + c` below is two correctly
// rounded fp32 operations (what the reference's shader compiler emits); a fused
// multiply-add appears only where written as th_fma().
//
// Algorithm provenance (paths relative to the reference tree):
//   snoise      glsl-noise 0.0.0 simplex/3d, required by src/logic.frag:36
//               (compiled text: docs/js/index.js:56, shader lines 48-138)
//   vary        src/logic.frag:41-43
#pragma once

#include <hip/hip_runtime.h>

#define TH_HD __host__ __device__ __forceinline__
#define TH_D __device__ __forceinline__

namespace th {

constexpr float kInert = -1000000.0f;          // src/const/inert.glsl:1
constexpr float kC6 = 1.0f / 6.0f;             // snoise C.x
constexpr float kC3 = 1.0f / 3.0f;             // snoise C.y
constexpr float kInv289 = 1.0f / 289.0f;
constexpr float kN7 = 0.142857142857f;         // snoise n_ (1/7)
constexpr float kNsX = kN7 * 2.0f;             // ns.x = n_*D.w - D.x
constexpr float kNsY = kN7 * 0.5f - 1.0f;      // ns.y = n_*D.y - D.z
constexpr float kNsZ = kN7;                    // ns.z = n_*D.z - D.x
constexpr float kTaylorA = 1.79284291400159f;
constexpr float kTaylorB = 0.85373472095314f;

// The noise gradient table is indexed by the ARGUMENT of the last permute();
// inside the guarded domain (|v| < kNoiseDomain) that argument is an integer
// in [kLutMin, kLutMax] (derivation in DESIGN.md "exact hash domain").
constexpr int kLutMin = -2;
constexpr int kLutMax = 581;
constexpr int kLutSize = kLutMax - kLutMin + 1;   // 584 entries of float4
constexpr float kNoiseDomain = 4194304.0f;         // 2^22

TH_HD float th_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TH_HD float th_floor(float x) { return __builtin_floorf(x); }

// a*b + c: fused only in FAST mode.
template <bool FAST>
TH_HD float mad(float a, float b, float c)
{
    if constexpr (FAST) return th_fma(a, b, c);
    else return a * b + c;
}

// ---- reference-order primitives (no assumptions about the operand range) ----

TH_HD float mod289_ref(float x) { return x - th_floor(x * kInv289) * 289.0f; }
TH_HD float permute_ref(float x) { return mod289_ref(((x * 34.0f) + 1.0f) * x); }
TH_HD float step_ref(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

// Gradient of one simplex corner from its hash p: "7x7 points over a square,
// mapped onto an octahedron", then taylorInvSqrt normalisation.
TH_HD void gradient_ref(float p, float &gx, float &gy, float &gz)
{
    float j = p - 49.0f * th_floor(p * kNsZ * kNsZ);
    float x_ = th_floor(j * kNsZ);
    float y_ = th_floor(j - 7.0f * x_);
    float x = x_ * kNsX + kNsY;
    float y = y_ * kNsX + kNsY;
    float h = 1.0f - __builtin_fabsf(x) - __builtin_fabsf(y);
    float sx = th_floor(x) * 2.0f + 1.0f;
    float sy = th_floor(y) * 2.0f + 1.0f;
    float sh = -step_ref(h, 0.0f);
    float px = x + sx * sh, py = y + sy * sh, pz = h;
    float norm = kTaylorA - kTaylorB * (px * px + py * py + pz * pz);
    gx = px * norm; gy = py * norm; gz = pz * norm;
}

// Full-range 3-D simplex noise in reference operation order (any input).
TH_HD float snoise_ref(float vx, float vy, float vz)
{
    float s = vx * kC3 + vy * kC3 + vz * kC3;
    float ix = th_floor(vx + s), iy = th_floor(vy + s), iz = th_floor(vz + s);
    float t = ix * kC6 + iy * kC6 + iz * kC6;
    float ax = vx - ix + t, ay = vy - iy + t, az = vz - iz + t;

    float gx = step_ref(ay, ax), gy = step_ref(az, ay), gz = step_ref(ax, az);
    float lx = 1.0f - gx, ly = 1.0f - gy, lz = 1.0f - gz;
    float i1x = __builtin_fminf(gx, lz), i1y = __builtin_fminf(gy, lx), i1z = __builtin_fminf(gz, ly);
    float i2x = __builtin_fmaxf(gx, lz), i2y = __builtin_fmaxf(gy, lx), i2z = __builtin_fmaxf(gz, ly);

    ix = mod289_ref(ix); iy = mod289_ref(iy); iz = mod289_ref(iz);

    float acc[4];
    float mk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float ox = k == 0 ? 0.0f : k == 1 ? i1x : k == 2 ? i2x : 1.0f;
        float oy = k == 0 ? 0.0f : k == 1 ? i1y : k == 2 ? i2y : 1.0f;
        float oz = k == 0 ? 0.0f : k == 1 ? i1z : k == 2 ? i2z : 1.0f;
        float cx, cy, cz;
        if (k == 0) { cx = ax; cy = ay; cz = az; }
        else if (k == 1) { cx = ax - i1x + kC6; cy = ay - i1y + kC6; cz = az - i1z + kC6; }
        else if (k == 2) { cx = ax - i2x + kC3; cy = ay - i2y + kC3; cz = az - i2z + kC3; }
        else { cx = ax - 0.5f; cy = ay - 0.5f; cz = az - 0.5f; }
        float p = permute_ref(permute_ref(permute_ref(iz + oz) + iy + oy) + ix + ox);
        float px, py, pz;
        gradient_ref(p, px, py, pz);
        float m = __builtin_fmaxf(0.6f - (cx * cx + cy * cy + cz * cz), 0.0f);
        m = m * m;
        mk[k] = m * m;
        acc[k] = px * cx + py * cy + pz * cz;
    }
    return 42.0f * (mk[0] * acc[0] + mk[1] * acc[1] + mk[2] * acc[2] + mk[3] * acc[3]);
}

// src/logic.frag:41-43
TH_HD float vary(float base, float offset, float variance) { return base + (offset * variance * base); }

// ---- exact small-integer hash (guarded domain only) -------------------------
// For integer-valued x with |(34x+1)x| < 2^24 every intermediate of permute()
// is an exactly representable integer, so fusing the multiply-adds cannot
// change any bit: fma(x,34,1) == x*34+1 and fma(-289,q,t) == t - q*289.
TH_HD float permute_int(float x)
{
    float t = th_fma(x, 34.0f, 1.0f) * x;
    return th_fma(-289.0f, th_floor(t * kInv289), t);
}
TH_HD float mod289_int(float x) { return th_fma(-289.0f, th_floor(x * kInv289), x); }

}  // namespace th
