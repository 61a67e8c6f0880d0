// th_logic.hpp - the per-particle update as device code: logic.frag::main (src/logic.frag:41-101) with its helpers -
// vary(), the two snoise(vec3) calls of glsl-noise, flowAtScreenPos / posToUV (NEAREST, CLAMP), get() decay - and the
// packed-state codec.  Included by the kernels that step particles: th_kernels.hip (logic_kernel, logic_fused_kernel,
// logic_sorted_kernel ...) and th_bins.hip (the frame pass: one step and the draw() emit of a slot in one go).
// Compiled with -ffp-contract=off in every unit (Makefile): a*b+c stays two rounded fp32 operations.
#pragma once
#include "th_kernels.hpp"
#include "th_math.hpp"

namespace th {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

// streaming (read-once / write-once) 16-byte accesses of the state ring
TH_D float4 load_stream(const float4 *p)
{
    v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
TH_D void store_stream(float4 *p, float4 a)
{
    v4f v = {a.x, a.y, a.z, a.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
}

// ---------------------------------------------------------------------------
// Packed state (TH_STATE_F16, config C5): 8 bytes per particle instead of 16.
//   word 0: position, two SNORM16 over [-2, 2): q = rint(clamp(p * 16384, -32767, 32767));
//           (-32768, -32768) = inert (src/const/inert.glsl), (-32768, 0) = NaN position
//   word 1: velocity, two IEEE fp16 (round to nearest even)
// The integrator arithmetic is unchanged (fp32, exact or fast) on the DECODED values; only the
// storage is quantised.  The reference has no half path: this encoding is defined by this build
// (DESIGN.md "packed state") and mirrored for the tests in tests/helpers.py.
// ---------------------------------------------------------------------------
TH_D float4 unpack_state(uint2 w)
{
    int xs = (int)(short)(w.x & 0xffffu), ys = (int)(short)(w.x >> 16);
    _Float16 hx, hy;
    unsigned short ux = (unsigned short)(w.y & 0xffffu), uy = (unsigned short)(w.y >> 16);
    __builtin_memcpy(&hx, &ux, 2); __builtin_memcpy(&hy, &uy, 2);
    float4 s;
    s.z = (float)hx; s.w = (float)hy;
    if (xs == -32768) {
        if (ys == -32768) { s.x = kInert; s.y = kInert; }
        else { s.x = __builtin_nanf(""); s.y = __builtin_nanf(""); }
    } else {
        s.x = (float)xs * 6.103515625e-05f; s.y = (float)ys * 6.103515625e-05f;       // exact: / 16384
    }
    return s;
}

TH_D uint2 pack_state(float4 s)
{
    unsigned px;
    if (!(s.x != kInert || s.y != kInert)) px = 0x80008000u;
    else if (s.x != s.x || s.y != s.y) px = 0x00008000u;
    else {
        int xs = (int)__builtin_rintf(__builtin_amdgcn_fmed3f(s.x * 16384.0f, -32767.0f, 32767.0f));
        int ys = (int)__builtin_rintf(__builtin_amdgcn_fmed3f(s.y * 16384.0f, -32767.0f, 32767.0f));
        px = ((unsigned)xs & 0xffffu) | ((unsigned)ys << 16);
    }
    _Float16 hx = (_Float16)s.z, hy = (_Float16)s.w;
    unsigned short ux, uy;
    __builtin_memcpy(&ux, &hx, 2); __builtin_memcpy(&uy, &hy, 2);
    return make_uint2(px, (unsigned)ux | ((unsigned)uy << 16));
}
// unpack_state(pack_state(s)) without the 8 bytes in between: what a packed ring holds of a state, as the next fused step reads it.
// The quantised position is an integer-valued float in [-32767, 32767]: the stored short converts back to exactly that float, and
// never to the sentinel -32768; the velocity goes through fp16 and back; inert and NaN positions come back as unpack_state gives
// them.  (Sixteen instructions fewer per fused step than the words packed, stored in registers and unpacked: round 6.)
TH_D float4 quantize_state(float4 s)
{
    float4 r;
    r.z = (float)(_Float16)s.z; r.w = (float)(_Float16)s.w;
    if (!(s.x != kInert || s.y != kInert)) { r.x = kInert; r.y = kInert; }
    else if (s.x != s.x || s.y != s.y) { r.x = __builtin_nanf(""); r.y = __builtin_nanf(""); }
    else {
        r.x = __builtin_rintf(__builtin_amdgcn_fmed3f(s.x * 16384.0f, -32767.0f, 32767.0f)) * 6.103515625e-05f;
        r.y = __builtin_rintf(__builtin_amdgcn_fmed3f(s.y * 16384.0f, -32767.0f, 32767.0f)) * 6.103515625e-05f;
    }
    return r;
}
// ---------------------------------------------------------------------------
// Reference-order evaluation of one texel: any input, any uniform set.  Used
// for lanes outside the fast path's proven domain and by the generic kernel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float4 logic_texel_ref(const LogicParams &p, uint32_t x, uint32_t y_global,
                                                float4 st, uint32_t local_index, float time)
{
    const th_logic_uniforms &u = p.u;
    float fcx = (float)x + 0.5f, fcy = (float)y_global + 0.5f;      // gl_FragCoord.xy
    float uvx = fcx / p.wf, uvy = fcy / p.hf;                        // :46
    float posx = st.x, posy = st.y, velx = st.z, vely = st.w;
    if (!(posx != kInert || posy != kInert)) return st;              // :52 (vec2 != : any component)

    float i = (fcx + (fcy * p.wf)) / (p.wf * p.hf);                  // :57-58
    float nscale = vary(u.noiseScale, i, u.varyNoiseScale);
    float nx = posx * nscale, ny = posy * nscale;                    // :62
    float ntime = time * vary(u.noiseSpeed, i, u.varyNoiseSpeed);  // :65
    float wx = snoise_ref(nx, ny, uvx + ntime);                      // :67
    float wy = snoise_ref(nx, ny, uvy + ntime + 1234.5678f);         // :68

    // :75 flowAtScreenPos(pos*viewSize): posToUV then one NEAREST/CLAMP tap, levels = 1
    float sx = posx * u.viewSize[0], sy = posy * u.viewSize[1];
    float fu = 0.0f + (1.0f * (sx + 1.0f)) / 2.0f;
    float fv = 0.0f + (1.0f * (sy + 1.0f)) / 2.0f;
    int tx = (int)__builtin_amdgcn_fmed3f(th_floor(fu * p.fwf), 0.0f, p.fwm1);
    int ty = (int)__builtin_amdgcn_fmed3f(th_floor(fv * p.fhf), 0.0f, p.fhm1);
    float4 ft = p.flow[(size_t)ty * p.fw + tx];
    float k = __builtin_fmaxf(0.0f, 1.0f - ((time - ft.z) * u.flowDecay));   // src/flow/get.glsl:4
    float ffx = (0.0f + ft.x * k * 1.0f) / 1.0f, ffy = (0.0f + ft.y * k * 1.0f) / 1.0f;

    float vfw = vary(u.forceWeight, i, u.varyForce);
    float vflw = vary(u.flowWeight, i, u.varyFlow);
    float vnw = vary(u.noiseWeight, i, u.varyNoise);
    float nvx = (velx * u.damping * u.dt) + (vfw * ((ffx * u.dt * vflw) + (wx * u.dt * vnw)));   // :79-82
    float nvy = (vely * u.damping * u.dt) + (vfw * ((ffy * u.dt * vflw) + (wy * u.dt * vnw)));

    float4 tg = p.targets[local_index];
    float vtg = vary(u.target, i, u.varyTarget);
    nvx += (tg.x - posx) * vtg;                                      // :85
    nvy += (tg.y - posy) * vtg;

    float speed = __builtin_sqrtf(nvx * nvx + nvy * nvy);            // :92 (correctly rounded)
    float r = __builtin_fminf(speed, u.speedLimit) / speed;          // :94 (0/0 -> NaN, as the reference)
    nvx *= r; nvy *= r;
    return make_float4(posx + nvx, posy + nvy, nvx, nvy);            // :97,100
}

// ---------------------------------------------------------------------------
// Simplex noise on the guarded domain, gradient + normalisation from the LDS
// table (snoise_corners + snoise_finish).  Bit-identical to snoise_ref for |v| < kNoiseDomain
// when !FAST.  sxy = vx*C.y + vy*C.y is shared by the two evaluations of one particle.
// ---------------------------------------------------------------------------
// Lattice part of one evaluation: corner offsets and the table index of each corner's gradient.
struct NoiseCorners {
    float ax, ay, az, bx, by, bz, cx, cy, cz, dx, dy, dz;
    int j0, j1, j2, j3;
};

// ---- hash stages through LDS tables (issue-bound launches: the fused integrator) --------------------------
// The first two permutation stages read the polynomial's own values from LDS instead of evaluating them, and the
// whole index chain runs on integers that already are byte offsets, so no stage needs an offset multiply:
//   permA[k] = 4 * permute_int(k)               k in [0, 290]: stage z (argument iz, iz + 1), 4 * value = offset unit of permB
//   permB[k] = 16 * (permute_int(k) - kLutMin)  k in [0, 580]: stage y; + 16 * (ix + i) = byte offset of the gradient entry
// Both are filled by the kernel with permute_int itself; mod289_int() results are exact integers in [0, 289].
constexpr int kPermA = 292, kPermB = 584;                    // entries (multiples of 4)
constexpr int kHashVec = (kPermA + kPermB) / 4;              // float4 slots in front of the gradient table
struct HashTables {
    const uint32_t *permA, *permB;
};

// [permA | permB | gradient table] as one block: computed once per context into global memory (hash_tables_kernel, with
// permute_int itself), copied into LDS by every workgroup (803 float4: three loads per thread instead of ~100 VALU
// instructions per thread to evaluate the polynomial - a fused launch has one workgroup per 256 particles)
TH_D void fill_hash_tables(float4 *smem, const float4 *lut_global)
{
    const float4 *block = lut_global - kHashVec;        // (LogicParams::lut points at the gradient table inside the block)
    for (int k = threadIdx.x; k < kHashVec + kLutSize; k += 256) smem[k] = block[k];
}


template <bool FAST>
TH_D NoiseCorners snoise_corners_tab(float vx, float vy, float vz, float sxy, const HashTables &T)
{
    NoiseCorners n;
    float s = mad<FAST>(vz, kC3, sxy);
    float ix = th_floor(vx + s), iy = th_floor(vy + s), iz = th_floor(vz + s);
    float t = mad<FAST>(iz, kC6, mad<FAST>(iy, kC6, ix * kC6));
    float ax = (vx - ix) + t, ay = (vy - iy) + t, az = (vz - iz) + t;

    // Traversal order masks as in snoise_corners.  Here the second and third corner offsets are selected instead of
    // subtracted: x0 - i1 with i1 in {0, 1} is x0 or x0 - 1 (exactly, signed zeros included), and x0 - 1 serves both
    // corners; the x / y steps also come out as index increments (0 or one table entry) for the hash chain below.
    const float ax1 = ax - 1.0f, ay1 = ay - 1.0f, az1 = az - 1.0f;
    float bx, by, bz, cx, cy, cz;
    uint32_t e1x, e2x, e1y, e2y;
    unsigned long long mz1, mz2;
    {
        unsigned long long l1, l2, l3, m;
        asm("v_cmp_lt_f32 %[l1], %[ax], %[ay]\n\t"
            "v_cmp_lt_f32 %[l2], %[ay], %[az]\n\t"
            "v_cmp_lt_f32 %[l3], %[az], %[ax]\n\t"
            "s_andn2_b64 %[m], %[l3], %[l1]\n\t"     "v_cndmask_b32 %[bx], %[ax], %[ax1], %[m]\n\t"   "v_cndmask_b32 %[e1x], 0, 16, %[m]\n\t"
            "s_andn2_b64 %[m], %[l1], %[l2]\n\t"     "v_cndmask_b32 %[by], %[ay], %[ay1], %[m]\n\t"   "v_cndmask_b32 %[e1y], 0, 4, %[m]\n\t"
            "s_andn2_b64 %[mz1], %[l2], %[l3]\n\t"   "v_cndmask_b32 %[bz], %[az], %[az1], %[mz1]\n\t"
            "s_orn2_b64 %[m], %[l3], %[l1]\n\t"      "v_cndmask_b32 %[cx], %[ax], %[ax1], %[m]\n\t"   "v_cndmask_b32 %[e2x], 0, 16, %[m]\n\t"
            "s_orn2_b64 %[m], %[l1], %[l2]\n\t"      "v_cndmask_b32 %[cy], %[ay], %[ay1], %[m]\n\t"   "v_cndmask_b32 %[e2y], 0, 4, %[m]\n\t"
            "s_orn2_b64 %[mz2], %[l2], %[l3]\n\t"    "v_cndmask_b32 %[cz], %[az], %[az1], %[mz2]"
            : [l1] "=&s"(l1), [l2] "=&s"(l2), [l3] "=&s"(l3), [m] "=&s"(m), [mz1] "=&s"(mz1), [mz2] "=&s"(mz2),
              [bx] "=&v"(bx), [by] "=&v"(by), [bz] "=&v"(bz), [cx] "=&v"(cx), [cy] "=&v"(cy), [cz] "=&v"(cz),
              [e1x] "=&v"(e1x), [e2x] "=&v"(e2x), [e1y] "=&v"(e1y), [e2y] "=&v"(e2y)
            : [ax] "v"(ax), [ay] "v"(ay), [az] "v"(az), [ax1] "v"(ax1), [ay1] "v"(ay1), [az1] "v"(az1)
            : "scc");
    }
    n.ax = ax; n.ay = ay; n.az = az;
    n.bx = bx + kC6; n.by = by + kC6; n.bz = bz + kC6;
    n.cx = cx + kC3; n.cy = cy + kC3; n.cz = cz + kC3;
    n.dx = ax - 0.5f; n.dy = ay - 0.5f; n.dz = az - 0.5f;

    const uint32_t xi = (uint32_t)mod289_int(ix), yi = (uint32_t)mod289_int(iy), zi = (uint32_t)mod289_int(iz);
    const uint32_t *pa = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T.permA) + (zi << 2));
    const uint32_t pz0 = pa[0], pz1 = pa[1];                    // 4 * permute(iz), 4 * permute(iz + 1)
    uint32_t sel1, sel2;
    asm("v_cndmask_b32 %0, %2, %3, %4\n\tv_cndmask_b32 %1, %2, %3, %5"
        : "=&v"(sel1), "=&v"(sel2) : "v"(pz0), "v"(pz1), "s"(mz1), "s"(mz2));
    const uint32_t y4 = yi << 2;
    auto stage_y = [&](uint32_t off) { return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T.permB) + off); };
    const uint32_t q0 = stage_y(pz0 + y4), q1 = stage_y(sel1 + y4 + e1y), q2 = stage_y(sel2 + y4 + e2y),
                   q3 = stage_y(pz1 + y4 + 4u);
    const uint32_t x16 = xi << 4;
    n.j0 = (int)(q0 + x16); n.j1 = (int)(q1 + x16 + e1x); n.j2 = (int)(q2 + x16 + e2x); n.j3 = (int)(q3 + x16 + 16u);
    return n;
}

// gradient entry at a byte offset produced by snoise_corners_tab
TH_D float4 lut_at_offset(const float4 *lut, int off)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(lut) + off);
}

template <bool FAST>
TH_D NoiseCorners snoise_corners(float vx, float vy, float vz, float sxy)
{
    NoiseCorners n;
    // first corner
    float s = mad<FAST>(vz, kC3, sxy);
    float ix = th_floor(vx + s), iy = th_floor(vy + s), iz = th_floor(vz + s);
    float t = mad<FAST>(iz, kC6, mad<FAST>(iy, kC6, ix * kC6));
    float ax = (vx - ix) + t, ay = (vy - iy) + t, az = (vz - iz) + t;

    // simplex traversal order: g = step(x0.yzx, x0.xyz) (g.x = !(x0.x < x0.y) ...), i1 = min(g, 1-g.zxy),
    // i2 = max(g, 1-g.zxy).  Written on the three "less-than" masks only (values are finite here):
    // i1 = (g.x & !g.z, ...) = (l3 & !l1, l1 & !l2, l2 & !l3), i2 = (g.x | !g.z, ...) = (l3 | !l1, ...)
    // Three compares, the six masks by scalar-unit algebra (s_andn2 / s_orn2), selects from the masks.
    // (hipcc would issue a second vector compare for every negated mask.)
    float i1x, i1y, i1z, i2x, i2y, i2z;
    unsigned long long mz1, mz2;           // b1z, b2z: select pz1 / pz0 below
    {
        unsigned long long l1, l2, l3, t;
        asm("v_cmp_lt_f32 %[l1], %[ax], %[ay]\n\t"
            "v_cmp_lt_f32 %[l2], %[ay], %[az]\n\t"
            "v_cmp_lt_f32 %[l3], %[az], %[ax]\n\t"
            "s_andn2_b64 %[t], %[l3], %[l1]\n\t"     "v_cndmask_b32 %[i1x], 0, 1.0, %[t]\n\t"
            "s_andn2_b64 %[t], %[l1], %[l2]\n\t"     "v_cndmask_b32 %[i1y], 0, 1.0, %[t]\n\t"
            "s_andn2_b64 %[mz1], %[l2], %[l3]\n\t"   "v_cndmask_b32 %[i1z], 0, 1.0, %[mz1]\n\t"
            "s_orn2_b64 %[t], %[l3], %[l1]\n\t"      "v_cndmask_b32 %[i2x], 0, 1.0, %[t]\n\t"
            "s_orn2_b64 %[t], %[l1], %[l2]\n\t"      "v_cndmask_b32 %[i2y], 0, 1.0, %[t]\n\t"
            "s_orn2_b64 %[mz2], %[l2], %[l3]\n\t"    "v_cndmask_b32 %[i2z], 0, 1.0, %[mz2]"
            : [l1] "=&s"(l1), [l2] "=&s"(l2), [l3] "=&s"(l3), [t] "=&s"(t), [mz1] "=&s"(mz1), [mz2] "=&s"(mz2),
              [i1x] "=&v"(i1x), [i1y] "=&v"(i1y), [i1z] "=&v"(i1z), [i2x] "=&v"(i2x), [i2y] "=&v"(i2y), [i2z] "=&v"(i2z)
            : [ax] "v"(ax), [ay] "v"(ay), [az] "v"(az)
            : "scc");
    }

    n.ax = ax; n.ay = ay; n.az = az;
    n.bx = (ax - i1x) + kC6; n.by = (ay - i1y) + kC6; n.bz = (az - i1z) + kC6;
    n.cx = (ax - i2x) + kC3; n.cy = (ay - i2y) + kC3; n.cz = (az - i2z) + kC3;
    n.dx = ax - 0.5f; n.dy = ay - 0.5f; n.dz = az - 0.5f;

    // permutation hash: exact small-integer arithmetic (th_math.hpp)
    ix = mod289_int(ix); iy = mod289_int(iy); iz = mod289_int(iz);
    float pz0 = permute_int(iz), pz1 = permute_int(iz + 1.0f);       // z offsets are only ever 0 or 1
    float q0 = permute_int(pz0 + iy);
    float sel1, sel2;
    asm("v_cndmask_b32 %0, %2, %3, %4\n\tv_cndmask_b32 %1, %2, %3, %5"
        : "=&v"(sel1), "=&v"(sel2) : "v"(pz0), "v"(pz1), "s"(mz1), "s"(mz2));
    float q1 = permute_int((sel1 + iy) + i1y);
    float q2 = permute_int((sel2 + iy) + i2y);
    float q3 = permute_int((pz1 + iy) + 1.0f);
    // Table index without a float->int conversion: the last-stage argument is a small integer, so
    // adding 2^23 (+ the table bias) leaves it in the low mantissa bits of the sum; every addition
    // stays exact (all values are integers below 2^24).  lut_index() turns the bits into an LDS offset.
    const float ixm = ix + (8388608.0f - (float)kLutMin);
    n.j0 = __float_as_int(q0 + ixm);
    n.j1 = __float_as_int((q1 + ixm) + i1x);
    n.j2 = __float_as_int((q2 + ixm) + i2x);
    n.j3 = __float_as_int((q3 + ixm) + 1.0f);
    return n;
}

// table entry of a magic-number index produced by snoise_corners: low 24 bits = entry number
TH_D float4 lut_at(const float4 *lut, int magic)
{
    // v_mul_u32_u24 multiplies the LOW 24 BITS of its operands: one full-rate op strips the exponent
    // and scales to the byte offset (written as asm so that the masking is not optimised away)
    unsigned off;
    asm("v_mul_u32_u24 %0, %1, 16" : "=v"(off) : "v"(magic));
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(lut) + off);
}

// Radial falloff and gradient dot products, given the four table entries.
template <bool FAST>
TH_D float snoise_finish(const NoiseCorners &n, float4 g0, float4 g1, float4 g2, float4 g3)
{
    float m0 = __builtin_fmaxf(0.6f - mad<FAST>(n.az, n.az, mad<FAST>(n.ay, n.ay, n.ax * n.ax)), 0.0f);
    float m1 = __builtin_fmaxf(0.6f - mad<FAST>(n.bz, n.bz, mad<FAST>(n.by, n.by, n.bx * n.bx)), 0.0f);
    float m2 = __builtin_fmaxf(0.6f - mad<FAST>(n.cz, n.cz, mad<FAST>(n.cy, n.cy, n.cx * n.cx)), 0.0f);
    float m3 = __builtin_fmaxf(0.6f - mad<FAST>(n.dz, n.dz, mad<FAST>(n.dy, n.dy, n.dx * n.dx)), 0.0f);
    m0 *= m0; m1 *= m1; m2 *= m2; m3 *= m3;
    m0 *= m0; m1 *= m1; m2 *= m2; m3 *= m3;
    float d0 = mad<FAST>(g0.z, n.az, mad<FAST>(g0.y, n.ay, g0.x * n.ax));
    float d1 = mad<FAST>(g1.z, n.bz, mad<FAST>(g1.y, n.by, g1.x * n.bx));
    float d2 = mad<FAST>(g2.z, n.cz, mad<FAST>(g2.y, n.cy, g2.x * n.cx));
    float d3 = mad<FAST>(g3.z, n.dz, mad<FAST>(g3.y, n.dy, g3.x * n.dx));
    float r = 42.0f * mad<FAST>(m3, d3, mad<FAST>(m2, d2, mad<FAST>(m1, d1, m0 * d0)));
    // Keep the unused .w of the four entries alive until here (zero instructions; tied to the result so that no
    // early wait is forced): the table reads then stay ds_read_b128 (4 lane groups over 64 banks) instead of being
    // narrowed to ds_read_b96 (8 lane groups over 32 banks: 20 LDS cycles per read on random entries, PMC-measured).
    asm("" : "+v"(r) : "v"(g0.w), "v"(g1.w), "v"(g2.w), "v"(g3.w));
    return r;
}

// ---------------------------------------------------------------------------
// The integrator.  Template switches are uniform-derived (chosen by the host
// per launch); every lane the specialised path cannot prove in-domain falls
// back to logic_texel_ref, so the result is the reference's in all cases.
//   NOISE  noiseWeight != 0
//   TARGET target != 0 or the targets texture holds a non-finite value
//   POW2   dataRes.x, dataRes.y powers of two: `/dataRes` == `*(1/dataRes)` exactly
//   DECODED the flow tap reads the per-step decoded float2 plane (8 B) instead of RGBA32F (16 B)
// ---------------------------------------------------------------------------
// One particle: state texel `st` of particle `pid` (= texel index in this context's rows).
constexpr int kTileShift = 5;            // 32 x 32-texel tiles of the flow field: the key of the tile-sorted slot order
template <bool FAST, bool NOISE, bool TARGET, bool POW2, bool DECODED, bool PTAB = false>
TH_D float4 integrate(const LogicParams &p, const float4 *lut, float4 st, uint32_t pid, float time,
                      const HashTables *tabs = nullptr)
{
    const th_logic_uniforms &u = p.u;
    float posx = st.x, posy = st.y, velx = st.z, vely = st.w;

    uint32_t x, y;
    if constexpr (POW2) { x = pid & (p.width - 1u); y = pid >> p.log2w; }
    else { y = pid / p.width; x = pid - y * p.width; }
    y += p.row0;

    // two compares (not max): a NaN in either component must fail the test.  pos_bound < |inert| (host), so inert
    // particles (src/logic.frag:52) fail it too and are passed through by logic_texel_ref.
    bool in_domain = __builtin_fabsf(posx) < p.pos_bound && __builtin_fabsf(posy) < p.pos_bound;
    if (__builtin_expect(!in_domain, 0)) {
        if (!(posx != kInert || posy != kInert)) return st;              // inert: pass through (src/logic.frag:52)
        // A NaN or infinite position component makes every output component NaN in the reference: the first noise
        // coordinate v = pos * noiseScale' is NaN or Inf, s = dot(v, C.yyy) and i = floor(v + s) follow, x0 = v - i + t
        // is NaN (Inf - Inf), the gradient dot products are NaN and 42 * dot(m, NaN) is NaN whatever m is; wander
        // is multiplied (never skipped) into newVel (src/logic.frag:79-82), min(NaN, limit)/NaN keeps it NaN (:94)
        // and newPos = pos + newVel (:97).  No arithmetic is needed to produce that (NaN payloads are not pinned,
        // DESIGN.md 4); particles that went 0/0 -> NaN (:92-94) would otherwise hold their whole wave on the
        // reference-order path below for good.
        if (!(__builtin_fabsf(posx) < __builtin_inff()) || !(__builtin_fabsf(posy) < __builtin_inff())) {
            const float q = __builtin_nanf("");
            return make_float4(q, q, q, q);
        }
        return logic_texel_ref(p, x, y, st, pid, time);
    }

    float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;
    float uvx, uvy, i;
    if constexpr (POW2) {
        uvx = fcx * p.inv_w; uvy = fcy * p.inv_h;
        i = (fcx + (fcy * p.wf)) * p.inv_wh;
    } else if constexpr (FAST) {
        uvx = fcx * p.inv_w; uvy = fcy * p.inv_h;
        i = mad<true>(fcy, p.wf, fcx) * p.inv_wh;
    } else {
        uvx = fcx / p.wf; uvy = fcy / p.hf;
        i = (fcx + (fcy * p.wf)) / (p.wf * p.hf);
    }

    // flow tap (issued first: its latency hides under the noise arithmetic)
    float sx = posx * u.viewSize[0], sy = posy * u.viewSize[1];
    // posToUV = (1*(v+1))/2, then *size: halving is exact, so ((v+1)*0.5)*size == (v+1)*(0.5*size) - one rounding
    // either way (half_fw = 0.5*fw from the host; a denormal (v+1)/2 lands in texel 0 on both routes)
    int tx = (int)__builtin_amdgcn_fmed3f((sx + 1.0f) * p.half_fw, 0.0f, p.fwm1);      // trunc == floor on [0, n-1]
    int ty = (int)__builtin_amdgcn_fmed3f((sy + 1.0f) * p.half_fh, 0.0f, p.fhm1);
    const int texel = ty * p.fw + tx;
    float ffx, ffy;      // getFlow(): data.xy * max(0, 1 - (time - data.z)*decay), src/flow/get.glsl:4
    float4 ft;
    if constexpr (DECODED) { float2 d = p.flow_dec[texel]; ffx = d.x; ffy = d.y; }
    else if (p.flow3) { const float *f3 = p.flow3 + 3u * (uint32_t)texel; ft = make_float4(f3[0], f3[1], f3[2], 0.0f); }     // (uniform branch)
    else ft = p.flow[texel];

    float wxs = 0.0f, wys = 0.0f;   // (wander * dt) * vary(noiseWeight)
    if constexpr (NOISE) {
        float nscale = vary(u.noiseScale, i, u.varyNoiseScale);
        float nx = posx * nscale, ny = posy * nscale;
        float ntime = time * vary(u.noiseSpeed, i, u.varyNoiseSpeed);
        float sxy = mad<FAST>(ny, kC3, nx * kC3);
        // both lattice parts first, so that all eight table reads are in flight together
        NoiseCorners na, nb;
        float4 a0, a1, a2, a3, b0, b1, b2, b3;
        if constexpr (PTAB) {
            na = snoise_corners_tab<FAST>(nx, ny, uvx + ntime, sxy, *tabs);
            nb = snoise_corners_tab<FAST>(nx, ny, (uvy + ntime) + 1234.5678f, sxy, *tabs);
            a0 = lut_at_offset(lut, na.j0); a1 = lut_at_offset(lut, na.j1); a2 = lut_at_offset(lut, na.j2); a3 = lut_at_offset(lut, na.j3);
            b0 = lut_at_offset(lut, nb.j0); b1 = lut_at_offset(lut, nb.j1); b2 = lut_at_offset(lut, nb.j2); b3 = lut_at_offset(lut, nb.j3);
        } else {
            na = snoise_corners<FAST>(nx, ny, uvx + ntime, sxy);
            nb = snoise_corners<FAST>(nx, ny, (uvy + ntime) + 1234.5678f, sxy);
            a0 = lut_at(lut, na.j0); a1 = lut_at(lut, na.j1); a2 = lut_at(lut, na.j2); a3 = lut_at(lut, na.j3);
            b0 = lut_at(lut, nb.j0); b1 = lut_at(lut, nb.j1); b2 = lut_at(lut, nb.j2); b3 = lut_at(lut, nb.j3);
        }
        float wx = snoise_finish<FAST>(na, a0, a1, a2, a3);
        float wy = snoise_finish<FAST>(nb, b0, b1, b2, b3);
        float vnw = vary(u.noiseWeight, i, u.varyNoise);
        wxs = (wx * u.dt) * vnw; wys = (wy * u.dt) * vnw;
    }

    if constexpr (!DECODED) {
        float k = __builtin_fmaxf(0.0f, 1.0f - ((time - ft.z) * u.flowDecay));
        ffx = ft.x * k; ffy = ft.y * k;
    }
    float vflw = vary(u.flowWeight, i, u.varyFlow);
    float fxs = (ffx * u.dt) * vflw, fys = (ffy * u.dt) * vflw;
    float vfw = vary(u.forceWeight, i, u.varyForce);
    float nvx, nvy;
    if constexpr (NOISE) {
        nvx = mad<FAST>(vfw, fxs + wxs, (velx * u.damping) * u.dt);
        nvy = mad<FAST>(vfw, fys + wys, (vely * u.damping) * u.dt);
    } else {
        // wander*dt*vary(0) is a signed zero here: adding it cannot change a non-zero
        // sum, and a zero sum ends in 0/0 = NaN either way (DESIGN.md "skipped terms")
        nvx = mad<FAST>(vfw, fxs, (velx * u.damping) * u.dt);
        nvy = mad<FAST>(vfw, fys, (vely * u.damping) * u.dt);
    }
    if constexpr (TARGET) {
        float4 tg = p.targets[pid];
        float vtg = vary(u.target, i, u.varyTarget);
        nvx = mad<FAST>(tg.x - posx, vtg, nvx);
        nvy = mad<FAST>(tg.y - posy, vtg, nvy);
    }

    // speed clamp: r = min(speed, limit)/speed is exactly 1 when 0 < speed <= limit,
    // i.e. when 0 < s2 <= s2_cap (sqrt_rn is monotonic; s2_cap from the host).
    float s2 = mad<FAST>(nvy, nvy, nvx * nvx);
    if (!(s2 > 0.0f && s2 <= p.s2_cap)) {
        float r;
        if constexpr (FAST) {
            r = __builtin_fminf(1.0f, u.speedLimit * __builtin_amdgcn_rsqf(s2));
            if (!(s2 > 0.0f)) r = __builtin_nanf("");
        } else {
            float speed = __builtin_sqrtf(s2);
            r = __builtin_fminf(speed, u.speedLimit) / speed;
        }
        nvx *= r; nvy *= r;
    }
    return make_float4(posx + nvx, posy + nvy, nvx, nvy);
}

}  // namespace th
