// th_deposit.hip - flow deposit: the particle lines of Tendrils.draw() blended into the flow field.
//
// Replaces the flow pass of draw() (src/index.js:278-303): particles.draw(render uniforms, gl.LINES) with the flow
// shader (src/flow/index.vert -> vert/main.vert:10-17, apply/state.glsl:5-16, index.frag) into the flow FBO, blend
// SRC_ALPHA / ONE_MINUS_SRC_ALPHA (src/index.js:267-268).  Semantics (vertex stream and pairing, width-1 line =
// hexagon of the two endpoint diamonds, clip-space clipping, 1/16-texel snapping, ceil() scan conversion, varying
// linear along the snapped endpoints, blending in stream order; a pair with an inert vertex draws nothing) are the
// ones pinned against captures of the reference in tests/golden/deposit_*.npz; every arithmetic step below is
// written in the same order and precision as the checker's restatement so that both agree bit for bit.
//
// GL blends fragments in primitive order, which a parallel machine has to reconstruct:
//   1. deposit_raster_kernel: one thread per line (workgroups walk the particle texture row-major: coalesced state
//      reads), rasterise ONCE: the line's fragment count and - lines of up to eight fragments, nearly all of them:
//      particles move about a texel per step - the texels themselves go into a 32-byte record per line; the lines
//      that need clipping or 64-bit edges, and the long ones, are listed for slower kernels
//   2. exclusive scan of the counts in STREAM order (column-major over the row-major count array: column sums per
//      64-row block, one small scan, column prefixes) -> every line's first slot: the fragment array is in stream order
//   3. deposit_emit_kernel: per line, the varying at the recorded texels (only the few long lines are rasterised
//      again) -> (texel, interpolated varying) in the line's slots
//   4. stable radix sort of the fragments by texel (th_sort.hip): each texel's fragments end up contiguous and still
//      in stream order
//   5. deposit_blend_kernel: the lane at the head of a texel's run blends it in order (long runs: the whole wave):
//      dst = src*a + dst*(1-a), exactly GL's order and arithmetic.
// No step depends on thread scheduling, and the cost does not depend on how crowded single texels are (the
// wake makes particles converge: thousands of fragments in one texel are normal after a few dozen frames).
#include "th_kernels.hpp"
#include "th_math.hpp"
#include <cstdlib>

namespace th {
namespace {

struct DepositVertex {
    bool live;
    float px, py;      // clip-space position (w = 1)
    float c[4];        // varying: (vel.x, vel.y, time, min(|vel|/speedLimit, 1)) - or the view pass's colour (mode 1)
    float c2[4];       // mode 2 (both passes of draw() in one): the view pass's colour beside the flow pass's varying
};

TH_D int dep_nearest(float u, int n)       // NEAREST + CLAMP_TO_EDGE on a float texture
{
    float f = th_floor(u * (float)n);
    if (!(f > 0.0f)) return 0;
    if (f > (float)(n - 1)) return n - 1;
    return (int)f;
}

// The view pass's vertex colour (src/render/index.vert:58-100): base colour + colour map + flow-aligned colour, each
// pre-multiplied and clamped, alpha scaled by the speed and a vignette.  Operation order as in the shader (and in the
// checker's restatement); sin(time*flowDecay) - a uniform-only expression, implementation-defined in GLSL - comes from
// the host (sin_term).  glsl-map: outMin + (outMax-outMin)*(v-inMin)/(inMax-inMin); mix(a, b, t) = a*(1-t) + b*t.
TH_D void dep_render_color(const DepositParams &p, float4 state, float uvx, float uvy, float (&c)[4])
{
    const float velx = state.z / p.speed_limit, vely = state.w / p.speed_limit;
    const float speed_rate = __builtin_fminf((velx * velx + vely * vely) / p.speed_alpha, 1.0f);
    float mapped[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (p.colormap) {       // uv*geomRes/dataRes, geomRes = [W, 2H]
        const float mu = uvx * (float)p.W / (float)p.W, mv = uvy * (float)(2u * p.H) / (float)p.H;
        const float4 m = p.colormap[(size_t)dep_nearest(mv, p.ch) * p.cw + dep_nearest(mu, p.cw)];
        mapped[0] = m.x; mapped[1] = m.y; mapped[2] = m.z; mapped[3] = m.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) mapped[k] = mapped[k] * p.colormap_alpha;
    const float al[3] = {velx * 1.0f + vely * 0.0f, velx * -0.5000000000000004f + vely * -0.8660254037844385f,
                         velx * -0.4999999999999998f + vely * 0.8660254037844387f};
    const float gbr[3] = {al[1] * (1.0f - p.flow_decay), al[2] * (1.0f - p.flow_decay), al[0] * (1.0f - p.flow_decay)};
    float flw[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float m = al[k] * (1.0f - p.sin_term) + gbr[k] * p.sin_term;
        flw[k] = p.flow_color[k] * (0.0f + (1.0f - 0.0f) * (m - -1.0f) / (1.0f - -1.0f));
    }
    flw[3] = p.flow_color[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = 0.0f;
    auto add = [&](const float *t) {
        const float a = t[3];
        const float pre[4] = {t[0] * a, t[1] * a, t[2] * a, a};
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = c[k] + __builtin_fminf(__builtin_fmaxf(pre[k], 0.0f), 1.0f);
    };
    add(p.base_color); add(mapped); add(flw);
    // vignette(pos, center = 0, limit = 1, curve = (0.2, 1, 1)) clamped to fadeRange (0.2, 1): src/filter/vignette.glsl:5-28
    const float amount = __builtin_fminf(1.0f - (__builtin_sqrtf(state.x * state.x + state.y * state.y) / 1.0f), 1.0f);
    const float ut = 1.0f - amount;
    const float bz = (0.2f * ut + 1.0f * amount) * ut + (1.0f * ut + 1.0f * amount) * amount;
    const float vg = __builtin_fmaxf(0.0f, bz);
    c[3] = c[3] * (speed_rate * __builtin_fminf(__builtin_fmaxf(vg, 0.2f), 1.0f));
}

// vertex j of column i of the stream Particles.generateLUT([W, 2H]) through src/state/state-at-frame.glsl:12-22
TH_D DepositVertex dep_fetch(const DepositParams &p, uint32_t i, uint32_t j)
{
    const int W = (int)p.W, H = (int)p.H;
    const float uvx = (float)((double)i * p.inv_x), uvy = (float)((double)j * p.inv_y);   // Float32Array of JS doubles
    const float near_index = uvy * (float)H;
    const float fl = th_floor(near_index);
    const float offset = near_index - fl;
    const float ly = fl / (float)H;
    const float4 *tex = offset > 0.25f ? p.cur : p.prev;
    int row = dep_nearest(ly, H) - (int)p.row0;              // row-band shard: the band (or its halo rows) must hold the row
    const int col = dep_nearest(uvx, W);
    // (one unconditional load from a selected address: a load under a branch is awaited at the join, and the second
    // vertex's load would only go out after the first had come back)
    const float4 *from = tex + (size_t)(row < 0 ? 0 : (row < (int)p.rows ? row : (int)p.rows - 1)) * W + col;
    if (!(row >= 0 && row < (int)p.rows)) {
        if (row == -1 && p.halo_lo) from = p.halo_lo + (offset > 0.25f ? 0 : W) + col;
        else if (row == (int)p.rows && p.halo_hi) from = p.halo_hi + (offset > 0.25f ? 0 : W) + col;
        else *p.oob = 1u;
    }
    const float4 t = *from;
    DepositVertex v;
    v.live = (t.x != kInert) || (t.y != kInert);
    v.px = t.x * p.view_x;
    v.py = t.y * p.view_y;
    if (p.mode != 1) {
        v.c[0] = t.z; v.c[1] = t.w; v.c[2] = p.time;
        v.c[3] = __builtin_fminf(__builtin_sqrtf(t.z * t.z + t.w * t.w) / p.speed_limit, 1.0f);
        if (p.mode == 2) dep_render_color(p, t, uvx, uvy, v.c2);
    } else dep_render_color(p, t, uvx, uvy, v.c);
    return v;
}

TH_D long long dep_ceil_div(long long a, long long b)     // b > 0
{
    long long q = a / b;
    if (a % b > 0) ++q;
    return q;
}

TH_D int dep_snap(float ndc, float scale, float offset) { return (int)__builtin_rintf(ndc * scale + offset); }

struct DepositLine {
    bool draws;
    bool short32;              // snapped endpoints less than 2^14 sixteenths apart: the varying's integers fit 32 bits
    DepositVertex a, b;
    int sx[2], sy[2];          // snapped endpoints (1/16 texel, texel centres at multiples of 16)
    int n;                     // polygon vertices after clipping
    int PX[12], PY[12];
};

// everything about line `id` (stream index = i*H + m) that does not depend on the texel, except the polygon
TH_D void dep_setup(const DepositParams &p, uint32_t i, uint32_t m, DepositLine &L)
{
    L.draws = false;
    L.short32 = false;
    L.n = 0;
    L.a = dep_fetch(p, i, 2u * m);
    L.b = dep_fetch(p, i, 2u * m + 1u);
    if (!L.a.live || !L.b.live) return;                                  // see the header: inert vertex = no line
    const float fw = (float)p.fw, fh = (float)p.fh;
    const float dx = (0.5f * fw) * (L.b.px - L.a.px), dy = (0.5f * fh) * (L.b.py - L.a.py);
    if (dx == 0.0f && dy == 0.0f) return;
    if (!(__builtin_fabsf(L.a.px) <= 1024.0f && __builtin_fabsf(L.a.py) <= 1024.0f &&
          __builtin_fabsf(L.b.px) <= 1024.0f && __builtin_fabsf(L.b.py) <= 1024.0f)) return;
    const float wx16 = 8.0f * fw, wy16 = 8.0f * fh;
    const float x0 = wx16 - 8.0f, y0 = wy16 - 8.0f;
    L.sx[0] = dep_snap(L.a.px, wx16, x0); L.sy[0] = dep_snap(L.a.py, wy16, y0);
    L.sx[1] = dep_snap(L.b.px, wx16, x0); L.sy[1] = dep_snap(L.b.py, wy16, y0);
    const int ex = L.sx[1] - L.sx[0], ey = L.sy[1] - L.sy[0];
    L.short32 = ex > -(1 << 14) && ex < (1 << 14) && ey > -(1 << 14) && ey < (1 << 14);
    L.draws = true;
}

// The width-1 line as the hexagon of its two endpoint diamonds, in clip space (six vertices, statically indexed).
// Returns kHexInside when all of it lies inside the view volume (it is rasterised as it stands), kHexOutside when all
// six vertices are beyond ONE of the four planes (the clipper would then leave nothing: 44 % of the lines of the C3
// bench, whose particles are spread over twice the view's height), else kHexClip.
enum { kHexInside = 0, kHexClip = 1, kHexOutside = 2 };
TH_D int dep_hexagon(const DepositParams &p, const DepositLine &L, float (&cx)[6], float (&cy)[6])
{
    const float fw = (float)p.fw, fh = (float)p.fh;
    const float dx = (0.5f * fw) * (L.b.px - L.a.px), dy = (0.5f * fh) * (L.b.py - L.a.py);
    const float hx = 0.5f / (0.5f * fw), hy = 0.5f / (0.5f * fh);      // half a texel in clip space
    const DepositVertex *vv[2] = {&L.a, &L.b};
#define TH_L(n, k) do { cx[n] = vv[k]->px - hx; cy[n] = vv[k]->py; } while (0)
#define TH_T(n, k) do { cx[n] = vv[k]->px; cy[n] = vv[k]->py + hy; } while (0)
#define TH_R(n, k) do { cx[n] = vv[k]->px + hx; cy[n] = vv[k]->py; } while (0)
#define TH_B(n, k) do { cx[n] = vv[k]->px; cy[n] = vv[k]->py - hy; } while (0)
    if (dx > dy) {
        if (dx > -dy) { TH_L(0, 0); TH_T(1, 0); TH_T(2, 1); TH_R(3, 1); TH_B(4, 1); TH_B(5, 0); }
        else          { TH_L(0, 1); TH_L(1, 0); TH_T(2, 0); TH_R(3, 0); TH_R(4, 1); TH_B(5, 1); }
    } else {
        if (dx > -dy) { TH_L(0, 0); TH_L(1, 1); TH_T(2, 1); TH_R(3, 1); TH_R(4, 0); TH_B(5, 0); }
        else          { TH_L(0, 1); TH_T(1, 1); TH_T(2, 0); TH_R(3, 0); TH_B(4, 0); TH_B(5, 1); }
    }
#undef TH_L
#undef TH_T
#undef TH_R
#undef TH_B
    bool inside = true, out0 = true, out1 = true, out2 = true, out3 = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const bool i0 = 1.0f + cx[k] >= 0.0f, i1 = 1.0f - cx[k] >= 0.0f, i2 = 1.0f - cy[k] >= 0.0f, i3 = 1.0f + cy[k] >= 0.0f;
        inside = inside && i0 && i1 && i2 && i3;
        out0 = out0 && !i0; out1 = out1 && !i1; out2 = out2 && !i2; out3 = out3 && !i3;
    }
    return inside ? kHexInside : ((out0 || out1 || out2 || out3) ? kHexOutside : kHexClip);
}

// the hexagon snapped to the 1/16-texel grid: the polygon of a line that needs no clipping
TH_D void dep_snap_hexagon(const DepositParams &p, const float (&cx)[6], const float (&cy)[6], int (&PX)[6], int (&PY)[6])
{
    const float wx16 = 8.0f * (float)p.fw, wy16 = 8.0f * (float)p.fh;
    const float x0 = wx16 - 8.0f, y0 = wy16 - 8.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { PX[k] = dep_snap(cx[k], wx16, x0); PY[k] = dep_snap(cy[k], wy16, y0); }
}

// the hexagon clipped against the view volume (lines that cross the view's edge: the rare case, runtime-indexed arrays)
TH_D void dep_clip_hexagon(const DepositParams &p, DepositLine &L, const float (&hx6)[6], const float (&hy6)[6])
{
    const float wx16 = 8.0f * (float)p.fw, wy16 = 8.0f * (float)p.fh;
    const float x0 = wx16 - 8.0f, y0 = wy16 - 8.0f;
    float cx[12], cy[12], tx[12], ty[12];
    for (int k = 0; k < 6; ++k) { cx[k] = hx6[k]; cy[k] = hy6[k]; }
    int n = 6;
    // Sutherland-Hodgman against left, right, top, bottom; intersection (dj*Vi - di*Vj) * (1/(dj - di)), inside vertex first
    for (int plane = 0; plane < 4 && n >= 3; ++plane) {
        int t = 0;
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
                tx[t] = cx[k]; ty[t] = cy[k]; ++t;
                if (dj < 0.0f) {
                    const float D = 1.0f / (dj - di);
                    tx[t] = (dj * cx[k] - di * cx[j]) * D; ty[t] = (dj * cy[k] - di * cy[j]) * D; ++t;
                }
            } else if (dj > 0.0f) {
                const float D = 1.0f / (di - dj);
                tx[t] = (di * cx[j] - dj * cx[k]) * D; ty[t] = (di * cy[j] - dj * cy[k]) * D; ++t;
            }
        }
        n = t;
        for (int k = 0; k < n; ++k) { cx[k] = tx[k]; cy[k] = ty[k]; }
    }
    if (n < 3) { L.draws = false; L.n = 0; return; }
    L.n = n;
    for (int k = 0; k < n; ++k) { L.PX[k] = dep_snap(cx[k], wx16, x0); L.PY[k] = dep_snap(cy[k], wy16, y0); }
}

// scan conversion: calls emit(x, y) for every covered texel.  Edges going up in y set `left`, edges going down set
// `right` (a later edge overwrites an earlier one on the same row, as in the captured rasteriser); texels
// left <= x < right.  Rows are walked in windows so that arbitrarily long lines need no large arrays.
template <int N, typename Emit>
TH_D void dep_raster_poly(const DepositParams &p, const int *PX, const int *PY, int count, Emit emit)
{
    // N > 0: a polygon of exactly N vertices held in registers (loops unrolled, static indices); N == 0: `count` vertices
    const int nv = N > 0 ? N : count;
    int ymin = PY[0], ymax = PY[0];
#pragma unroll
    for (int k = 1; k < (N > 0 ? N : 12); ++k) if (k < nv) { ymin = PY[k] < ymin ? PY[k] : ymin; ymax = PY[k] > ymax ? PY[k] : ymax; }
    int r0 = (ymin + 15) >> 4, r1 = (ymax + 15) >> 4;
    if (r0 < 0) r0 = 0;
    if (r1 > p.fh) r1 = p.fh;
    constexpr int kWindow = N == 6 ? 4 : 8;
    for (int base = r0; base < r1; base += kWindow) {
        const int top = base + kWindow < r1 ? base + kWindow : r1;
        int left[kWindow], right[kWindow];
#pragma unroll
        for (int k = 0; k < kWindow; ++k) { left[k] = p.fw; right[k] = 0; }
#pragma nounroll
        for (int k = 0; k < nv; ++k) {
            const int kn = k + 1 == nv ? 0 : k + 1;
            int Xa, Ya, Xb, Yb;
            if constexpr (N == 6) {
                // the six vertices stay in registers: the loop is not unrolled (six copies of its body cost 190 VGPRs),
                // a vertex is picked with a chain of selects instead of an index
                auto pick = [](const int *v, int i) { int r = v[0]; r = i == 1 ? v[1] : r; r = i == 2 ? v[2] : r; r = i == 3 ? v[3] : r;
                                                      r = i == 4 ? v[4] : r; r = i == 5 ? v[5] : r; return r; };
                Xa = pick(PX, k); Ya = pick(PY, k); Xb = pick(PX, kn); Yb = pick(PY, kn);
            } else { Xa = PX[k]; Ya = PY[k]; Xb = PX[kn]; Yb = PY[kn]; }
            if (Ya == Yb) continue;
            const bool swap = Yb < Ya;
            const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
            int e0 = (Y1 + 15) >> 4, e1 = (Y2 + 15) >> 4;
            if (e0 < base) e0 = base;
            if (e1 > top) e1 = top;
            const long long DX = X2 - X1, DY = Y2 - Y1;
            // short edges inside a 32768-texel-wide view (every practical case): the same quotient in 32-bit arithmetic
            const bool small = DY < 1024 && DX > -4096 && DX < 4096 && X1 > -(1 << 19) && X1 < (1 << 19);
            const float rden = __builtin_amdgcn_rcpf((float)(16 * (int)DY));
            for (int y = e0; y < e1; ++y) {
                long long x;
                if (small) {
                    // |num| < 2^30, 16 <= den < 2^14, |quotient| < 2^16: the float estimate of the floor is within one of
                    // it (relative error of the conversion, v_rcp_f32 and the product < 2^-21), the remainder decides
                    const int num = (int)DX * ((y << 4) - Y1) + X1 * (int)DY, den = 16 * (int)DY;
                    int q = (int)__builtin_floorf((float)num * rden);
                    int r = num - q * den;
                    if (r < 0) { --q; r += den; }
                    if (r >= den) { ++q; r -= den; }
                    x = r > 0 ? q + 1 : q;                     // ceil
                } else x = dep_ceil_div(DX * (((long long)y << 4) - Y1) + (long long)X1 * DY, 16 * DY);
                if (x < 0) x = 0;
                if (x > p.fw) x = p.fw;
#pragma unroll
                for (int w = 0; w < kWindow; ++w)          // static indexing keeps the spans in registers
                    if (w == y - base) { if (swap) right[w] = (int)x; else left[w] = (int)x; }
            }
        }
#pragma unroll
        for (int w = 0; w < kWindow; ++w)
            if (base + w < top)
                for (int x = left[w]; x < right[w]; ++x) emit(x, base + w);
    }
}

// The common case - a hexagon inside the view whose six edges all take the 32-bit division above (whole polygon within
// 4096 x 1024 sixteenths) - row by row with its six vertices in registers and nothing indexed at run time: per row
// every edge that crosses it sets its end of the span, in vertex order as above.  Same quotients, same spans.
TH_D bool dep_hexagon_is_small(const int (&PX)[6], const int (&PY)[6], int &ymin, int &ymax)
{
    int xmin = PX[0], xmax = PX[0];
    ymin = PY[0]; ymax = PY[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        xmin = PX[k] < xmin ? PX[k] : xmin; xmax = PX[k] > xmax ? PX[k] : xmax;
        ymin = PY[k] < ymin ? PY[k] : ymin; ymax = PY[k] > ymax ? PY[k] : ymax;
    }
    return xmin > -(1 << 19) && xmax < (1 << 19) && xmax - xmin < 4096 && ymax - ymin < 1024;
}

template <typename Emit>
TH_D void dep_raster_small_hexagon(const DepositParams &p, const int (&PX)[6], const int (&PY)[6], int ymin, int ymax, Emit emit)
{
    int r0 = (ymin + 15) >> 4, r1 = (ymax + 15) >> 4;
    if (r0 < 0) r0 = 0;
    if (r1 > p.fh) r1 = p.fh;
    for (int y = r0; y < r1; ++y) {
        int left = p.fw, right = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int kn = k == 5 ? 0 : k + 1;
            const int Xa = PX[k], Ya = PY[k], Xb = PX[kn], Yb = PY[kn];
            const bool swap = Yb < Ya;
            const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
            const bool crosses = y >= ((Y1 + 15) >> 4) && y < ((Y2 + 15) >> 4);        // never for Ya == Yb
            const int DX = X2 - X1, DY = Y2 - Y1, den = DY > 0 ? 16 * DY : 16;
            const int num = DX * ((y << 4) - Y1) + X1 * DY;
            int q = (int)__builtin_floorf((float)num * __builtin_amdgcn_rcpf((float)den));
            int r = num - q * den;
            if (r < 0) { --q; r += den; }
            if (r >= den) { ++q; r -= den; }
            int x = r > 0 ? q + 1 : q;
            x = x < 0 ? 0 : (x > p.fw ? p.fw : x);
            if (crosses) { if (swap) right = x; else left = x; }
        }
        for (int x = left; x < right; ++x) emit(x, y);
    }
}

// a line, whichever way it has to go: straight from its hexagon, or clipped first
template <typename Emit>
TH_D void dep_raster_line(const DepositParams &p, DepositLine &L, Emit emit)
{
    float cx[6], cy[6];
    const int where = dep_hexagon(p, L, cx, cy);
    if (where == kHexInside) {
        int PX[6], PY[6];
        dep_snap_hexagon(p, cx, cy, PX, PY);
        dep_raster_poly<6>(p, PX, PY, 6, emit);
    } else if (where == kHexClip) {
        dep_clip_hexagon(p, L, cx, cy);
        if (L.draws) dep_raster_poly<0>(p, L.PX, L.PY, L.n, emit);
    }
}

// the varying of line L at texel (x, y): linear along the snapped endpoints, extrapolated, unclamped.  dep_param: the
// interpolation parameter (false: both endpoints snap to the same point, the first vertex's value is taken)
TH_D bool dep_param(const DepositLine &L, int x, int y, float &t)
{
    if (L.short32) {        // the same integers in 32 bits (a fragment lies within a texel of its line): the same floats
        const int ex = L.sx[1] - L.sx[0], ey = L.sy[1] - L.sy[0], den = ex * ex + ey * ey;
        if (den == 0) return false;
        const int num = ((x << 4) - L.sx[0]) * ex + ((y << 4) - L.sy[0]) * ey;
        t = (float)num / (float)den;
    } else {
        const long long ex = L.sx[1] - L.sx[0], ey = L.sy[1] - L.sy[0], den = ex * ex + ey * ey;
        if (den == 0) return false;
        const long long num = ((long long)(x << 4) - L.sx[0]) * ex + ((long long)(y << 4) - L.sy[0]) * ey;
        t = (float)num / (float)den;
    }
    return true;
}
TH_D float4 dep_mix(const float (&a)[4], const float (&b)[4], bool along, float t)
{
    if (!along) return make_float4(a[0], a[1], a[2], a[3]);
    return make_float4(a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), a[2] + t * (b[2] - a[2]), a[3] + t * (b[3] - a[3]));
}

constexpr uint32_t kNeedsSlow = 0xffffffffu;       // count[] marker between deposit_raster_kernel and its _slow pass

constexpr uint32_t kRecordTexels = 8;             // texels a line's record holds (two uint4 per line): x | y << 16
struct LineRecord { uint32_t n, r[kRecordTexels]; };
TH_D void rec_add(LineRecord &q, int x, int y)
{
    const uint32_t xy = (uint32_t)x | ((uint32_t)y << 16);
#pragma unroll
    for (uint32_t k = 0; k < kRecordTexels; ++k) {      // static indices and selects: the record stays in registers
        uint32_t v = q.n == k ? xy : q.r[k];
        asm volatile("" : "+v"(v));                     // (hipcc would turn the chain into an indexed store to scratch)
        q.r[k] = v;
    }
    ++q.n;
}
TH_D void rec_store(const DepositParams &p, uint32_t t, const LineRecord &q)
{
    p.record[2u * t] = make_uint4(q.r[0], q.r[1], q.r[2], q.r[3]);
    if (q.n > 4u) p.record[2u * t + 1u] = make_uint4(q.r[4], q.r[5], q.r[6], q.r[7]);
}

// Lines the fast kernels leave to a slower one (hexagons that need clipping or 64-bit edges; lines of more fragments than
// a record holds) are appended to lists: kDepLists segments with a counter each (one hot counter would serialise the
// appends of the whole chip; the segment of a line is picked from its 256-line group, so a segment can never receive
// more than its share of ALL lines: no overflow check), one atomic per wave that has any.  The slow kernels then run on
// full waves instead of sifting every line for the few.
constexpr uint32_t kDepLists = 64, kDepListStride = 64;          // counters 256 B apart

TH_D void dep_list_append(const DepositParams &p, uint32_t which, uint32_t group, bool mine, uint32_t t)
{
    const unsigned long long m = __ballot(mine);
    if (m == 0ull) return;
    // (the lanes of a wave share their segment: a fast kernel's wave lies inside one group, a slow kernel's workgroup
    // works through one segment of its input list and appends to the same segment of the other)
    const uint32_t seg = __builtin_amdgcn_readfirstlane(group) & (kDepLists - 1u);
    const uint32_t lane = __lane_id(), leader = (uint32_t)__builtin_ctzll(m);
    uint32_t first = 0;
    if (lane == leader) first = atomicAdd(&p.list_n[(which * kDepLists + seg) * kDepListStride], (uint32_t)__builtin_popcountll(m));
    first = __shfl(first, leader);
    if (mine) p.lists[((size_t)which * kDepLists + seg) * p.list_cap + first + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = t;
}

// the workgroups of a slow kernel: (segment, part) = (blockIdx % kDepLists, blockIdx / kDepLists)
template <typename Work>
TH_D void dep_list_work(const DepositParams &p, uint32_t which, Work work)
{
    const uint32_t seg = blockIdx.x & (kDepLists - 1u), part = blockIdx.x / kDepLists, parts = gridDim.x / kDepLists;
    const uint32_t n = p.list_n[(which * kDepLists + seg) * kDepListStride];
    const uint32_t *list = p.lists + ((size_t)which * kDepLists + seg) * p.list_cap;
    for (uint32_t e0 = part * 256u; e0 < n; e0 += parts * 256u) {         // whole waves stay together (the appends ballot)
        const uint32_t e = e0 + threadIdx.x;
        work(e < n, e < n ? list[e] : 0u, seg);
    }
}
enum { kListSlow = 0, kListLong = 1 };

// pass 1: rasterise every line once: fragment count and (count <= kRecordTexels) the covered texels.  Workgroups walk
// the particle texture row-major in pieces of 256 columns of a row (coalesced state reads, no division per line); the
// line's place in the fragment array is its position in the vertex stream (column-major): the scan below is over
// that order.  Only the common case is done here, with everything in registers; the rest goes to the slow list.
__global__ __launch_bounds__(256) void deposit_raster_kernel(const DepositParams p)
{
    const uint32_t pieces = (p.W + 255u) >> 8, groups = pieces * p.rows;
    for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const uint32_t row = g / pieces, col = ((g - row * pieces) << 8) + threadIdx.x;
        const bool have = col < p.W;
        const uint32_t t = row * p.W + col;
        LineRecord r{};
        bool slow = false;
        if (have) {
            DepositLine L;
            dep_setup(p, col, p.row0 + row, L);
            if (L.draws) {
                float cx[6], cy[6];
                const int where = dep_hexagon(p, L, cx, cy);
                if (where == kHexInside) {
                    int PX[6], PY[6], ymin, ymax;
                    dep_snap_hexagon(p, cx, cy, PX, PY);
                    if (dep_hexagon_is_small(PX, PY, ymin, ymax))
                        dep_raster_small_hexagon(p, PX, PY, ymin, ymax, [&](int x, int y) { rec_add(r, x, y); });
                    else slow = true;
                } else if (where == kHexClip) slow = true;
            }
            p.count[t] = slow ? kNeedsSlow : r.n;
            if (r.n) rec_store(p, t, r);
        }
        dep_list_append(p, kListSlow, g, slow, t);
        dep_list_append(p, kListLong, g, r.n > kRecordTexels, t);
    }
}

__global__ __launch_bounds__(256) void deposit_raster_slow_kernel(const DepositParams p)
{
    dep_list_work(p, kListSlow, [&](bool have, uint32_t t, uint32_t seg) {
        LineRecord r{};
        if (have) {
            const uint32_t row = t / p.W, col = t - row * p.W;
            DepositLine L;
            dep_setup(p, col, p.row0 + row, L);
            dep_raster_line(p, L, [&](int x, int y) { rec_add(r, x, y); });
            p.count[t] = r.n;
            if (r.n) rec_store(p, t, r);
        }
        dep_list_append(p, kListLong, seg, r.n > kRecordTexels, t);
    });
}

// one fragment into slot `at` of the stream-ordered fragment array
TH_D void dep_put(const DepositParams &p, const DepositLine &L, uint32_t id, uint32_t at, int x, int y)
{
    const uint32_t texel = (uint32_t)y * (uint32_t)p.fw + (uint32_t)x;
    if (p.keys64) {         // sharded form: owner of the texel | texel | stream index of the line
        const uint32_t owner = p.owners > 1u ? (texel / p.owner_chunk < p.owners - 1u ? texel / p.owner_chunk : p.owners - 1u) : 0u;
        p.keys64[at] = ((unsigned long long)owner << kOwnerShift) | ((unsigned long long)texel << 32) | id;
    }
    else if (p.keys) p.keys[at] = texel;       // (the sort numbers the fragments itself; no keys: a pass that reuses the sorted order)
    float t = 0.0f;
    const bool along = dep_param(L, x, y, t);
    if (p.mode == 2) {      // both passes' varyings side by side: one gather brings both into the sorted order
        p.colors[2u * at] = dep_mix(L.a.c, L.b.c, along, t);
        p.colors[2u * at + 1u] = dep_mix(L.a.c2, L.b.c2, along, t);
    } else p.colors[at] = dep_mix(L.a.c, L.b.c, along, t);
}

// pass 3: the fragments of the lines of up to kRecordTexels fragments, from their records, into the lines' slots.  Threads
// walk patches of kPatchCols x 64 lines, one column per wave: a wave's lines are 64 consecutive positions of the
// stream, so its fragments form ONE contiguous run of the fragment array (walking row-major, every lane writes
// somewhere else: 1.2 ms for this pass at C3 against 0.3); a patch reads whole 128-byte lines of the state rows.
constexpr uint32_t kPatchCols = 8, kPatchRows = 64;

__global__ __launch_bounds__(kPatchCols * 64) void deposit_emit_kernel(const DepositParams p)
{
    const uint32_t patches_x = (p.W + kPatchCols - 1) / kPatchCols, patches_y = (p.rows + kPatchRows - 1) / kPatchRows;
    const uint32_t patches = patches_x * patches_y;
    for (uint32_t patch = blockIdx.x; patch < patches; patch += gridDim.x) {
        // patches in column-major order too: consecutive workgroups continue each other's runs
        const uint32_t px = patch / patches_y, py = patch - px * patches_y;
        const uint32_t col = px * kPatchCols + (threadIdx.x >> 6), row = py * kPatchRows + (threadIdx.x & 63u);
        if (col >= p.W || row >= p.rows) continue;
        const uint32_t t = row * p.W + col;
        const uint32_t n = p.count[t];
        if (n == 0 || n > kRecordTexels) continue;
        const uint32_t id = col * p.H + p.row0 + row;
        const uint32_t at = p.offset[t];
        DepositLine L;
        dep_setup(p, col, p.row0 + row, L);           // vertices and snapped endpoints
        const uint4 ra = p.record[2u * t];
        uint4 rb = make_uint4(0u, 0u, 0u, 0u);
        if (n > 4u) rb = p.record[2u * t + 1u];
        const uint32_t xy[kRecordTexels] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (uint32_t k = 0; k < kRecordTexels; ++k)
            if (k < n) dep_put(p, L, id, at + k, (int)(xy[k] & 0xffffu), (int)(xy[k] >> 16));
    }
}

// ... and the lines of more fragments than a record holds, rasterised again
__global__ __launch_bounds__(256) void deposit_emit_long_kernel(const DepositParams p)
{
    dep_list_work(p, kListLong, [&](bool have, uint32_t t, uint32_t) {
        if (!have) return;
        const uint32_t row = t / p.W, col = t - row * p.W;
        const uint32_t id = col * p.H + p.row0 + row;
        uint32_t at = p.offset[t];
        DepositLine L;
        dep_setup(p, col, p.row0 + row, L);
        dep_raster_line(p, L, [&](int x, int y) { dep_put(p, L, id, at, x, y); ++at; });
    });
}

TH_D void dep_blend_rgba(float4 &d, float4 c) { const float sa = c.w, da = 1.0f - sa; d.x = c.x * sa + d.x * da; d.y = c.y * sa + d.y * da; d.z = c.z * sa + d.z * da; d.w = c.w * sa + d.w * da; }

// the view pass's blend: the RGBA8 drawing buffer - the fragment colour is clamped to [0, 1], blended with the stored
// colour c/255 and stored as round(255 x), fragment after fragment (what the captured GL does)
TH_D void dep_blend_rgba8(uchar4 &q, float4 c)
{
    c.x = __builtin_fminf(__builtin_fmaxf(c.x, 0.0f), 1.0f); c.y = __builtin_fminf(__builtin_fmaxf(c.y, 0.0f), 1.0f);
    c.z = __builtin_fminf(__builtin_fmaxf(c.z, 0.0f), 1.0f); c.w = __builtin_fminf(__builtin_fmaxf(c.w, 0.0f), 1.0f);
    const float sa = c.w, da = 1.0f - sa, k = 1.0f / 255.0f;
    auto mix8 = [&](float src, unsigned char dst) {
        const float o = src * sa + ((float)dst * k) * da;
        return (unsigned char)(__builtin_fminf(__builtin_fmaxf(o, 0.0f), 1.0f) * 255.0f + 0.5f);
    };
    q = make_uchar4(mix8(c.x, q.x), mix8(c.y, q.y), mix8(c.z, q.z), mix8(c.w, q.w));
}

// pass 5: fragments sorted by texel (stable: stream order inside a texel).  The lane at the head of a texel's run blends
// its first kShortRun fragments itself, four read ahead of the dependent blends - nearly every run ends there.  What
// is left of a longer run (the wake makes particles converge: thousands of fragments in one texel are normal after
// a few dozen frames) is then blended by the whole wave: 64 fragments per coalesced load, the next 64 in flight,
// every lane doing the same sequential arithmetic on values broadcast with v_readlane - the same operations in the
// same order, at 64 fragments per memory round trip instead of 4.
constexpr int kShortRun = 16;

// a fragment's side of the blend (everything that does not depend on the destination), and the destination's
struct BlendSource { float x, y, z, w, da; };
TH_D float lane_float(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
struct FlowTarget {                     // dep_blend in two halves
    using Texel = float4;
    TH_D static BlendSource source(float4 c) { const float sa = c.w; return BlendSource{c.x * sa, c.y * sa, c.z * sa, c.w * sa, 1.0f - sa}; }
    TH_D static void apply(float4 &d, const BlendSource &s) { d.x = s.x + d.x * s.da; d.y = s.y + d.y * s.da; d.z = s.z + d.z * s.da; d.w = s.w + d.w * s.da; }
    TH_D static float4 *plane(const DepositParams &p) { return p.flow; }
    TH_D static float4 from_lane(float4 d, int lane) { return make_float4(lane_float(d.x, lane), lane_float(d.y, lane), lane_float(d.z, lane), lane_float(d.w, lane)); }
};
struct ViewTarget {                     // dep_blend_rgba8 in two halves
    using Texel = uchar4;
    TH_D static BlendSource source(float4 c)
    {
        c.x = __builtin_fminf(__builtin_fmaxf(c.x, 0.0f), 1.0f); c.y = __builtin_fminf(__builtin_fmaxf(c.y, 0.0f), 1.0f);
        c.z = __builtin_fminf(__builtin_fmaxf(c.z, 0.0f), 1.0f); c.w = __builtin_fminf(__builtin_fmaxf(c.w, 0.0f), 1.0f);
        const float sa = c.w;
        return BlendSource{c.x * sa, c.y * sa, c.z * sa, c.w * sa, 1.0f - sa};
    }
    TH_D static void apply(uchar4 &q, const BlendSource &s)
    {
        const float k = 1.0f / 255.0f;
        auto mix8 = [&](float src, unsigned char dst) {
            const float o = src + ((float)dst * k) * s.da;
            return (unsigned char)(__builtin_fminf(__builtin_fmaxf(o, 0.0f), 1.0f) * 255.0f + 0.5f);
        };
        q = make_uchar4(mix8(s.x, q.x), mix8(s.y, q.y), mix8(s.z, q.z), mix8(s.w, q.w));
    }
    TH_D static uchar4 *plane(const DepositParams &p) { return p.view; }
    TH_D static uchar4 from_lane(uchar4 d, int lane) { return __builtin_bit_cast(uchar4, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), lane)); }
};

// what the fragments are sorted by.  TexelKeys: the local deposit (the key is the texel, the order inside a run is the
// stream's).  BandKeys: the sharded deposit - an owner's fragments sorted stably by texel; inside a texel's run the
// fragments of every source band still follow each other in that band's stream order, band after band: where the
// stream index falls, the next band begins, and the run has to be merged by stream index.
struct TexelKeys {
    const uint32_t *k;
    static constexpr bool kBands = false;
    TH_D uint32_t texel(uint32_t at) const { return k[at]; }
    TH_D void both(uint32_t at, uint32_t &texel, uint32_t &id) const { texel = k[at]; id = 0u; }
};
struct BandKeys {
    const unsigned long long *k;
    static constexpr bool kBands = true;
    TH_D uint32_t texel(uint32_t at) const { return (uint32_t)(k[at] >> 32) & kTexelMask; }
    TH_D void both(uint32_t at, uint32_t &texel, uint32_t &id) const { const unsigned long long v = k[at]; texel = (uint32_t)(v >> 32) & kTexelMask; id = (uint32_t)v; }
};

// a run made of several bands (BandKeys), by its head lane: short runs by repeated selection of the next stream index,
// long ones by a merge of up to kMaxBands band cursors
constexpr int kMaxBands = 32;
template <typename Target>
TH_D void blend_banded_run(typename Target::Texel *plane, const BandKeys &keys, const float4 *colors, uint32_t i, uint32_t total,
                           uint32_t texel, uint32_t *too_many)
{
    uint32_t end = i + 1u, bands = 1u;
    while (end < total && keys.texel(end) == texel) { bands += (uint32_t)keys.k[end] < (uint32_t)keys.k[end - 1u] ? 1u : 0u; ++end; }
    typename Target::Texel d = plane[texel];
    if (end - i <= 16u) {
        unsigned long long after = 0ull;            // (stream index + 1 of the fragment blended last)
        for (uint32_t n = i; n < end; ++n) {
            uint32_t best = i;
            unsigned long long best_id = ~0ull;
            for (uint32_t j = i; j < end; ++j) {
                const unsigned long long id = (uint32_t)keys.k[j];
                if (id >= after && id < best_id) { best = j; best_id = id; }
            }
            Target::apply(d, Target::source(colors[best]));
            after = best_id + 1ull;
        }
    } else if (bands <= (uint32_t)kMaxBands) {
        uint32_t pos[kMaxBands], lim[kMaxBands];
        uint32_t nb = 0;
        pos[0] = i;
        for (uint32_t j = i + 1u; j < end; ++j)
            if ((uint32_t)keys.k[j] < (uint32_t)keys.k[j - 1u]) { lim[nb] = j; pos[++nb] = j; }
        lim[nb++] = end;
        unsigned long long head[kMaxBands];          // stream index at every band's cursor (one past the largest: band exhausted)
        for (uint32_t b = 0; b < nb; ++b) head[b] = (uint32_t)keys.k[pos[b]];
        for (uint32_t n = i; n < end; ++n) {
            uint32_t best = 0;
            for (uint32_t b = 1; b < nb; ++b) if (head[b] < head[best]) best = b;
            Target::apply(d, Target::source(colors[pos[best]]));
            ++pos[best];                              // only the band that moved reads its next key
            head[best] = pos[best] < lim[best] ? (unsigned long long)(uint32_t)keys.k[pos[best]] : 0x100000000ull;
        }
    } else *too_many = 1u;
    plane[texel] = d;
}

// colors[at * stride]: stride 2 when the two passes' varyings lie side by side (mode 2; `colors` then points at the pass's own)
template <typename Target, typename Keys>
__global__ __launch_bounds__(256) void deposit_blend_kernel(typename Target::Texel *plane, const Keys keys, const float4 *colors,
                                                            uint32_t stride, uint32_t total, uint32_t *too_many)
{
    using Texel = typename Target::Texel;
    const uint32_t lane = __lane_id();
    __shared__ BlendSource staged[256];            // (per wave: its 64 lanes' source halves of the batch being blended)
    for (uint32_t base = blockIdx.x * 256u; base < total; base += gridDim.x * 256u) {       // (whole waves stay together)
        const uint32_t i = base + threadIdx.x;
        uint32_t texel = 0, j = i, last = 0;
        bool head = false, unfinished = false, banded = false;
        if (i < total) {
            keys.both(i, texel, last);
            head = i == 0 || keys.texel(i - 1) != texel;
        }
        Texel d{};
        if (head) {
            d = plane[texel];
            bool done = false;
            for (int round = 0; round < kShortRun / 4 && !done; ++round) {
                float4 c[4];
                uint32_t k[4], id[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t at = j + (uint32_t)q < total ? j + (uint32_t)q : total - 1u;
                    c[q] = colors[(size_t)at * stride];
                    keys.both(at, k[q], id[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!done && j + (uint32_t)q < total && k[q] == texel) {
                        if (Keys::kBands && id[q] < last) { banded = true; done = true; }
                        else { Target::apply(d, Target::source(c[q])); last = id[q]; }
                    } else done = true;
                }
                if (!done) j += 4u;
            }
            unfinished = !done;
            if (done && !banded) plane[texel] = d;
        }
        // the long runs of this wave, one after the other, by all its lanes
        unsigned long long longs = __ballot(unfinished);
        while (longs != 0ull) {
            const int owner = __builtin_ctzll(longs);
            longs &= longs - 1ull;
            const uint32_t run_texel = (uint32_t)__builtin_amdgcn_readlane((int)texel, owner);
            uint32_t at0 = (uint32_t)__builtin_amdgcn_readlane((int)j, owner);
            uint32_t run_last = (uint32_t)__builtin_amdgcn_readlane((int)last, owner);
            Texel rd = Target::from_lane(d, owner);
            auto fetch = [&](uint32_t first, float4 &c, bool &same, uint32_t &id) {
                const uint32_t at = first + lane;
                const bool in = at < total;
                c = colors[(size_t)(in ? at : total - 1u) * stride];
                uint32_t t;
                keys.both(in ? at : total - 1u, t, id);
                same = in && t == run_texel;
            };
            float4 c, cn;
            bool same, samen, falls = false;
            uint32_t id = 0, idn = 0;
            fetch(at0, c, same, id);
            while (true) {
                fetch(at0 + 64u, cn, samen, idn);              // in flight while this batch is blended
                const unsigned long long in_run = __ballot(same);
                const int n = in_run == ~0ull ? 64 : __builtin_ctzll(~in_run);
                if constexpr (Keys::kBands) {
                    const uint32_t up = (uint32_t)__shfl_up((int)id, 1);        // (by every lane: a masked-off source lane reads as 0)
                    const uint32_t before = lane ? up : run_last;
                    if (__ballot((int)lane < n && id < before) != 0ull) { falls = true; break; }
                    if (n) run_last = (uint32_t)__builtin_amdgcn_readlane((int)id, n - 1);
                }
                // every lane its own fragment's half; then in order, read back from the wave's LDS slots (a broadcast read per
                // fragment, issued eight ahead: the chain left is the blend's own multiply and add - v_readlane goes through
                // an SGPR and its wait states, three times as long per fragment on the run that sets a crowded frame's time)
                staged[threadIdx.x] = Target::source(c);
                const BlendSource *mine = &staged[threadIdx.x & ~63u];
                int q = 0;
                for (; q + 8 <= n; q += 8) {
                    BlendSource s8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) s8[k] = mine[q + k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) Target::apply(rd, s8[k]);
                }
                for (; q < n; ++q) Target::apply(rd, mine[q]);
                if (n < 64) break;
                c = cn; same = samen; id = idn; at0 += 64u;
            }
            if (lane == (uint32_t)owner) { if (falls) banded = true; else plane[run_texel] = rd; }
        }
        if constexpr (Keys::kBands) if (banded) blend_banded_run<Target>(plane, keys, colors, i, total, texel, too_many);
    }
}

// Tendrils.drawFill (src/index.js:350-356): one full-screen quad of `color`, blended like everything else
__global__ __launch_bounds__(256) void view_fill_kernel(uchar4 *view, size_t n, float4 color)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uchar4 d = view[i];
        dep_blend_rgba8(d, color);
        view[i] = d;
    }
}

__global__ __launch_bounds__(256) void deposit_gather_colors_kernel(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[index[i]];
}
// pairs of varyings (mode 2): 32 bytes from one random place instead of 16 from two
__global__ __launch_bounds__(256) void deposit_gather_pairs_kernel(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const size_t from = 2u * (size_t)index[i];
        const float4 a = src[from], b = src[from + 1u];
        dst[2u * (size_t)i] = a; dst[2u * (size_t)i + 1u] = b;
    }
}

// ---- exclusive scan of the per-line fragment counts in stream order ---------------------------------------------
// The counts lie row-major (count[row*W + col], as the threads that produced them walk), the stream runs column-major
// (col*rows + row): offset(row, col) = (fragments of the columns before col) + (fragments of rows before `row` in col).
constexpr uint32_t kColRows = 64;            // rows per partial sum

// part[rb*W + col] = sum of count[r][col] over the rows of row block rb
__global__ __launch_bounds__(256) void colscan_partial_kernel(const uint32_t *count, uint32_t W, uint32_t rows, uint32_t *part)
{
    const uint32_t col = blockIdx.x * 256u + threadIdx.x, rb = blockIdx.y;
    if (col >= W) return;
    const uint32_t r0 = rb * kColRows, r1 = r0 + kColRows < rows ? r0 + kColRows : rows;
    uint32_t s = 0;
    if (r1 - r0 == kColRows) {              // a full block: all its loads in flight together
        uint32_t v[kColRows];
#pragma unroll
        for (uint32_t k = 0; k < kColRows; ++k) v[k] = count[(size_t)(r0 + k) * W + col];
#pragma unroll
        for (uint32_t k = 0; k < kColRows; ++k) s += v[k];
    } else for (uint32_t r = r0; r < r1; ++r) s += count[(size_t)r * W + col];
    part[(size_t)rb * W + col] = s;
}

// part -> exclusive prefix over the row blocks of every column (one thread per column), the column's total to coltotal
// (saturated at 0xffffffff: 64-bit running sums, a total beyond 2^32 must be seen)
__global__ __launch_bounds__(256) void colscan_prefix_kernel(uint32_t *part, uint32_t W, uint32_t nrb, uint32_t *coltotal)
{
    const uint32_t col = blockIdx.x * 256u + threadIdx.x;
    if (col >= W) return;
    unsigned long long s = 0;
    uint32_t rb = 0;
    for (; rb + 16u <= nrb; rb += 16u) {            // 16 row blocks at a time, their loads in flight together
        uint32_t v[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) v[k] = part[(size_t)(rb + k) * W + col];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) {
            part[(size_t)(rb + k) * W + col] = (uint32_t)(s > 0xffffffffull ? 0xffffffffull : s);
            s += v[k];
        }
    }
    for (; rb < nrb; ++rb) {
        const uint32_t v = part[(size_t)rb * W + col];
        part[(size_t)rb * W + col] = (uint32_t)(s > 0xffffffffull ? 0xffffffffull : s);
        s += v;
    }
    coltotal[col] = (uint32_t)(s > 0xffffffffull ? 0xffffffffull : s);
}

// one workgroup: coltotal -> colbase[col] = fragments of the columns before col (in place); the total to *total
__global__ __launch_bounds__(1024) void colscan_bases_kernel(uint32_t W, uint32_t *colbase, uint32_t *total)
{
    __shared__ unsigned long long sh[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < W; c0 += 1024u) {
        const uint32_t col = c0 + threadIdx.x;
        const unsigned long long s = col < W ? colbase[col] : 0ull;
        sh[threadIdx.x] = s;
        __syncthreads();
        for (uint32_t o = 1; o < 1024u; o <<= 1) {
            unsigned long long add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0ull;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        const unsigned long long excl = carry + sh[threadIdx.x] - s;
        if (col < W) colbase[col] = (uint32_t)(excl > 0xffffffffull ? 0xffffffffull : excl);
        __syncthreads();
        if (threadIdx.x == 0) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = (uint32_t)(carry > 0xffffffffull ? 0xffffffffull : carry);
}

__global__ __launch_bounds__(256) void colscan_offsets_kernel(const uint32_t *count, uint32_t W, uint32_t rows, const uint32_t *part,
                                                              const uint32_t *colbase, uint32_t *offset)
{
    const uint32_t col = blockIdx.x * 256u + threadIdx.x, rb = blockIdx.y;
    if (col >= W) return;
    const uint32_t r0 = rb * kColRows, r1 = r0 + kColRows < rows ? r0 + kColRows : rows;
    uint32_t run = colbase[col] + part[(size_t)rb * W + col];
    if (r1 - r0 == kColRows) {
        uint32_t v[kColRows];
#pragma unroll
        for (uint32_t k = 0; k < kColRows; ++k) v[k] = count[(size_t)(r0 + k) * W + col];
#pragma unroll
        for (uint32_t k = 0; k < kColRows; ++k) { offset[(size_t)(r0 + k) * W + col] = run; run += v[k]; }
    } else for (uint32_t r = r0; r < r1; ++r) { offset[(size_t)r * W + col] = run; run += count[(size_t)r * W + col]; }
}

// ---- trail export: the line list of draw() (12 floats per line, stream order) ---------------------------------
// pass 1 marks the lines that exist (two live vertices, non-zero length), the scan places them, pass 2 writes them
template <bool WRITE>
__global__ __launch_bounds__(256) void export_lines_kernel(const DepositParams p, float *out)
{
    const uint32_t lines = p.W * p.rows;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < lines; t += gridDim.x * 256u) {
        const uint32_t row = t / p.W, col = t - row * p.W;
        const uint32_t i = col, m = p.row0 + row;
        const DepositVertex a = dep_fetch(p, i, 2u * m), b = dep_fetch(p, i, 2u * m + 1u);
        const bool exists = a.live && b.live && !(a.px == b.px && a.py == b.py);
        if constexpr (WRITE) {
            if (!exists) continue;
            float *o = out + 12ull * p.offset[t];
            o[0] = a.px; o[1] = a.py; o[2] = b.px; o[3] = b.py;
            for (int k = 0; k < 4; ++k) { o[4 + k] = a.c[k]; o[8 + k] = b.c[k]; }
        } else {
            p.count[t] = exists ? 1u : 0u;
        }
    }
}

// ---- GeometrySpawner's draw (src/spawn/geometry/index.js:97-115): triangles into the spawner's float buffer ---------
// gl_Position = (position*viewSize, 0, 1) (src/geom/vert/index.vert:3-5), constant colour (src/geom/frag/index.frag),
// blend SRC_ALPHA / ONE_MINUS_SRC_ALPHA in primitive order; same rasteriser conventions as the lines above, either
// winding drawn.  Kernel 1 clips, snaps and orients every triangle; kernel 2: one thread per texel walks the
// triangles in order.
__global__ void triangle_setup_kernel(const float *positions, int ntri, float view_x, float view_y, int w, int h, TrianglePoly *polys)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntri) return;
    const float wx16 = 8.0f * (float)w, wy16 = 8.0f * (float)h;
    const float x0 = wx16 - 8.0f, y0 = wy16 - 8.0f;
    float cx[12], cy[12], tx[12], ty[12];
    int n = 3;
    for (int k = 0; k < 3; ++k) {
        cx[k] = positions[6 * t + 2 * k] * view_x;
        cy[k] = positions[6 * t + 2 * k + 1] * view_y;
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
                tx[q] = cx[k]; ty[q] = cy[k]; ++q;
                if (dj < 0.0f) {
                    const float D = 1.0f / (dj - di);
                    tx[q] = (dj * cx[k] - di * cx[j]) * D; ty[q] = (dj * cy[k] - di * cy[j]) * D; ++q;
                }
            } else if (dj > 0.0f) {
                const float D = 1.0f / (di - dj);
                tx[q] = (di * cx[j] - dj * cx[k]) * D; ty[q] = (di * cy[j] - dj * cy[k]) * D; ++q;
            }
        }
        n = q;
        for (int k = 0; k < n; ++k) { cx[k] = tx[k]; cy[k] = ty[k]; }
    }
    TrianglePoly P;
    P.n = 0;
    if (n >= 3) {
        for (int k = 0; k < n; ++k) { P.x[k] = dep_snap(cx[k], wx16, x0); P.y[k] = dep_snap(cy[k], wy16, y0); }
        long long area2 = 0;
        for (int k = 0; k < n; ++k) {
            const int j = k + 1 == n ? 0 : k + 1;
            area2 += (long long)P.x[k] * P.y[j] - (long long)P.x[j] * P.y[k];
        }
        if (area2 != 0) {
            if (area2 > 0)
                for (int a = 0, b = n - 1; a < b; ++a, --b) {
                    int tmp = P.x[a]; P.x[a] = P.x[b]; P.x[b] = tmp;
                    tmp = P.y[a]; P.y[a] = P.y[b]; P.y[b] = tmp;
                }
            P.n = n;
        }
    }
    polys[t] = P;
}

__global__ __launch_bounds__(256) void triangle_fill_kernel(const TrianglePoly *polys, int ntri, float4 color, float4 *img, int w, int h)
{
    const uint32_t texels = (uint32_t)w * (uint32_t)h;
    for (uint32_t texel = blockIdx.x * 256u + threadIdx.x; texel < texels; texel += gridDim.x * 256u) {
        const int y = (int)(texel / (uint32_t)w), x = (int)(texel - (uint32_t)y * (uint32_t)w);
        float4 d = img[texel];
        bool touched = false;
        for (int t = 0; t < ntri; ++t) {
            const TrianglePoly &P = polys[t];
            int left = w, right = 0;
            for (int k = 0; k < P.n; ++k) {
                const int kn = k + 1 == P.n ? 0 : k + 1;
                const int Xa = P.x[k], Ya = P.y[k], Xb = P.x[kn], Yb = P.y[kn];
                if (Ya == Yb) continue;
                const bool swap = Yb < Ya;
                const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
                if (y < ((Y1 + 15) >> 4) || y >= ((Y2 + 15) >> 4)) continue;
                const long long DX = X2 - X1, DY = Y2 - Y1;
                long long e = dep_ceil_div(DX * (((long long)y << 4) - Y1) + (long long)X1 * DY, 16 * DY);
                if (e < 0) e = 0;
                if (e > w) e = w;
                if (swap) right = (int)e; else left = (int)e;
            }
            if (x >= left && x < right) { dep_blend_rgba(d, color); touched = true; }
        }
        if (touched) img[texel] = d;
    }
}

int deposit_grid(uint32_t n)
{
    static const uint32_t cap = [] { const char *e = getenv("TH_DEP_GRID"); return e ? (uint32_t)atoi(e) : 16384u; }();      // (8192: 1-2 % slower draws, 65536: the same)
    uint32_t g = (n + 255u) / 256u;
    return (int)(g < 1u ? 1u : (g > cap ? cap : g));
}

}  // namespace

uint32_t deposit_scan_words(uint32_t W, uint32_t rows) { return ((rows + kColRows - 1) / kColRows) * W + W; }

// words of the slow / long line lists: counters, then 2 * kDepLists segments of *cap entries
size_t deposit_list_words(uint32_t W, uint32_t rows, uint32_t *cap)
{
    const uint32_t groups = ((W + 255u) >> 8) * rows;
    *cap = ((groups + kDepLists - 1u) / kDepLists) * 256u;
    return (size_t)2 * kDepLists * kDepListStride + (size_t)2 * kDepLists * *cap;
}
size_t deposit_list_counter_bytes() { return (size_t)2 * kDepLists * kDepListStride * sizeof(uint32_t); }

void launch_deposit_count(const DepositParams &p, hipStream_t s)
{
    const uint32_t groups = ((p.W + 255u) >> 8) * p.rows;
    (void)hipMemsetAsync(p.list_n, 0, deposit_list_counter_bytes(), s);
    hipLaunchKernelGGL(deposit_raster_kernel, dim3(groups < 65536u * 16u ? (groups ? groups : 1u) : 65536u * 16u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(deposit_raster_slow_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
}

// scratch: deposit_scan_words(W, rows) words (row-block partial sums, then the column bases)
void launch_deposit_scan(const DepositParams &p, uint32_t *scratch, uint32_t *total, hipStream_t s)
{
    const uint32_t nrb = (p.rows + kColRows - 1) / kColRows;
    uint32_t *part = scratch, *colbase = scratch + (size_t)nrb * p.W;
    const dim3 grid((p.W + 255u) / 256u, nrb);
    hipLaunchKernelGGL(colscan_partial_kernel, grid, dim3(256), 0, s, p.count, p.W, p.rows, part);
    hipLaunchKernelGGL(colscan_prefix_kernel, dim3((p.W + 255u) / 256u), dim3(256), 0, s, part, p.W, nrb, colbase);
    hipLaunchKernelGGL(colscan_bases_kernel, dim3(1), dim3(1024), 0, s, p.W, colbase, total);
    hipLaunchKernelGGL(colscan_offsets_kernel, grid, dim3(256), 0, s, p.count, p.W, p.rows, part, colbase, p.offset);
}

void launch_deposit_scatter(const DepositParams &p, hipStream_t s)
{
    const uint32_t patches = ((p.W + kPatchCols - 1) / kPatchCols) * ((p.rows + kPatchRows - 1) / kPatchRows);
    hipLaunchKernelGGL(deposit_emit_kernel, dim3(patches < 65536u * 16u ? (patches ? patches : 1u) : 65536u * 16u), dim3(kPatchCols * 64u), 0, s, p);
    hipLaunchKernelGGL(deposit_emit_long_kernel, dim3(kDepLists * 8u), dim3(256), 0, s, p);
}

int deposit_key_bits(const DepositParams &p)
{
    const uint32_t texels = (uint32_t)p.fw * (uint32_t)p.fh;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < texels) ++bits;
    return bits;
}

void launch_export_mark(const DepositParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(export_lines_kernel<false>, dim3(deposit_grid(p.W * p.rows)), dim3(256), 0, s, p, (float *)nullptr);
}

void launch_export_write(const DepositParams &p, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(export_lines_kernel<true>, dim3(deposit_grid(p.W * p.rows)), dim3(256), 0, s, p, out);
}

void launch_triangles(const float *positions, int ntri, float view_x, float view_y, float4 color, TrianglePoly *polys,
                      float4 *img, int w, int h, hipStream_t s)
{
    if (ntri <= 0) return;
    hipLaunchKernelGGL(triangle_setup_kernel, dim3((ntri + 63) / 64), dim3(64), 0, s, positions, ntri, view_x, view_y, w, h, polys);
    hipLaunchKernelGGL(triangle_fill_kernel, dim3(deposit_grid((uint32_t)w * (uint32_t)h)), dim3(256), 0, s, polys, ntri, color, img, w, h);
}

void launch_deposit_gather_colors(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(deposit_gather_colors_kernel, dim3(deposit_grid(n)), dim3(256), 0, s, dst, src, index, n);
}

// sharded form: colours in arrival order, gathered into the sorted order first (temp: total float4)
void launch_deposit_blend64(float4 *flow, const unsigned long long *keys_sorted, const uint32_t *slots_sorted,
                            const float4 *colors, float4 *colors_sorted, uint32_t total, uint32_t *too_many, hipStream_t s)
{
    if (!total) return;
    launch_deposit_gather_colors(colors_sorted, colors, slots_sorted, total, s);
    hipLaunchKernelGGL((deposit_blend_kernel<FlowTarget, BandKeys>), dim3(deposit_grid(total)), dim3(256), 0, s, flow, BandKeys{keys_sorted},
                       (const float4 *)colors_sorted, 1u, total, too_many);
}

void launch_deposit_blend(const DepositParams &p, uint32_t total, hipStream_t s)
{
    const dim3 grid(deposit_grid(total));
    if (p.mode == 2) {      // both passes: one gather of the pairs, then each target's blend over its own half
        hipLaunchKernelGGL(deposit_gather_pairs_kernel, grid, dim3(256), 0, s, p.colors_sorted, (const float4 *)p.colors, (const uint32_t *)p.slots_sorted, total);
        hipLaunchKernelGGL((deposit_blend_kernel<FlowTarget, TexelKeys>), grid, dim3(256), 0, s, p.flow, TexelKeys{p.keys_sorted},
                           (const float4 *)p.colors_sorted, 2u, total, (uint32_t *)nullptr);
        hipLaunchKernelGGL((deposit_blend_kernel<ViewTarget, TexelKeys>), grid, dim3(256), 0, s, p.view, TexelKeys{p.keys_sorted},
                           (const float4 *)p.colors_sorted + 1, 2u, total, (uint32_t *)nullptr);
        return;
    }
    launch_deposit_gather_colors(p.colors_sorted, p.colors, p.slots_sorted, total, s);
    if (p.mode == 0) hipLaunchKernelGGL((deposit_blend_kernel<FlowTarget, TexelKeys>), grid, dim3(256), 0, s, p.flow,
                                        TexelKeys{p.keys_sorted}, (const float4 *)p.colors_sorted, 1u, total, (uint32_t *)nullptr);
    else hipLaunchKernelGGL((deposit_blend_kernel<ViewTarget, TexelKeys>), grid, dim3(256), 0, s, p.view,
                            TexelKeys{p.keys_sorted}, (const float4 *)p.colors_sorted, 1u, total, (uint32_t *)nullptr);
}

void launch_view_fill(uchar4 *view, size_t texels, float4 color, hipStream_t s)
{
    if (texels) hipLaunchKernelGGL(view_fill_kernel, dim3(deposit_grid((uint32_t)(texels < 0xffffffffull ? texels : 0xffffffffull))), dim3(256), 0, s, view, texels, color);
}

}  // namespace th
