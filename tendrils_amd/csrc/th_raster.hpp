// th_raster.hpp - device-side line machinery shared by the two draw() pipelines (th_deposit.hip: the stream-ordered
// pipeline; th_bins.hip: the binned pipeline over tile-sorted slots): the vertex stream of Particles.draw
// (src/particles.js:147-158, src/state/state-at-frame.glsl:12-22), the width-1 line rasteriser, the varyings of the flow
// pass (src/flow/vert/main.vert:10-17, apply/state.glsl:5-16) and of the view pass (src/render/index.vert:58-100), the
// per-line records and slow / long line lists, and the two blend targets.  Semantics as pinned in th_deposit.hip.
#pragma once
#include "th_kernels.hpp"
#include "th_logic.hpp"
#include "th_math.hpp"

namespace th {

// texel `at` of a state buffer of the pass: f32 texels, or what a packed ring's 8-byte texel decodes to (the lines are made of that)
TH_D float4 dep_state(const DepositParams &p, const float4 *buf, size_t at)
{
    if (p.packed) return unpack_state(reinterpret_cast<const uint2 *>(buf)[at]);
    return buf[at];
}
// where texel (row, col) of this context's rows lies in cur / prev: its index in texel order, or - a slot order, and a texel some
// other line's vertex reads - what LineSources tables (a texel it does not table: nowhere, ~0)
TH_D size_t dep_slot_of(const DepositParams &p, int row, int col)
{
    if (!p.src.slot) return (size_t)row * p.W + (size_t)col;
    const uint32_t ri = p.src.row_index[row], ci = p.src.col_index[col];
    if (ri != 0xffffu) return p.src.slot[(size_t)ri * p.W + (size_t)col];
    if (ci != 0xffffu) return p.src.slot[(size_t)p.src.nrows * p.W + (size_t)ci * p.rows + (size_t)row];
    return ~(size_t)0;
}

struct DepositVertex {
    bool live;
    float px, py;      // clip-space position (w = 1)
    float c[4];        // varying: (vel.x, vel.y, time, min(|vel|/speedLimit, 1)) - or the view pass's colour (mode 1)
    float c2[4];       // mode 2 (both passes of draw() in one): the view pass's colour beside the flow pass's varying
    bool from_cur;     // the vertex reads `current` (else `previous`)
    float uvx, uvy;    // its coordinates in the vertex stream (the colour map is looked up there)
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
// `own_row` (local) / `own_at`: the line's own particle - row and index in cur / prev.  The binned pipeline walks SLOTS (cur / prev in
// the ring's slot order): a vertex that is the line's own particle - every vertex, for shapes whose LUT does not drift
// off the line's texel (th_api.hip: lines_are_local) - is read at own_at; in texel order own_at is the texel index anyway.
// `own`: (optional) the own particle's texels of cur and prev, already loaded.
// the varyings of a vertex from its state texel `t` (dep_fetch computes them unless told to leave them for later: a pass that
// rasterises first needs them only for the lines that cover a texel at all)
TH_D void dep_vertex_colors(const DepositParams &p, float4 t, DepositVertex &v)
{
    if (p.mode != 1) {
        v.c[0] = t.z; v.c[1] = t.w; v.c[2] = p.time;
        v.c[3] = __builtin_fminf(__builtin_sqrtf(t.z * t.z + t.w * t.w) / p.speed_limit, 1.0f);
        if (p.mode == 2) dep_render_color(p, t, v.uvx, v.uvy, v.c2);
    } else dep_render_color(p, t, v.uvx, v.uvy, v.c);
}

// the line's own texel of both buffers, when the caller holds them already (by VALUE, and chosen between with a select of values:
// handed over as a pointer to a two-element array they lived in scratch memory - 64 bytes written and read back per line of a
// pass over sixteen million, a quarter of what the binned pass's first kernel wrote)
struct OwnTexels { bool have = false; float4 cur{}, prev{}; };
// PLAIN: f32 texels, and a texel's place in cur / prev is its index or the line's own slot - no packed ring, no table of where
// other lines' particles lie (the binned pass's kernel for the shapes of the frame loop it was tuned on: nothing it does not need)
template <bool PLAIN = false>
TH_D DepositVertex dep_fetch(const DepositParams &p, uint32_t i, uint32_t j, uint32_t own_row, size_t own_at, const OwnTexels own = OwnTexels{}, bool colors = true)
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
    const bool self = row == (int)own_row && col == (int)i;
    float4 t;
    if constexpr (PLAIN) {
        // (one unconditional load from a selected address: a load under a branch is awaited at the join, and the second
        // vertex's load would only go out after the first had come back)
        size_t at = (size_t)(row < 0 ? 0 : (row < (int)p.rows ? row : (int)p.rows - 1)) * W + col;
        if (self) at = own_at;
        const float4 *from = tex + at;
        if (!(row >= 0 && row < (int)p.rows)) {
            if (row == -1 && p.halo_lo) from = p.halo_lo + (offset > 0.25f ? 0 : W) + col;
            else if (row == (int)p.rows && p.halo_hi) from = p.halo_hi + (offset > 0.25f ? 0 : W) + col;
            else *p.oob = 1u;
        }
        if (own.have && self) {
            const bool c = offset > 0.25f;
            t = make_float4(c ? own.cur.x : own.prev.x, c ? own.cur.y : own.prev.y, c ? own.cur.z : own.prev.z, c ? own.cur.w : own.prev.w);
        } else t = *from;
    } else {
        const bool inside = row >= 0 && row < (int)p.rows;
        size_t at = self ? own_at : dep_slot_of(p, row < 0 ? 0 : (row < (int)p.rows ? row : (int)p.rows - 1), col);
        if (at == ~(size_t)0) { at = own_at; if (inside) *p.oob = 1u; }        // (a texel the tables should hold and do not: the pass is refused)
        const float4 *halo = nullptr;                            // (the neighbouring bands' edge rows: always f32, in texel order)
        if (!inside) {
            if (row == -1 && p.halo_lo) halo = p.halo_lo + (offset > 0.25f ? 0 : W) + col;
            else if (row == (int)p.rows && p.halo_hi) halo = p.halo_hi + (offset > 0.25f ? 0 : W) + col;
            else *p.oob = 1u;
        }
        if (own.have && self) {
            const bool c = offset > 0.25f;
            t = make_float4(c ? own.cur.x : own.prev.x, c ? own.cur.y : own.prev.y, c ? own.cur.z : own.prev.z, c ? own.cur.w : own.prev.w);
        } else t = halo ? *halo : dep_state(p, tex, at);
    }
    DepositVertex v;
    v.live = (t.x != kInert) || (t.y != kInert);
    v.px = t.x * p.view_x;
    v.py = t.y * p.view_y;
    v.from_cur = offset > 0.25f; v.uvx = uvx; v.uvy = uvy;
    if (colors) dep_vertex_colors(p, t, v);
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
    int n;                     // polygon vertices after clipping (the vertices: PolygonWords)
};

// The words a clipped polygon lives in while it is indexed at run time (the clipper's two vertex lists of up to 12 points, then
// the snapped vertices in the second one's place): a column of a table in LDS - in a thread's own arrays they were scratch
// memory, hundreds of nanoseconds per indexed access instead of tens (the lines that cross the view's edge are few, but a
// kernel of them stands alone between two passes of a draw: th_bins.hip bins_listed_kernel 76 -> 54 us).
template <int STRIDE>
struct LdsWords {
    float *column;             // word k of this thread's 48: column[k * STRIDE]
    TH_D float &f(int k) const { return column[k * STRIDE]; }
    TH_D int &i(int k) const { return reinterpret_cast<int *>(column)[k * STRIDE]; }
};
template <typename Words>
struct PolygonX { Words &w; TH_D int operator[](int k) const { return w.i(24 + k); } };
template <typename Words>
struct PolygonY { Words &w; TH_D int operator[](int k) const { return w.i(36 + k); } };

// everything about line `id` (stream index = i*H + m) that does not depend on the texel, except the polygon
template <bool PLAIN = false>
TH_D void dep_setup(const DepositParams &p, uint32_t i, uint32_t m, DepositLine &L, size_t own_at, const OwnTexels own = OwnTexels{}, bool colors = true)
{
    L.draws = false;
    L.short32 = false;
    L.n = 0;
    L.a = dep_fetch<PLAIN>(p, i, 2u * m, m - p.row0, own_at, own, colors);
    L.b = dep_fetch<PLAIN>(p, i, 2u * m + 1u, m - p.row0, own_at, own, colors);
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

// The line as the hexagon of its two endpoint diamonds, in clip space (six vertices, statically indexed).  Width 1 (every
// pinned case: the captured GL clamps gl.lineWidth to 1): diamonds of half a texel.  A wider line (th_line_width on a
// context whose th_line_width_range lets it through; unpinned) is the same construction with the diamonds scaled by the
// width: the hexagon then measures `width` texels across in the line's minor direction, as the GL specification asks of a
// non-antialiased wide line, and everything downstream (clipping, snapping, scan conversion, the varying projected on
// the line) is the same code.
// Returns kHexInside when all of it lies inside the view volume (it is rasterised as it stands), kHexOutside when all
// six vertices are beyond ONE of the four planes (the clipper would then leave nothing: 44 % of the lines of the C3
// bench, whose particles are spread over twice the view's height), else kHexClip.
enum { kHexInside = 0, kHexClip = 1, kHexOutside = 2 };
TH_D int dep_hexagon(const DepositParams &p, const DepositLine &L, float (&cx)[6], float (&cy)[6])
{
    const float fw = (float)p.fw, fh = (float)p.fh;
    const float dx = (0.5f * fw) * (L.b.px - L.a.px), dy = (0.5f * fh) * (L.b.py - L.a.py);
    const float hx = p.line_half / (0.5f * fw), hy = p.line_half / (0.5f * fh);      // half the diamond (half a texel x width) in clip space
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

// does the part of the hexagon inside the view span more than kSpanTexels rows or columns of the target?  (a wave's work then,
// not a lane's: th_bins.hip bins_span_lines)
constexpr float kSpanTexels = 24.0f;
TH_D bool dep_hexagon_spans(const DepositParams &p, const float (&cx)[6], const float (&cy)[6])
{
    float x0 = cx[0], x1 = cx[0], y0 = cy[0], y1 = cy[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) { x0 = __builtin_fminf(x0, cx[k]); x1 = __builtin_fmaxf(x1, cx[k]); y0 = __builtin_fminf(y0, cy[k]); y1 = __builtin_fmaxf(y1, cy[k]); }
    x0 = __builtin_fmaxf(x0, -1.0f); x1 = __builtin_fminf(x1, 1.0f); y0 = __builtin_fmaxf(y0, -1.0f); y1 = __builtin_fminf(y1, 1.0f);
    return (x1 - x0) * (0.5f * (float)p.fw) > kSpanTexels || (y1 - y0) * (0.5f * (float)p.fh) > kSpanTexels;
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
template <typename Words>
TH_D void dep_clip_hexagon(const DepositParams &p, DepositLine &L, const float (&hx6)[6], const float (&hy6)[6], Words &w)
{
    const float wx16 = 8.0f * (float)p.fw, wy16 = 8.0f * (float)p.fh;
    const float x0 = wx16 - 8.0f, y0 = wy16 - 8.0f;
    struct At { Words &w; int first; TH_D float &operator[](int k) const { return w.f(first + k); } };
    const At cx{w, 0}, cy{w, 12}, tx{w, 24}, ty{w, 36};
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
    for (int k = 0; k < n; ++k) { w.i(24 + k) = dep_snap(cx[k], wx16, x0); w.i(36 + k) = dep_snap(cy[k], wy16, y0); }      // (PolygonX / PolygonY)
}

// scan conversion: calls emit(x, y) for every covered texel.  Edges going up in y set `left`, edges going down set
// `right` (a later edge overwrites an earlier one on the same row, as in the captured rasteriser); texels
// left <= x < right.  Rows are walked in windows so that arbitrarily long lines need no large arrays.
template <int N, typename VX, typename VY, typename Emit>
TH_D void dep_raster_poly(const DepositParams &p, const VX &PX, const VY &PY, int count, Emit emit)
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
                auto pick = [](const auto &v, int i) { int r = v[0]; r = i == 1 ? v[1] : r; r = i == 2 ? v[2] : r; r = i == 3 ? v[3] : r;
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

// ... the same spans with TWO divisions per row instead of six: every row of a polygon whose vertices run once up and once
// down in y is crossed by exactly one edge going up and one going down (half-open row ranges [ceil(Y1/16), ceil(Y2/16)):
// the edges of a chain take the rows in turn), so the row first selects its two edges - in vertex order, a later edge
// over an earlier one, as the span assignment above does - and then divides.  Same quotients, same spans.
template <typename Emit>
TH_D void dep_raster_small_hexagon2(const DepositParams &p, const int (&PX)[6], const int (&PY)[6], int ymin, int ymax, Emit emit)
{
    int r0 = (ymin + 15) >> 4, r1 = (ymax + 15) >> 4;
    if (r0 < 0) r0 = 0;
    if (r1 > p.fh) r1 = p.fh;
    int X1[6], Y1[6], DX[6], DY[6], e0[6], en[6];
    bool up[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int kn = k == 5 ? 0 : k + 1;
        const int Xa = PX[k], Ya = PY[k], Xb = PX[kn], Yb = PY[kn];
        const bool swap = Yb < Ya;
        up[k] = !swap;
        X1[k] = swap ? Xb : Xa; Y1[k] = swap ? Yb : Ya;
        DX[k] = (swap ? Xa : Xb) - X1[k]; DY[k] = (swap ? Ya : Yb) - Y1[k];
        e0[k] = (Y1[k] + 15) >> 4; en[k] = ((Y1[k] + DY[k] + 15) >> 4) - e0[k];           // rows [e0, e0 + en); none for Ya == Yb
    }
    // (every factor fits 24 bits here - |dx| < 4096, |16 y - y1| < 2048, |x1| < 2^19, dy < 1024, |q| < 2^16, den < 2^14 -
    // and every product 31: the 24-bit multiplier gives the same integers at the full VALU rate, v_mul_lo_u32 at a quarter)
    auto ceil_at = [&](int y, int x1, int y1, int dx, int dy) {
        const int den = dy > 0 ? 16 * dy : 16;
        const int num = __mul24(dx, (y << 4) - y1) + __mul24(x1, dy);
        int q = (int)__builtin_floorf((float)num * __builtin_amdgcn_rcpf((float)den));
        int r = num - __mul24(q, den);
        if (r < 0) { --q; r += den; }
        if (r >= den) { ++q; r -= den; }
        int x = r > 0 ? q + 1 : q;
        return x < 0 ? 0 : (x > p.fw ? p.fw : x);
    };
    for (int y = r0; y < r1; ++y) {
        int lx = 0, ly = 0, ldx = 0, ldy = 0, rx = 0, ry = 0, rdx = 0, rdy = 0;
        bool hl = false, hr = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool crosses = (unsigned)(y - e0[k]) < (unsigned)en[k];
            const bool l = crosses && up[k], r = crosses && !up[k];
            lx = l ? X1[k] : lx; ly = l ? Y1[k] : ly; ldx = l ? DX[k] : ldx; ldy = l ? DY[k] : ldy; hl = hl || l;
            rx = r ? X1[k] : rx; ry = r ? Y1[k] : ry; rdx = r ? DX[k] : rdx; rdy = r ? DY[k] : rdy; hr = hr || r;
        }
        const int left = hl ? ceil_at(y, lx, ly, ldx, ldy) : p.fw, right = hr ? ceil_at(y, rx, ry, rdx, rdy) : 0;
        for (int x = left; x < right; ++x) emit(x, y);
    }
}

// ... and ONE row of it: the span [left, right) of row y of the small hexagon (the same edge selection, the same
// quotients).  For passes that deal a wave's (line, row) pairs evenly to its lanes instead of letting every lane walk
// the rows of its own line - a wave then runs as many rows as its lines have, not 64 times as many as its longest line.
TH_D void dep_hexagon_row_span(const DepositParams &p, const int (&PX)[6], const int (&PY)[6], int y, int &left, int &right)
{
    int lx = 0, ly = 0, ldx = 0, ldy = 0, rx = 0, ry = 0, rdx = 0, rdy = 0;
    bool hl = false, hr = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int kn = k == 5 ? 0 : k + 1;
        const int Xa = PX[k], Ya = PY[k], Xb = PX[kn], Yb = PY[kn];
        const bool swap = Yb < Ya;
        const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, DX = (swap ? Xa : Xb) - X1, DY = (swap ? Ya : Yb) - Y1;
        const int e0 = (Y1 + 15) >> 4, en = ((Y1 + DY + 15) >> 4) - e0;
        const bool crosses = (unsigned)(y - e0) < (unsigned)en;
        const bool l = crosses && !swap, r = crosses && swap;
        lx = l ? X1 : lx; ly = l ? Y1 : ly; ldx = l ? DX : ldx; ldy = l ? DY : ldy; hl = hl || l;
        rx = r ? X1 : rx; ry = r ? Y1 : ry; rdx = r ? DX : rdx; rdy = r ? DY : rdy; hr = hr || r;
    }
    auto ceil_at = [&](int x1, int y1, int dx, int dy) {
        const int den = dy > 0 ? 16 * dy : 16;
        const int num = __mul24(dx, (y << 4) - y1) + __mul24(x1, dy);
        int q = (int)__builtin_floorf((float)num * __builtin_amdgcn_rcpf((float)den));
        int r = num - __mul24(q, den);
        if (r < 0) { --q; r += den; }
        if (r >= den) { ++q; r -= den; }
        int x = r > 0 ? q + 1 : q;
        return x < 0 ? 0 : (x > p.fw ? p.fw : x);
    };
    left = hl ? ceil_at(lx, ly, ldx, ldy) : p.fw;
    right = hr ? ceil_at(rx, ry, rdx, rdy) : 0;
}

// a line, whichever way it has to go: straight from its hexagon, or clipped first (`words`: where the clipped polygon lives)
template <typename Emit, typename Words>
TH_D void dep_raster_line(const DepositParams &p, DepositLine &L, Emit emit, Words &words)
{
    float cx[6], cy[6];
    const int where = dep_hexagon(p, L, cx, cy);
    if (where == kHexInside) {
        int PX[6], PY[6];
        dep_snap_hexagon(p, cx, cy, PX, PY);
        dep_raster_poly<6>(p, PX, PY, 6, emit);
    } else if (where == kHexClip) {
        dep_clip_hexagon(p, L, cx, cy, words);
        if (L.draws) dep_raster_poly<0>(p, PolygonX<Words>{words}, PolygonY<Words>{words}, L.n, emit);
    }
}


// the varying of line L at texel (x, y): linear along the snapped endpoints, extrapolated, unclamped.  dep_param: the
// interpolation parameter (false: both endpoints snap to the same point, the first vertex's value is taken)
TH_D bool dep_param(const DepositLine &L, int x, int y, float &t)
{
    if (L.short32) {        // the same integers in 32 bits (a fragment lies within half the line's width + a texel of its line): the same floats
        // (|ex|, |ey| < 2^14 and a fragment within 33 texels of its line - widths up to kMaxLineWidth: 24-bit factors, 30-bit sums)
        const int ex = L.sx[1] - L.sx[0], ey = L.sy[1] - L.sy[0], den = __mul24(ex, ex) + __mul24(ey, ey);
        if (den == 0) return false;
        const int num = __mul24((x << 4) - L.sx[0], ex) + __mul24((y << 4) - L.sy[0], ey);
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

// the workgroups of a slow kernel: (segment, part) = (block % kDepLists, block / kDepLists) of `blocks` workgroups (by
// default the whole grid)
template <typename Work>
TH_D void dep_list_work(const DepositParams &p, uint32_t which, Work work, uint32_t block = blockIdx.x, uint32_t blocks = gridDim.x)
{
    const uint32_t seg = block & (kDepLists - 1u), part = block / kDepLists, parts = blocks / kDepLists;
    const uint32_t n = p.list_n[(which * kDepLists + seg) * kDepListStride];
    const uint32_t *list = p.lists + ((size_t)which * kDepLists + seg) * p.list_cap;
    for (uint32_t e0 = part * 256u; e0 < n; e0 += parts * 256u) {         // whole waves stay together (the appends ballot)
        const uint32_t e = e0 + threadIdx.x;
        work(e < n, e < n ? list[e] : 0u, seg);
    }
}
enum { kListSlow = 0, kListLong = 1, kListSpan = 2, kListKinds = 3 };      // (kListSpan: the binned pipeline's lines that span many rows or columns - th_bins.hip)

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
// every lane turning its own fragment into its side of the blend, and the destination's four channels applied side by
// side, a lane each - the same operations in the same order, at 64 fragments per memory round trip instead of 4 and a
// quarter of the chain's instructions.
constexpr int kShortRun = 16;

// a fragment's side of the blend (everything that does not depend on the destination), and the destination's
struct BlendSource { float x, y, z, w, da; };
TH_D float lane_float(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
struct FlowTarget {                     // dep_blend in two halves
    using Texel = float4;
    TH_D static BlendSource source(float4 c) { const float sa = c.w; return BlendSource{c.x * sa, c.y * sa, c.z * sa, c.w * sa, 1.0f - sa}; }
    TH_D static void apply(float4 &d, const BlendSource &s) { d.x = s.x + d.x * s.da; d.y = s.y + d.y * s.da; d.z = s.z + d.z * s.da; d.w = s.w + d.w * s.da; }
    // one channel of the blend (apply(), component by component): what a lane does when a long run's channels are applied side by side
    TH_D static float channel(float4 d, uint32_t c) { return c == 0u ? d.x : (c == 1u ? d.y : (c == 2u ? d.z : d.w)); }
    TH_D static void apply_channel(float &d, float src, float da) { d = src + d * da; }
    TH_D static float4 from_channels(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
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
    // ... with the texel held as four integer-valued floats while a run is blended into it: the same arithmetic, the round trip
    // through bytes (convert, pack, unpack, convert) taken out of the dependent chain - the byte the cast gives for a
    // non-negative x is floor(x), and the float of that byte is floor(x) again.  (The chain of one texel's run is what a
    // crowded frame waits for: thousands of fragments applied one after the other.)
    TH_D static float4 unpack(uchar4 q) { return make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w); }
    TH_D static uchar4 pack(float4 f) { return make_uchar4((unsigned char)f.x, (unsigned char)f.y, (unsigned char)f.z, (unsigned char)f.w); }
    TH_D static void apply_unpacked(float4 &q, const BlendSource &s)
    {
        const float k = 1.0f / 255.0f;
        auto mix8 = [&](float src, float dst) {
            const float o = src + (dst * k) * s.da;
            return __builtin_floorf(__builtin_fminf(__builtin_fmaxf(o, 0.0f), 1.0f) * 255.0f + 0.5f);
        };
        q = make_float4(mix8(s.x, q.x), mix8(s.y, q.y), mix8(s.z, q.z), mix8(s.w, q.w));
    }
    TH_D static float channel(uchar4 q, uint32_t c) { return (float)(c == 0u ? q.x : (c == 1u ? q.y : (c == 2u ? q.z : q.w))); }
    TH_D static void apply_channel(float &q, float src, float da)          // (apply_unpacked, one channel)
    {
        const float o = src + (q * (1.0f / 255.0f)) * da;
        q = __builtin_floorf(__builtin_fminf(__builtin_fmaxf(o, 0.0f), 1.0f) * 255.0f + 0.5f);
    }
    TH_D static uchar4 from_channels(float x, float y, float z, float w) { return make_uchar4((unsigned char)x, (unsigned char)y, (unsigned char)z, (unsigned char)w); }
    TH_D static uchar4 *plane(const DepositParams &p) { return p.view; }
    TH_D static uchar4 from_lane(uchar4 d, int lane) { return __builtin_bit_cast(uchar4, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), lane)); }
};

}  // namespace th
