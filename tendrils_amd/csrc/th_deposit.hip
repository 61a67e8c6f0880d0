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
//   1. deposit_raster_kernel<false>: one thread per line, rasterise, store the line's fragment count at its stream
//      index (no atomics)
//   2. exclusive scan over the stream -> every line's first fragment slot: the fragment array is in stream order
//   3. deposit_raster_kernel<true>: rasterise again, write (texel, interpolated varying) into the line's slots
//   4. stable radix sort of the fragments by texel (rocPRIM through hipCUB): each texel's fragments end up
//      contiguous and still in stream order
//   5. deposit_blend_kernel: the thread at the head of a texel's run walks it and blends sequentially:
//      dst = src*a + dst*(1-a), exactly GL's order and arithmetic.
// No step depends on thread scheduling, and the cost does not depend on how crowded single texels are (the
// wake makes particles converge: thousands of fragments in one texel are normal after a few dozen frames).
#include <hipcub/hipcub.hpp>

#include "th_kernels.hpp"
#include "th_math.hpp"

namespace th {
namespace {

struct DepositVertex {
    bool live;
    float px, py;      // clip-space position (w = 1)
    float c[4];        // varying: (vel.x, vel.y, time, min(|vel|/speedLimit, 1))
};

TH_D int dep_nearest(float u, int n)       // NEAREST + CLAMP_TO_EDGE on a float texture
{
    float f = th_floor(u * (float)n);
    if (!(f > 0.0f)) return 0;
    if (f > (float)(n - 1)) return n - 1;
    return (int)f;
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
    float4 t;
    if (row >= 0 && row < (int)p.rows) t = tex[(size_t)row * W + col];
    else if (row == -1 && p.halo_lo) t = p.halo_lo[(offset > 0.25f ? 0 : W) + col];
    else if (row == (int)p.rows && p.halo_hi) t = p.halo_hi[(offset > 0.25f ? 0 : W) + col];
    else { *p.oob = 1u; t = tex[(size_t)(row < 0 ? 0 : (int)p.rows - 1) * W + col]; }
    DepositVertex v;
    v.live = (t.x != kInert) || (t.y != kInert);
    v.px = t.x * p.view_x;
    v.py = t.y * p.view_y;
    v.c[0] = t.z; v.c[1] = t.w; v.c[2] = p.time;
    v.c[3] = __builtin_fminf(__builtin_sqrtf(t.z * t.z + t.w * t.w) / p.speed_limit, 1.0f);
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
    DepositVertex a, b;
    int sx[2], sy[2];          // snapped endpoints (1/16 texel, texel centres at multiples of 16)
    int n;                     // polygon vertices after clipping
    int PX[12], PY[12];
};

// everything about line `id` (stream index = i*H + m) that does not depend on the texel
TH_D void dep_setup(const DepositParams &p, uint32_t id, DepositLine &L, bool need_polygon)
{
    const uint32_t i = id / p.H, m = id - i * p.H;
    L.draws = false;
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
    L.draws = true;
    if (!need_polygon) return;

    const float hx = 0.5f / (0.5f * fw), hy = 0.5f / (0.5f * fh);      // half a texel in clip space
    float cx[12], cy[12], tx[12], ty[12];
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
    int n = 6;
    bool inside = true;
    for (int k = 0; k < 6; ++k)
        inside = inside && (1.0f + cx[k] >= 0.0f) && (1.0f - cx[k] >= 0.0f) && (1.0f - cy[k] >= 0.0f) && (1.0f + cy[k] >= 0.0f);
    if (!inside) {
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
        if (n < 3) { L.draws = false; return; }
    }
    L.n = n;
    for (int k = 0; k < n; ++k) { L.PX[k] = dep_snap(cx[k], wx16, x0); L.PY[k] = dep_snap(cy[k], wy16, y0); }
}

// scan conversion: calls emit(x, y) for every covered texel.  Edges going up in y set `left`, edges going down set
// `right` (a later edge overwrites an earlier one on the same row, as in the captured rasteriser); texels
// left <= x < right.  Rows are walked in windows so that arbitrarily long lines need no large arrays.
template <typename Emit>
TH_D void dep_raster(const DepositParams &p, const DepositLine &L, Emit emit)
{
    int ymin = L.PY[0], ymax = L.PY[0];
    for (int k = 1; k < L.n; ++k) { ymin = L.PY[k] < ymin ? L.PY[k] : ymin; ymax = L.PY[k] > ymax ? L.PY[k] : ymax; }
    int r0 = (ymin + 15) >> 4, r1 = (ymax + 15) >> 4;
    if (r0 < 0) r0 = 0;
    if (r1 > p.fh) r1 = p.fh;
    constexpr int kWindow = 8;
    for (int base = r0; base < r1; base += kWindow) {
        const int top = base + kWindow < r1 ? base + kWindow : r1;
        int left[kWindow], right[kWindow];
#pragma unroll
        for (int k = 0; k < kWindow; ++k) { left[k] = p.fw; right[k] = 0; }
        for (int k = 0; k < L.n; ++k) {
            const int kn = k + 1 == L.n ? 0 : k + 1;
            const int Xa = L.PX[k], Ya = L.PY[k], Xb = L.PX[kn], Yb = L.PY[kn];
            if (Ya == Yb) continue;
            const bool swap = Yb < Ya;
            const int X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
            int e0 = (Y1 + 15) >> 4, e1 = (Y2 + 15) >> 4;
            if (e0 < base) e0 = base;
            if (e1 > top) e1 = top;
            const long long DX = X2 - X1, DY = Y2 - Y1;
            // short edges inside a 32768-texel-wide view (every practical case): the same quotient in 32-bit arithmetic
            const bool small = DY < 1024 && DX > -4096 && DX < 4096 && X1 > -(1 << 19) && X1 < (1 << 19);
            for (int y = e0; y < e1; ++y) {
                long long x;
                if (small) {
                    const int num = (int)DX * ((y << 4) - Y1) + X1 * (int)DY, den = 16 * (int)DY;
                    int q = num / den;
                    if (num % den > 0) ++q;
                    x = q;
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

// the varying of line L at texel (x, y): linear along the snapped endpoints, extrapolated, unclamped
TH_D float4 dep_varying(const DepositLine &L, int x, int y)
{
    const long long ex = L.sx[1] - L.sx[0], ey = L.sy[1] - L.sy[0], den = ex * ex + ey * ey;
    if (den == 0) return make_float4(L.a.c[0], L.a.c[1], L.a.c[2], L.a.c[3]);
    const long long num = ((long long)(x << 4) - L.sx[0]) * ex + ((long long)(y << 4) - L.sy[0]) * ey;
    const float t = (float)num / (float)den;
    return make_float4(L.a.c[0] + t * (L.b.c[0] - L.a.c[0]), L.a.c[1] + t * (L.b.c[1] - L.a.c[1]),
                       L.a.c[2] + t * (L.b.c[2] - L.a.c[2]), L.a.c[3] + t * (L.b.c[3] - L.a.c[3]));
}

// passes 1 and 3: count the line's fragments, or write them into its slots of the stream-ordered fragment array
template <bool SCATTER>
__global__ __launch_bounds__(256) void deposit_raster_kernel(const DepositParams p)
{
    const uint32_t lines = p.W * p.rows;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < lines; t += gridDim.x * 256u) {
        // threads walk the particle texture row-major (coalesced state reads; walking the column-major stream
        // instead makes the fragment writes contiguous but the state reads strided: 3.6 -> 5.3 ms per draw at C3);
        // the line's place in the fragment array is its position in the vertex stream
        const uint32_t row = t / p.W, col = t - row * p.W;
        const uint32_t id = col * p.H + p.row0 + row;            // position in the whole texture's vertex stream
        const uint32_t local = col * p.rows + row;               // ... and among this band's lines (same order)
        DepositLine L;
        dep_setup(p, id, L, true);
        if constexpr (SCATTER) {
            if (!L.draws) continue;
            uint32_t at = p.offset[local];
            dep_raster(p, L, [&](int x, int y) {
                const uint32_t texel = (uint32_t)y * (uint32_t)p.fw + (uint32_t)x;
                if (p.keys64) p.keys64[at] = ((unsigned long long)texel << 32) | id;
                else p.keys[at] = texel;
                p.slots[at] = at;
                p.colors[at] = dep_varying(L, x, y);
                ++at;
            });
        } else {
            uint32_t n = 0;
            if (L.draws) dep_raster(p, L, [&](int, int) { ++n; });
            p.count[local] = n;
        }
    }
}

TH_D void dep_blend_rgba(float4 &d, float4 c) { const float sa = c.w, da = 1.0f - sa; d.x = c.x * sa + d.x * da; d.y = c.y * sa + d.y * da; d.z = c.z * sa + d.z * da; d.w = c.w * sa + d.w * da; }

TH_D void dep_blend(float4 &d, float4 c)
{
    const float sa = c.w, da = 1.0f - sa;
    d.x = c.x * sa + d.x * da;
    d.y = c.y * sa + d.y * da;
    d.z = c.z * sa + d.z * da;
    d.w = c.w * sa + d.w * da;
}

// pass 5: fragments sorted by texel (stable: stream order inside a texel); the head of each run blends it
__global__ __launch_bounds__(256) void deposit_blend_kernel(const DepositParams p, uint32_t total)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t texel = p.keys_sorted[i];
        if (i > 0 && p.keys_sorted[i - 1] == texel) continue;
        float4 d = p.flow[texel];
        // the run's varyings are contiguous (gathered into sorted order): read four ahead of the dependent blends
        uint32_t j = i;
        while (true) {
            float4 c[4];
            uint32_t k[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t at = j + (uint32_t)q < total ? j + (uint32_t)q : total - 1u;
                c[q] = p.colors_sorted[at];
                k[q] = p.keys_sorted[at];
            }
            bool done = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!done && j + (uint32_t)q < total && k[q] == texel) dep_blend(d, c[q]);
                else done = true;
            }
            if (done) break;
            j += 4u;
        }
        p.flow[texel] = d;
    }
}

// sharded form: the same walk over fragments sorted by (texel, global stream index)
__global__ __launch_bounds__(256) void deposit_blend64_kernel(float4 *flow, const unsigned long long *keys, const uint32_t *slots,
                                                              const float4 *colors, uint32_t total)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t texel = (uint32_t)(keys[i] >> 32);
        if (i > 0 && (uint32_t)(keys[i - 1] >> 32) == texel) continue;
        float4 d = flow[texel];
        uint32_t j = i;
        do {
            dep_blend(d, colors[slots[j]]);
            ++j;
        } while (j < total && (uint32_t)(keys[j] >> 32) == texel);
        flow[texel] = d;
    }
}

__global__ __launch_bounds__(256) void deposit_iota_kernel(uint32_t *dst, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = i;
}

__global__ __launch_bounds__(256) void deposit_gather_colors_kernel(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[index[i]];
}

// ---- exclusive scan of the per-line fragment counts (three small kernels; 1024 elements per block) -------------
constexpr uint32_t kScanBlock = 1024;

__global__ __launch_bounds__(256) void scan_local_kernel(const uint32_t *in, uint32_t *out, uint32_t *block_sums, uint32_t n)
{
    __shared__ uint32_t sh[256];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {           // Hillis-Steele over the 256 thread sums
        uint32_t add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;                 // exclusive prefix of this thread inside the block
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = sh[255];
}

__global__ __launch_bounds__(256) void scan_blocks_kernel(uint32_t *block_sums, uint32_t nblocks, uint32_t *total)
{
    __shared__ uint32_t carry;
    __shared__ uint32_t sh[256];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256u) {
        const uint32_t idx = base + threadIdx.x;
        const uint32_t v = idx < nblocks ? block_sums[idx] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 256u; o <<= 1) {
            uint32_t add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (idx < nblocks) block_sums[idx] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *out, const uint32_t *block_sums, uint32_t n)
{
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) out[base + k] += add;
}

// ---- trail export: the line list of draw() (12 floats per line, stream order) ---------------------------------
// pass 1 marks the lines that exist (two live vertices, non-zero length), the scan places them, pass 2 writes them
template <bool WRITE>
__global__ __launch_bounds__(256) void export_lines_kernel(const DepositParams p, float *out)
{
    const uint32_t lines = p.W * p.rows;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < lines; t += gridDim.x * 256u) {
        const uint32_t row = t / p.W, col = t - row * p.W;
        const uint32_t id = col * p.H + p.row0 + row, local = col * p.rows + row;
        const uint32_t i = id / p.H, m = id - i * p.H;
        const DepositVertex a = dep_fetch(p, i, 2u * m), b = dep_fetch(p, i, 2u * m + 1u);
        const bool exists = a.live && b.live && !(a.px == b.px && a.py == b.py);
        if constexpr (WRITE) {
            if (!exists) continue;
            float *o = out + 12ull * p.offset[local];
            o[0] = a.px; o[1] = a.py; o[2] = b.px; o[3] = b.py;
            for (int k = 0; k < 4; ++k) { o[4 + k] = a.c[k]; o[8 + k] = b.c[k]; }
        } else {
            p.count[local] = exists ? 1u : 0u;
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
    uint32_t g = (n + 255u) / 256u;
    return (int)(g < 1u ? 1u : (g > 8192u ? 8192u : g));
}

}  // namespace

uint32_t deposit_scan_blocks(uint32_t n) { return (n + kScanBlock - 1) / kScanBlock; }

void launch_deposit_count(const DepositParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(deposit_raster_kernel<false>, dim3(deposit_grid(p.W * p.rows)), dim3(256), 0, s, p);
}

void launch_deposit_scan(const DepositParams &p, uint32_t *block_sums, uint32_t *total, hipStream_t s)
{
    const uint32_t lines = p.W * p.rows, nb = deposit_scan_blocks(lines);
    hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(256), 0, s, p.count, p.offset, block_sums, lines);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, s, block_sums, nb, total);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, s, p.offset, block_sums, lines);
}

void launch_deposit_scatter(const DepositParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(deposit_raster_kernel<true>, dim3(deposit_grid(p.W * p.rows)), dim3(256), 0, s, p);
}

static int deposit_key_bits(const DepositParams &p)
{
    const uint32_t texels = (uint32_t)p.fw * (uint32_t)p.fh;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < texels) ++bits;
    return bits;
}

size_t deposit_sort_temp_bytes(const DepositParams &p, uint32_t total)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, p.keys, p.keys_sorted, p.slots, p.slots_sorted, (int)total, 0,
                                             deposit_key_bits(p), (hipStream_t) nullptr);
    return bytes;
}

hipError_t launch_deposit_sort(const DepositParams &p, uint32_t total, void *temp, size_t temp_bytes, hipStream_t s)
{
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, p.keys, p.keys_sorted, p.slots, p.slots_sorted, (int)total, 0,
                                              deposit_key_bits(p), s);
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

size_t deposit_sort64_temp_bytes(uint32_t total, int begin_bit, int end_bit)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)total, begin_bit, end_bit,
                                             (hipStream_t) nullptr);
    return bytes;
}

hipError_t launch_deposit_sort64(const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                                 uint32_t *vals_out, uint32_t total, int begin_bit, int end_bit, void *temp, size_t temp_bytes, hipStream_t s)
{
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)total, begin_bit, end_bit, s);
}

void launch_deposit_iota(uint32_t *dst, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(deposit_iota_kernel, dim3(deposit_grid(n)), dim3(256), 0, s, dst, n);
}

void launch_deposit_gather_colors(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(deposit_gather_colors_kernel, dim3(deposit_grid(n)), dim3(256), 0, s, dst, src, index, n);
}

void launch_deposit_blend64(float4 *flow, const unsigned long long *keys_sorted, const uint32_t *slots_sorted,
                            const float4 *colors, uint32_t total, hipStream_t s)
{
    if (total) hipLaunchKernelGGL(deposit_blend64_kernel, dim3(deposit_grid(total)), dim3(256), 0, s, flow, keys_sorted, slots_sorted, colors, total);
}

void launch_deposit_blend(const DepositParams &p, uint32_t total, hipStream_t s)
{
    launch_deposit_gather_colors(p.colors_sorted, p.colors, p.slots_sorted, total, s);
    hipLaunchKernelGGL(deposit_blend_kernel, dim3(deposit_grid(total)), dim3(256), 0, s, p, total);
}

}  // namespace th
