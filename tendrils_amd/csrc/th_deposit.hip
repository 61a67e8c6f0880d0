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
#include "th_raster.hpp"
#include <cstdlib>

namespace th {
namespace {


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
            dep_setup(p, col, p.row0 + row, L, (size_t)row * p.W + col);
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
    __shared__ float polygons[48 * 256];                  // (the clipped polygon, indexed at run time: LDS, not scratch memory)
    LdsWords<256> words{polygons + threadIdx.x};
    dep_list_work(p, kListSlow, [&](bool have, uint32_t t, uint32_t seg) {
        LineRecord r{};
        if (have) {
            const uint32_t row = t / p.W, col = t - row * p.W;
            DepositLine L;
            dep_setup(p, col, p.row0 + row, L, (size_t)row * p.W + col);
            dep_raster_line(p, L, [&](int x, int y) { rec_add(r, x, y); }, words);
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
        dep_setup(p, col, p.row0 + row, L, (size_t)row * p.W + col);           // vertices and snapped endpoints
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
    __shared__ float polygons[48 * 256];
    LdsWords<256> words{polygons + threadIdx.x};
    dep_list_work(p, kListLong, [&](bool have, uint32_t t, uint32_t) {
        if (!have) return;
        const uint32_t row = t / p.W, col = t - row * p.W;
        const uint32_t id = col * p.H + p.row0 + row;
        uint32_t at = p.offset[t];
        DepositLine L;
        dep_setup(p, col, p.row0 + row, L, (size_t)row * p.W + col);
        dep_raster_line(p, L, [&](int x, int y) { dep_put(p, L, id, at, x, y); ++at; }, words);
    });
}


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
TH_D void blend_banded_run(typename Target::Texel *plane, const BandKeys &keys, const float4 *colors, uint32_t stride, uint32_t i, uint32_t total,
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
            Target::apply(d, Target::source(colors[(size_t)best * stride]));
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
            Target::apply(d, Target::source(colors[(size_t)pos[best] * stride]));
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
            // the destination, a channel per lane (lane & 3; every group of four lanes does the same: nothing diverges)
            float rc = Target::channel(Target::from_lane(d, owner), lane & 3u);
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
                const float *mine = reinterpret_cast<const float *>(&staged[threadIdx.x & ~63u]);      // {x, y, z, w, da} side by side
                const float *comp = mine + (lane & 3u), *das = mine + 4;
                int q = 0;
                for (; q + 8 <= n; q += 8) {
                    float s8[8], d8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { s8[k] = comp[(q + k) * 5]; d8[k] = das[(q + k) * 5]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) Target::apply_channel(rc, s8[k], d8[k]);
                }
                for (; q < n; ++q) Target::apply_channel(rc, comp[q * 5], das[q * 5]);
                if (n < 64) break;
                c = cn; same = samen; id = idn; at0 += 64u;
            }
            const Texel rd = Target::from_channels(lane_float(rc, 0), lane_float(rc, 1), lane_float(rc, 2), lane_float(rc, 3));
            if (lane == (uint32_t)owner) { if (falls) banded = true; else plane[run_texel] = rd; }
        }
        if constexpr (Keys::kBands) if (banded) blend_banded_run<Target>(plane, keys, colors, stride, i, total, texel, too_many);
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

// Tendrils.copyBuffer (src/index.js:370-383, src/screen/copy.frag): a full-screen quad textured with a view buffer of the
// target's own shape - gl_FragCoord.xy / viewRes samples every texel at its centre - blended like everything else.
// (dst == src: every texel over itself.)
__global__ __launch_bounds__(256) void view_copy_kernel(uchar4 *view, const uchar4 *src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uchar4 s = src[i];
        uchar4 d = view[i];
        dep_blend_rgba8(d, make_float4((float)s.x / 255.0f, (float)s.y / 255.0f, (float)s.z / 255.0f, (float)s.w / 255.0f));
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
        const DepositVertex a = dep_fetch(p, i, 2u * m, row, t), b = dep_fetch(p, i, 2u * m + 1u, row, t);
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

// words of the slow / long / spanning line lists: counters, then kListKinds * kDepLists segments of *cap entries
size_t deposit_list_words(uint32_t W, uint32_t rows, uint32_t *cap)
{
    const uint32_t groups = ((W + 255u) >> 8) * rows;
    *cap = ((groups + kDepLists - 1u) / kDepLists) * 256u;
    return (size_t)kListKinds * kDepLists * kDepListStride + (size_t)kListKinds * kDepLists * *cap;
}
size_t deposit_list_counter_bytes() { return (size_t)kListKinds * kDepLists * kDepListStride * sizeof(uint32_t); }

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

void launch_deposit_gather_pairs(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(deposit_gather_pairs_kernel, dim3(deposit_grid(n)), dim3(256), 0, s, dst, src, index, n);
}

// ... both passes of a draw() in one merge: the fragments carry the flow pass's varying and the view pass's colour side by
// side (th_draw_emit); one gather of the pairs, then each target's blend over its own half - as launch_deposit_blend does
void launch_draw_blend64(float4 *flow, uchar4 *view, const unsigned long long *keys_sorted, const uint32_t *slots_sorted,
                         const float4 *colors, float4 *colors_sorted, uint32_t total, uint32_t *too_many, hipStream_t s)
{
    if (!total) return;
    launch_deposit_gather_pairs(colors_sorted, colors, slots_sorted, total, s);
    hipLaunchKernelGGL((deposit_blend_kernel<FlowTarget, BandKeys>), dim3(deposit_grid(total)), dim3(256), 0, s, flow, BandKeys{keys_sorted},
                       (const float4 *)colors_sorted, 2u, total, too_many);
    hipLaunchKernelGGL((deposit_blend_kernel<ViewTarget, BandKeys>), dim3(deposit_grid(total)), dim3(256), 0, s, view, BandKeys{keys_sorted},
                       (const float4 *)colors_sorted + 1, 2u, total, too_many);
}

// first fragment of every owner's part of keys parted by owner (bounds[r] = first key with owner >= r; bounds[world] = n)
__global__ void owner_bounds_kernel(const unsigned long long *keys, uint32_t n, uint32_t world, unsigned long long *bounds)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > world) return;
    const unsigned long long want = (unsigned long long)r << kOwnerShift;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (keys[mid] < want) lo = mid + 1u; else hi = mid; }
    bounds[r] = r == world ? n : lo;
}
void launch_owner_bounds(const unsigned long long *keys, uint32_t n, uint32_t world, unsigned long long *bounds, hipStream_t s)
{
    hipLaunchKernelGGL(owner_bounds_kernel, dim3(1), dim3(64), 0, s, keys, n, world, bounds);
}

// The small words of a sharded draw's exchange, made where they are needed instead of travelling through the host (every trip
// was a copy, a stream synchronize and a copy back - th_shard.hip):
//   exchange_send_counts  what this rank holds for every owner = the differences of its owners' bounds (status bits: none - a
//                         rank with something to report uploads its words itself)
//   exchange_recv_base    where every source's part starts in the received arrays = the running sum of the received counts
//   exchange_word         one word from the kernel arguments (the status the ranks agree on)
__global__ void exchange_send_counts_kernel(const unsigned long long *bounds, uint32_t world, unsigned long long *send)
{
    if (threadIdx.x < world) send[threadIdx.x] = bounds[threadIdx.x + 1u] - bounds[threadIdx.x];
}
__global__ void exchange_recv_base_kernel(const unsigned long long *recv, uint32_t world, unsigned long long *base)
{
    if (threadIdx.x == 0u) {
        unsigned long long run = 0;
        for (uint32_t r = 0; r < world; ++r) { base[r] = run; run += recv[r] & 0xffffffffull; }
    }
}
__global__ void exchange_word_kernel(uint32_t *dst, uint32_t word) { if (threadIdx.x == 0u) *dst = word; }
void launch_exchange_send_counts(const unsigned long long *bounds, uint32_t world, unsigned long long *send, hipStream_t s)
{
    hipLaunchKernelGGL(exchange_send_counts_kernel, dim3(1), dim3(64), 0, s, bounds, world, send);
}
void launch_exchange_recv_base(const unsigned long long *recv, uint32_t world, unsigned long long *base, hipStream_t s)
{
    hipLaunchKernelGGL(exchange_recv_base_kernel, dim3(1), dim3(64), 0, s, recv, world, base);
}
void launch_exchange_word(uint32_t *dst, uint32_t word, hipStream_t s)
{
    hipLaunchKernelGGL(exchange_word_kernel, dim3(1), dim3(64), 0, s, dst, word);
}

// ... and the view pass of a row-band shard: the same merge into the RGBA8 view buffer
void launch_view_blend64(uchar4 *view, const unsigned long long *keys_sorted, const uint32_t *slots_sorted,
                         const float4 *colors, float4 *colors_sorted, uint32_t total, uint32_t *too_many, hipStream_t s)
{
    if (!total) return;
    launch_deposit_gather_colors(colors_sorted, colors, slots_sorted, total, s);
    hipLaunchKernelGGL((deposit_blend_kernel<ViewTarget, BandKeys>), dim3(deposit_grid(total)), dim3(256), 0, s, view, BandKeys{keys_sorted},
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

void launch_view_copy(uchar4 *view, const uchar4 *src, size_t texels, hipStream_t s)
{
    if (texels) hipLaunchKernelGGL(view_copy_kernel, dim3(deposit_grid((uint32_t)(texels < 0xffffffffull ? texels : 0xffffffffull))), dim3(256), 0, s, view, src, texels);
}

void launch_view_fill(uchar4 *view, size_t texels, float4 color, hipStream_t s)
{
    if (texels) hipLaunchKernelGGL(view_fill_kernel, dim3(deposit_grid((uint32_t)(texels < 0xffffffffull ? texels : 0xffffffffull))), dim3(256), 0, s, view, texels, color);
}

}  // namespace th
