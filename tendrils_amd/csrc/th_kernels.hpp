// th_kernels.hpp - launch parameter blocks shared by th_kernels.hip (device)
// and th_api.hip (host).
#pragma once
#include <string>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tendrils_hip.h"

namespace th {

constexpr uint32_t kMaxFusedSteps = 32;

// Tile-sorted slot order (th_kernels.hip "Tile-sorted slot order")
constexpr uint32_t kTileChunk = 4096;        // slots per workgroup (16 per thread), all of one tile
constexpr uint32_t kMaxTileBins = 8192;      // sort classes (2 per tile + 2): larger flow fields keep the texel order
constexpr uint32_t kSortReplicas = 64;       // copies of the global sort counters (power of two)
constexpr uint32_t kBinSlotsLog2 = 6, kBinSlots = 1u << kBinSlotsLog2;   // LDS table of the tiles one workgroup meets

struct TileGeom {                // how a particle maps to its sort class: the tap arithmetic of the integrator + whether its line can draw
    float view_x, view_y, half_fw, half_fh, fwm1, fhm1;
    uint32_t tiles_x, ntiles;    // classes 2*tile + idle; tile `ntiles` = particles that tap nothing (inert, NaN / infinite position)
    uint32_t width, log2w, pow2w, row0;      // particle id -> global row of the state texture
    const uint32_t *row_draws;   // bit per global row: draw() can make a line of the particles of this row
};

struct TileChunk { uint32_t start, count, tile, pad; };
struct ChunkRecord { uint32_t key[kBinSlots], count[kBinSlots]; };   // a chunk's table of (tile, particles) of its next positions

// Kernel argument block of the integrator.  Passed by value: it lands in the
// kernarg segment and every field is wave-uniform (SGPRs).
struct LogicParams {
    const float4 *in;        // ring buffers[1]  (this context's rows)
    float4 *out;             // ring buffers[0] or an explicit target
    const float4 *flow;      // RGBA32F flow texture, fw x fh
    const float2 *flow_dec;  // flow decoded for this step's time: xy * max(0, 1-(time-z)*decay)
    const float *flow3;      // (fused passes) the flow texels' x, y, z alone, 12 B apart: three quarters of the field's footprint in the
                             // XCDs' L2s - the band one XCD taps (a sorted eighth of the slots) then fits its 4 MB; nullptr: p.flow
    const float4 *targets;   // RGBA32F targets texture (local rows)
    const float4 *lut;       // noise gradient table (kLutSize float4), preceded in memory by the hash tables (hash_table_vectors() float4)
    uint32_t count;          // texels held by this context = width * local rows
    uint32_t width;
    uint32_t log2w;          // valid when pow2 != 0
    uint32_t row0;
    float wf, hf;            // dataRes (global), as floats
    float inv_w, inv_h, inv_wh;   // exact reciprocals, valid when pow2 != 0
    int32_t fw, fh;
    float fwf, fhf, fwm1, fhm1, half_fw, half_fh;
    th_logic_uniforms u;
    float s2_cap;            // largest s2 with sqrt_rn(s2) <= speedLimit (see th_api.hip)
    float pos_bound;         // |pos| below this keeps the noise coordinates inside kNoiseDomain
    const uint32_t *perm;    // sorted slot order of `in`: slot -> particle id (nullptr = identity, texel order)
    // logic_sorted_kernel
    TileGeom geom;
    const TileChunk *chunks; // chunk table of the input order
    const uint32_t *nchunks;
    uint32_t *cursor;        // SCATTER: rank cursors of the new order (tile_scan_kernel)
    uint32_t *perm_out;      // SCATTER: particle ids of the new order
    float4 *in_moved;        // SCATTER: (optional) receives the INPUT state at the new slots as well
    uint32_t *hist;          // COUNT: histogram of the output positions' tiles
    ChunkRecord *records;    // COUNT: written per chunk; SCATTER with use_records: read per chunk
    uint32_t use_records;
    const float *time_dev;   // graph replays: `time` is read from here instead of u.time (nullptr = u.time)
    // (single steps of a frame loop) a byte per 64 slots: some line of them - from the slot's input position to its output
    // position - may touch the view; draw() skips the blocks of 256 slots whose four bytes are zero (th_bins.hip).  A line is
    // hidden for sure when both its ends lie beyond the same edge of the view by more than the margin the bounds carry.
    uint8_t *seen;
    float seen_xlo, seen_xhi, seen_ylo, seen_yhi;
    // fused multi-step launches (logic_fused_kernel): nsteps consecutive steps per particle in one pass
    float4 *out_prev;        // receives state nsteps-1 (p.out receives state nsteps); may alias p.in
    uint32_t nsteps;
    struct StatsPartial *stats_part;   // (optional) per workgroup: the statistics of state nsteps, taken while it is in registers
    float times[kMaxFusedSteps];   // `time` of each fused step
};

struct TileSortParams {
    const float4 *state;         // the state being sorted, in any slot order
    const uint32_t *perm_in;     // that order (nullptr = texel order)
    uint32_t count;
    TileGeom g;
    uint32_t *hist;              // [kSortReplicas][kMaxTileBins], zero on entry of tile_hist, cleared by tile_scan
    uint32_t *cursor;            // [kSortReplicas][kMaxTileBins] next free slot of every bin, per copy
    uint32_t *totals, *starts;   // [kMaxTileBins] each: a bin's count over all copies / its first slot (between the scan's launches)
    TileChunk *chunks;           // chunk table of the NEW order
    uint32_t *nchunks;
    float4 *state_out;           // tile_scatter only
    uint32_t *perm_out;
    ChunkRecord *block_records;  // per 4096-slot block of the input: written by tile_hist, used by tile_scatter (may be null)
    uint32_t packed;             // the state is the packed 8-byte form (TH_STATE_F16): state / state_out point at uint2 texels
};

struct OpticalFlowParams {
    const uchar4 *view, *last;
    float4 *flow;
    int32_t fr_w, fr_h;      // frame size
    int32_t out_w, out_h;    // flow texture size
    float grad_x, grad_y;    // d(uv)/d(pixel) of the full-screen pass: fl(2/out_w), fl(2/out_h)
    th_optical_flow_uniforms u;
};

struct SpawnBallParams {
    float4 *out;
    uint32_t count, width, row0;
    th_spawn_ball_uniforms u;
};

struct SpawnSampleParams {
    const float4 *particles;
    float4 *out;
    const float4 *data;
    uint32_t count, width, row0;
    float wf, hf;
    int32_t dw, dh;
    th_spawn_sample_uniforms u;
    unsigned long long *accepted;   // += particles that took a candidate
};

// flow deposit (th_deposit.hip): draw()'s flow pass
// key of a fragment in the sharded deposit: 24 texel bits above the 32-bit stream index, the owner rank above them
constexpr int kOwnerShift = 56;
constexpr uint32_t kTexelMask = 0xffffffu;

constexpr float kMaxLineWidth = 64.0f;      // th_line_width_range's upper end (TH_MAX_LINE_WIDTH): dep_param's 32-bit arithmetic holds up to it

// Shapes whose vertex lookup lands on ANOTHER particle (th_order.hip: line_rows - a few rows / columns of W >= 8192, of heights
// such as 100, 1080, 3000): where that particle lies when the state is held in a slot order.  A source texel lies in a source
// ROW (the row the lookup of some other row lands on) or in a source COLUMN; `slot` holds their slots - row_index[row] * W +
// col, then nrows * W + col_index[col] * rows + (row - row0) - filled once per slot order (bins_block_flags_kernel).
struct LineSources {
    const uint16_t *row_index;   // per LOCAL row: its place among the source rows, 0xffff: not one
    const uint16_t *col_index;   // per column
    const uint32_t *slot;        // nullptr: texel order (the slot of a texel is its index), or every vertex reads its line's own particle
    uint32_t nrows, ncols;
};

struct DepositParams {
    const float4 *cur, *prev;    // buffers[0], buffers[1] in texel order (a packed ring: uint2 texels behind the same pointers)
    float4 *flow;
    uint32_t W, H;               // particle texture shape (H = the WHOLE texture's height)
    uint32_t row0, rows;         // the rows held by cur/prev (a row-band shard; row0 = 0, rows = H otherwise)
    uint32_t *oob;               // set to 1 when a vertex lookup leaves the band (and its halo rows)
    const float4 *halo_lo, *halo_hi;   // row row0-1 / row0+rows of the neighbouring bands: W texels of `cur`, then W of `prev`
    int32_t fw, fh;              // flow texture shape
    float view_x, view_y, time, speed_limit;
    float line_half;             // half the width the lines are drawn with, in texels (th_line_width after the clamp to its range; 0.5 = width 1)
    // view pass (mode 1): the vertex colours of src/render/index.vert:58-100, blended into the RGBA8 view buffer
    int32_t mode;                // 0 = flow pass (varying = vel, time, alpha), 1 = view pass, 2 = both (varyings in pairs)
    float flow_decay, speed_alpha, colormap_alpha, sin_term;
    float base_color[4], flow_color[4];
    const float4 *colormap;      // cw x ch RGBA32F, NEAREST / CLAMP (nullptr = the 1x1 zero texture)
    int32_t cw, ch;
    uchar4 *view;
    double inv_x, inv_y;         // 1/(max(W,2)-1), 1/(max(2H,2)-1): Particles.generateLUT (src/particles.js:171-190)
    uint32_t *count, *offset;    // per line (row-major, as the threads walk): fragments, first slot in the stream-ordered fragment array
    uint4 *record;               // per line (two uint4): the texels (x | y << 16) of a line of <= 8 fragments
    uint32_t *list_n, *lists;    // slow / long line lists (th_deposit.hip: kDepLists segments of list_cap entries each)
    uint32_t list_cap;
    uint32_t *keys, *slots;      // per fragment (stream order): flow texel, own slot
    uint32_t *keys_sorted, *slots_sorted;   // the same after the stable sort by texel
    float4 *colors;              // per fragment (stream order): interpolated varying
    float4 *colors_sorted;       // ... gathered into the sorted order
    unsigned long long *keys64;  // sharded form, per fragment: owner << kOwnerShift | texel << 32 | global stream index
    uint32_t owners, owner_chunk;    // ranks that own flow texels (contiguous ranges of owner_chunk texels)
    const uint32_t *row_draws;   // bit per global row of the state texture: its lines can draw at all (th_api.hip: line_rows)
    // binned pipeline (th_bins.hip): lines are walked by SLOT - cur / prev in the ring's slot order, perm[slot] = particle
    // id (nullptr = texel order); fragments go straight from the rasteriser into the 16 x 16-texel bin of the target they
    // fall into.  A bin is kBinReplicas lists of pages of kBinPage places: page 0 of list r of bin b is page
    // b * kBinReplicas + r, further pages come from a pool.
    const uint32_t *perm;
    const uint32_t *draw_blocks;                   // the blocks of 256 slots that hold a line that can draw at all, in rising order (launch_bins_block_list)
    const uint32_t *block_seen;                    // per block of 256 slots: the four bytes the step left (LogicParams::seen); nullptr: every listed block is walked
    uint32_t draw_nblocks;
    uint32_t bins_x, nbins;
    uint32_t *bin_cursor, bin_stride;              // per list (r * bin_stride + the bin's transposed index, th_bins.hip: list_cursor): places handed out so far (virtual indices inside the list)
    uint32_t *page_table;                          // per list x max_pages: page id of the list's n-th page, n >= 1 (0: not handed out yet)
    uint32_t max_pages;                            // pages one list can grow to in this pass (the host widens the table when a bin outgrew it)
    uint32_t pool_pages;                           // pages in the pool: ids nbins * kBinReplicas .. + pool_pages - 1
    uint32_t *totals;                              // device words (th_bins.hip: kTot*)
    uint32_t *totals_host, totals_seq;             // (host memory the device writes) the totals once the plan is done, then totals_seq behind them: the host's poll ends
    unsigned long long *frag_keys;                 // per place, chunk-major: (y << 12 | x) << 32 | stream index of the line; ~0 = empty
    // ... bins of more places than one workgroup orders in LDS (crowd_*_kernel)
    uint32_t *large_bins, *large_key0;             // the large bins (in any order); first regrouped key of each (+ 1)
    uint32_t nlarge;
    uint32_t *crowd_count, *crowd_start, *crowd_cursor;   // per large bin: fragments per texel (256), first of every texel (257), fill cursors (256)
    unsigned long long *crowd_keys;                // per fragment of a large bin, grouped by texel: stream index << 32 | place of its varying
    uint32_t *crowd_sorted;                        // ... and the places alone, every texel's run in blend order (crowd_sort_kernel, long_sort_kernel, giant_sort_kernel)
    uint32_t *crowd_long, *crowd_giant;            // texels of large bins whose runs one wave does not order (large bin << 8 | texel): up to kGiantRun fragments / more
    unsigned long long *crowd_parted;              // the giants' keys parted by the leading bits of their stream indices (same positions as crowd_keys)
    uint32_t *crowd_giant_win;                     // per entry of crowd_giant: first window, windows (first = ~0: left to crowd_blend_kernel)
    uint32_t *crowd_windows, crowd_windows_cap;    // per window: entry of crowd_giant, first key inside the run, keys
    // (round 6's fields at the END: in front, they moved every other field's place in the kernel arguments, and the emit - 106
    // SGPRs, some of them spilled into VGPR lanes - came out of the compiler scheduled otherwise and 4 % slower, instruction for
    // instruction the same kernel: profiles/r6_d_blend_experiments.txt)
    uint32_t packed;             // cur / prev hold the packed 8-byte form (TH_STATE_F16): read through dep_state (th_raster.hpp)
    LineSources src;
};

// row-band shards drawing with the binned pipeline (th_bins.hip "the bins travel to the ranks that own them")
struct OwnerParams {
    uint32_t world, rank;
    uint32_t bin_lo[33];                 // owner r owns the bins [bin_lo[r], bin_lo[r + 1]): whole bin rows of the target
    // sender
    uint32_t *counts;                    // [nbins] places per bin
    unsigned long long *offsets;         // [nbins + 1] their exclusive scan: where each bin lies in the outgoing arrays
    unsigned long long *owner_bounds;    // [world + 1] offsets at the owners' first bins (and the total)
    unsigned long long *out_keys;
    float4 *out_colors;
    // owner
    uint32_t nb;                         // my bins
    uint32_t pool_used;                  // pool pages my own emitting pass took
    const uint32_t *table;               // [world][nb] what every source holds for each of my bins
    unsigned long long *src_prefix;      // [world][nb] where each of my bins starts inside that source's part
    uint32_t *bin_total;                 // [nb]
    uint32_t *bin_page;                  // [nb] first pool page of the bin's lists
    const unsigned long long *recv_base; // [world] first fragment of every source's part in the received arrays
    const unsigned long long *in_keys;
    const float4 *in_colors;
};

struct TrianglePoly {           // a clipped, snapped, oriented triangle (th_deposit.hip)
    int32_t n;
    int32_t x[8], y[8];
};

struct StatsPartial {
    unsigned long long live, nan, capped;
    double sum_speed, max_speed;
};

// launchers (defined in th_kernels.hip)
void launch_logic(const LogicParams &p, int mode, bool noise, bool target, bool pow2, bool decoded,
                  bool generic, bool packed, hipStream_t stream);
void launch_logic_fused(const LogicParams &p, int mode, bool noise, bool target, bool pow2, bool packed, hipStream_t stream);
void launch_pack_state(void *dst, const float4 *src, uint32_t n, hipStream_t stream);      // f32 texels -> TH_STATE_F16
void launch_unpack_state(float4 *dst, const void *src, uint32_t n, hipStream_t stream);
void launch_flow_pack3(const float4 *flow, float *xyz, size_t n, hipStream_t stream);
void launch_flow_decode(const float4 *flow, float2 *dec, size_t n, float time, const float *time_dev, float decay,
                        hipStream_t stream);
void launch_logic_sorted(const LogicParams &p, int mode, bool noise, bool target, bool pow2, bool in_tiled, bool scatter,
                         bool count, uint32_t max_chunks, hipStream_t stream);
void launch_tile_hist(const TileSortParams &b, hipStream_t stream);
void launch_tile_scan(const TileSortParams &b, hipStream_t stream);
void launch_tile_scatter(const TileSortParams &b, hipStream_t stream);
void launch_unpermute_state(float4 *dst, const float4 *src, const uint32_t *perm, uint32_t n, bool packed, hipStream_t stream);   // dst[perm[s]] = src[s]
void launch_permute_state(float4 *dst, const float4 *src, const uint32_t *perm, uint32_t n, bool packed, hipStream_t stream);     // dst[s] = src[perm[s]]
void launch_fill(float4 *dst, float4 value, size_t n, hipStream_t stream);
void launch_hash_tables(float4 *block, hipStream_t stream);      // the hash tables in front of the gradient table (th_kernels.hip)
int hash_table_vectors();
void launch_finite_check(const float4 *src, size_t n, unsigned int *flag, hipStream_t stream);
// the fold of `nparts` per-workgroup partials a fused launch left (LogicParams::stats_part; fused_stats_parts() of them)
void launch_stats_fold(const StatsPartial *parts, uint32_t nparts, StatsPartial *scratch, size_t n, const unsigned long long *respawned,
                       th_counters *out, hipStream_t stream);
uint32_t fused_stats_parts(uint32_t count, bool sorted);
void launch_stats(const float4 *state, size_t n, float speed_limit, StatsPartial *partials, const unsigned long long *respawned,
                  th_counters *out, hipStream_t stream);
void launch_counter_add(unsigned long long *counter, unsigned long long n, hipStream_t s);
void launch_optical_flow(const OpticalFlowParams &p, hipStream_t stream);
uint32_t deposit_scan_words(uint32_t W, uint32_t rows);
size_t deposit_list_words(uint32_t W, uint32_t rows, uint32_t *cap);
void launch_deposit_count(const DepositParams &p, hipStream_t stream);
void launch_deposit_scan(const DepositParams &p, uint32_t *scratch, uint32_t *total, hipStream_t stream);
void launch_deposit_scatter(const DepositParams &p, hipStream_t stream);
int deposit_key_bits(const DepositParams &p);
void launch_export_mark(const DepositParams &p, hipStream_t stream);
void launch_export_write(const DepositParams &p, float *out, hipStream_t stream);
void launch_triangles(const float *positions, int ntri, float view_x, float view_y, float4 color, TrianglePoly *polys,
                      float4 *img, int w, int h, hipStream_t stream);
void launch_deposit_blend(const DepositParams &p, uint32_t total, hipStream_t stream);
void launch_view_fill(uchar4 *view, size_t texels, float4 color, hipStream_t stream);
void launch_view_copy(uchar4 *view, const uchar4 *src, size_t texels, hipStream_t stream);
// binned pipeline (th_bins.hip)
constexpr int kBinShift = 4;                       // 16 x 16 texel bins
constexpr uint32_t kBinCap = 4096;                 // places of a bin that one workgroup orders in LDS
constexpr uint32_t kBinReplicas = 16;              // lists per bin (th_bins.hip)
constexpr uint32_t kBinPage = 256;                 // places per page
constexpr uint32_t kBinFirstPages = 128;           // pages one list can grow to at first (half a million places per bin) ...
constexpr uint32_t kBinPagesLimit = 4096;          // ... and after the table has been widened as far as it goes (16 M places per bin)
constexpr int32_t kBinsMaxExtent = 4096;           // fragment keys hold 12 bits per texel coordinate
// totals[]: device words of one pass
enum { kTotFragments = 0, kTotOob = 1, kTotFlags = 2, kTotLarge = 3, kTotGiant = 4, kTotLong = 5, kTotPool = 6, kTotCrowdKeys = 7, kTotWindows = 8, kTotWords = 12 };
enum { kBinsPoolExhausted = 1u, kBinsBoundBroken = 2u, kBinsBinFull = 4u, kBinsWaitBroken = 8u };      // (8: a page nobody published - page_of)
void launch_bins_block_list(const DepositParams &p, uint8_t *flags, uint32_t *list, uint32_t *count, uint32_t *src_slots, hipStream_t stream);   // list: a word per block of 256 slots; src_slots: LineSources::slot to fill (or nullptr)
void launch_bins_edge_rows(const DepositParams &p, float4 *rows, hipStream_t stream);   // a band's first and last row of cur and of prev, f32, in texel order: [first cur | first prev | last cur | last prev]
void launch_bins_fused(const DepositParams &p, hipStream_t stream);                   // rasterise + emit every line's fragments into its bins; then the large-bin plan
void launch_bins_owner_counts(const DepositParams &p, const OwnerParams &o, hipStream_t stream);
void launch_bins_owner_extract(const DepositParams &p, const OwnerParams &o, hipStream_t stream);
void launch_bins_owner_insert(const DepositParams &p, const OwnerParams &o, hipStream_t stream);
void launch_bins_regroup(const DepositParams &p, hipStream_t stream);                 // the large bins' fragments regrouped by texel
void launch_bins_part_giants(const DepositParams &p, hipStream_t stream);             // the giants' keys parted by stream index (behind the regroup, ahead of launch_bins_blend_giants)
void launch_bins_blend_giants(const DepositParams &p, hipStream_t stream);            // their runs of more than kGiantRun fragments, a workgroup each (the longest chains of a draw)
void launch_bins_sort_long(const DepositParams &p, hipStream_t stream);               // their runs a wave orders on its own (kWaveRun + 1 .. kGiantRun fragments) ...
void launch_bins_walk_long(const DepositParams &p, hipStream_t stream);               // ... walked
void launch_bins_blend_crowd(const DepositParams &p, hipStream_t stream);             // their other runs, a wave each: order by stream index, blend
void launch_bins_blend(const DepositParams &p, hipStream_t stream);                   // the bins one workgroup orders: by (texel, stream index), blend (needs no host value)
size_t crowd_words_per_bin();
// How a context's ranks exchange bytes.  th_shard.hip does the path's arithmetic - which fragments go to which owner, where
// the bands and the owned texel ranges lie - and hands plain (pointer, count, offset) lists to one of two transports:
//   rccl      one process per GPU, RCCL over xGMI (th_comm.hip; librccl bound at run time) - the product's;
//   loopback  the ranks are contexts of ONE process, each driven by its own host thread, device-to-device copies and a
//             host-side rendezvous (th_loopback.hip) - so that the exchange logic runs with more ranks than a box has GPUs.
// Every function returns 0 or leaves comm_error().  All of them are collective: every rank of the communicator calls.
struct Transport {
    const char *name;
    int (*destroy)(void *comm);
    // th_counters in place: [0, 40) five u64 counts (sum), [40, 48) sum_speed (sum), [48, 56) max_speed (max)
    int (*allreduce_counters)(void *comm, void *counters_dev, hipStream_t stream);
    // every rank's bytes[r] bytes at `send` land at recv + offset[r] on every rank (parts may differ in size)
    int (*allgather_bytes)(void *comm, const void *send, void *recv, const size_t *bytes, const size_t *offset, int rank, int world, hipStream_t stream);
    // send_counts[r] elements of `elem` bytes from send + send_off[r] * elem to rank r, and likewise received
    int (*alltoallv)(void *comm, const void *send, const size_t *send_counts, const size_t *send_off, void *recv, const size_t *recv_counts,
                     const size_t *recv_off, size_t elem, int world, hipStream_t stream);
    // a device word per rank -> the largest of them on every rank, in place
    int (*allreduce_max_u32)(void *comm, uint32_t *word_dev, hipStream_t stream);
};
const char *comm_error();
int comm_fail(const std::string &why);               // records comm_error(), returns 1
int comm_available(int *version);
int comm_unique_id(void *out, size_t bytes);         // rccl: ncclGetUniqueId
int loopback_unique_id(void *out, size_t bytes);     // a fresh in-process world
bool loopback_id(const void *id_bytes);
// joins the communicator `id` names - an RCCL one or an in-process one - as `rank` of `world`
int comm_init(void **comm, const Transport **transport, const void *id_bytes, size_t bytes, int rank, int world);
int loopback_init(void **comm, const Transport **transport, const void *id_bytes, size_t bytes, int rank, int world);
// stable LSD radix sort of (key, u32 value) pairs by key bits [begin_bit, end_bit) (th_sort.hip): the passes ping-pong
// between the (a) and (b) buffers; returns 0 when the result is in (a), 1 when it is in (b)
constexpr uint32_t kRadixBits = 8;
size_t radix_sort_temp_bytes(uint32_t n, int begin_bit, int end_bit);
// iota: the values are the elements' positions 0..n-1 (vals_a need not be filled)
int launch_radix_sort_u32(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint32_t n, int begin_bit,
                          int end_bit, void *temp, bool iota, hipStream_t stream);
int launch_radix_sort_u64(unsigned long long *keys_a, uint32_t *vals_a, unsigned long long *keys_b, uint32_t *vals_b, uint32_t n,
                          int begin_bit, int end_bit, void *temp, bool iota, hipStream_t stream);
void launch_deposit_gather_colors(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n, hipStream_t stream);
void launch_deposit_blend64(float4 *flow, const unsigned long long *keys_sorted, const uint32_t *slots_sorted,
                            const float4 *colors, float4 *colors_sorted, uint32_t total, uint32_t *too_many, hipStream_t stream);
void launch_owner_bounds(const unsigned long long *keys, uint32_t n, uint32_t world, unsigned long long *bounds, hipStream_t stream);   // world <= 32
void launch_exchange_send_counts(const unsigned long long *bounds, uint32_t world, unsigned long long *send, hipStream_t stream);
void launch_exchange_recv_base(const unsigned long long *recv, uint32_t world, unsigned long long *base, hipStream_t stream);
void launch_exchange_word(uint32_t *dst, uint32_t word, hipStream_t stream);
void launch_view_blend64(uchar4 *view, const unsigned long long *keys_sorted, const uint32_t *slots_sorted,
                         const float4 *colors, float4 *colors_sorted, uint32_t total, uint32_t *too_many, hipStream_t stream);
void launch_deposit_gather_pairs(float4 *dst, const float4 *src, const uint32_t *index, uint32_t n, hipStream_t stream);
void launch_draw_blend64(float4 *flow, uchar4 *view, const unsigned long long *keys_sorted, const uint32_t *slots_sorted,
                         const float4 *colors, float4 *colors_sorted, uint32_t total, uint32_t *too_many, hipStream_t stream);   // both targets from pairs of varyings
void launch_spawn_ball(const SpawnBallParams &p, hipStream_t stream);
void launch_spawn_sample(const SpawnSampleParams &p, hipStream_t stream);
void launch_spawn_direct(const SpawnSampleParams &p, hipStream_t stream);
constexpr int kStatsBlocks = 1024;

}  // namespace th
