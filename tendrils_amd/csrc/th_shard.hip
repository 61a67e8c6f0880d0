// th_shard.hip - row-band shards: the draw() exchange (emit by owner, merge, th_draw_sharded), the job's
// communicator, the particle-texture gather of the spawners, the counter all-reduce.
#include "th_ctx.hpp"

using namespace thi;

extern "C" {

th_status th_deposit_set_owners(th_context *c, int32_t world)
{
    TH_REQUIRE(c, "null context");
    // (the owner's merge walks up to 32 source bands per texel - th_deposit.hip: kMaxBands: more ranks than that could only be
    // refused after the blend had begun)
    TH_REQUIRE(world >= 1 && world <= 32, "owner count %d outside [1, 32]", world);
    c->dep_owners = (uint32_t)world;
    return TH_OK;
}

// the (counted) fragments of this band's lines, keyed (owner, texel, global stream index) and parted by owner
static th_status emit_parted(th_context *c, th::DepositParams &p, uint32_t total, uint64_t *count, void **keys_dev, void **colors_dev)
{
    *count = total; *keys_dev = nullptr; *colors_dev = nullptr;
    if (total == 0) return TH_OK;
    const bool pairs = p.mode == 2;                  // th_draw_emit: two varyings per fragment, side by side
    if (th_status s = deposit_reserve(c, total, true, pairs)) return s;
    p.keys64 = c->dep_u64[0]; p.slots = c->dep_u32[1]; p.colors = c->dep_colors;
    p.owners = c->dep_owners;
    p.owner_chunk = (uint32_t)(((uint64_t)c->fw * c->fh + p.owners - 1u) / p.owners);
    th::launch_deposit_scatter(p, c->stream);
    *keys_dev = c->dep_u64[0]; *colors_dev = c->dep_colors;
    if (p.owners > 1u) {
        // the fragment array is in this band's stream order: ONE stable pass on the owner bits parts it by destination
        // (every part still in stream order); the owners sort by texel
        int owner_bits = 1;
        while ((1u << owner_bits) < p.owners) ++owner_bits;
        if (th_status s = deposit_temp(c, th::radix_sort_temp_bytes(total, th::kOwnerShift, th::kOwnerShift + owner_bits))) return s;
        const int in_b = th::launch_radix_sort_u64(c->dep_u64[0], c->dep_u32[1], c->dep_u64[1], c->dep_u32[3], total, th::kOwnerShift,
                                                   th::kOwnerShift + owner_bits, c->dep_temp, true, c->stream);
        if (pairs) th::launch_deposit_gather_pairs(c->dep_colors_sorted, c->dep_colors, in_b ? c->dep_u32[3] : c->dep_u32[1], total, c->stream);
        else th::launch_deposit_gather_colors(c->dep_colors_sorted, c->dep_colors, in_b ? c->dep_u32[3] : c->dep_u32[1], total, c->stream);
        *keys_dev = in_b ? c->dep_u64[1] : c->dep_u64[0]; *colors_dev = c->dep_colors_sorted;
    }
    TH_HIP(hipGetLastError());
    TH_HIP(hipStreamSynchronize(c->stream));               // the caller hands the buffers to a collective on its own stream
    return TH_OK;
}

th_status th_deposit_emit(th_context *c, const th_deposit_uniforms *u, uint64_t *count, void **keys_dev, void **colors_dev)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(count && keys_dev && colors_dev, "null outputs");
    TH_REQUIRE((uint64_t)c->fw * c->fh <= (uint64_t)th::kTexelMask + 1u, "the sharded deposit keys hold 24 texel bits: flow %dx%d is too large", c->fw, c->fh);
    th::DepositParams p;
    uint32_t total = 0;
    if (th_status s = deposit_count(c, u, p, &total)) return s;
    return emit_parted(c, p, total, count, keys_dev, colors_dev);
}

// the view pass of a row-band shard: the same lines with the render shader's colours
th_status th_view_emit(th_context *c, const th_render_uniforms *u, uint64_t *count, void **keys_dev, void **colors_dev)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(count && keys_dev && colors_dev, "null outputs");
    TH_REQUIRE((uint64_t)c->fw * c->fh <= (uint64_t)th::kTexelMask + 1u, "the sharded deposit keys hold 24 texel bits: view %dx%d is too large", c->fw, c->fh);
    th::DepositParams p;
    if (th_status s = view_params(c, u, p)) return s;
    th::launch_deposit_count(p, c->stream);
    uint32_t total = 0;
    if (th_status s = deposit_scan_total(c, p, &total)) return s;
    return emit_parted(c, p, total, count, keys_dev, colors_dev);
}

// both passes of a row-band shard's draw() in one: every fragment with the flow pass's varying and the view pass's colour
// side by side (32 bytes), rasterised, parted and - by the host or th_draw_sharded - exchanged once
th_status th_draw_emit(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *count, void **keys_dev, void **colors_dev)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(du && ru && count && keys_dev && colors_dev, "null argument");
    TH_REQUIRE((uint64_t)c->fw * c->fh <= (uint64_t)th::kTexelMask + 1u, "the sharded deposit keys hold 24 texel bits: target %dx%d is too large", c->fw, c->fh);
    TH_REQUIRE(memcmp(du->viewSize, ru->viewSize, sizeof du->viewSize) == 0 && memcmp(&du->time, &ru->time, sizeof du->time) == 0 &&
               memcmp(&du->speedLimit, &ru->speedLimit, sizeof du->speedLimit) == 0,
               "the two passes of one draw share viewSize, time and speedLimit");
    TH_REQUIRE(drawn_line_width(c, TH_PASS_FLOW) == drawn_line_width(c, TH_PASS_VIEW),
               "the two passes draw their lines %g and %g wide: th_deposit_emit and th_view_emit rasterise them apart",
               (double)drawn_line_width(c, TH_PASS_FLOW), (double)drawn_line_width(c, TH_PASS_VIEW));
    th::DepositParams p;
    if (th_status s = deposit_prepare(c, du, p)) return s;
    p.mode = 2;
    view_fields(c, ru, p);
    th::launch_deposit_count(p, c->stream);
    uint32_t total = 0;
    if (th_status s = deposit_scan_total(c, p, &total)) return s;
    return emit_parted(c, p, total, count, keys_dev, colors_dev);
}

th_status th_deposit_set_halo(th_context *c, const void *lo_dev, const void *hi_dev)
{
    TH_REQUIRE(c, "null context");
    c->halo_lo = static_cast<const float4 *>(lo_dev);
    c->halo_hi = static_cast<const float4 *>(hi_dev);
    return TH_OK;
}

static th_status merge_reserve(th_context *c, uint32_t total, int target);

// target: 0 = the flow texture, 1 = the view buffer, 2 = both (the fragments carry pairs of varyings: th_draw_emit)
static th_status merge_parted(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count, int target)
{
    if (count == 0) return TH_OK;
    TH_REQUIRE(keys_dev && colors_dev && count < (1ull << 31), "bad fragment buffers");
    const uint32_t total = (uint32_t)count;
    const bool into_view = target == 1;
    if (th_status s = merge_reserve(c, total, target)) return s;
    // what arrives is one part per source band, every part in that band's stream order: a stable sort by texel (the
    // owner bits above and the stream index below are left alone), then the blend merges the bands inside each texel
    const int bits = 32 + deposit_texel_bits(c);
    TH_HIP(hipMemsetAsync(c->dep_total, 0, 8 * sizeof(uint32_t), c->stream));
    // (the sort ping-pongs between its two buffer pairs: the caller's keys are copied, not sorted in place)
    TH_HIP(hipMemcpyAsync(c->mrg_keys, keys_dev, (size_t)total * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    const int in_b = th::launch_radix_sort_u64(c->mrg_keys, c->mrg_vals[0], c->mrg_keys2, c->mrg_vals[1], total, 32, bits, c->dep_temp, true, c->stream);
    if (target == 2)
        th::launch_draw_blend64(c->flow, c->view, in_b ? c->mrg_keys2 : c->mrg_keys, in_b ? c->mrg_vals[1] : c->mrg_vals[0],
                                static_cast<const float4 *>(colors_dev), c->mrg_colors, total, c->dep_total, c->stream);
    else if (into_view)
        th::launch_view_blend64(c->view, in_b ? c->mrg_keys2 : c->mrg_keys, in_b ? c->mrg_vals[1] : c->mrg_vals[0],
                                static_cast<const float4 *>(colors_dev), c->mrg_colors, total, c->dep_total, c->stream);
    else
        th::launch_deposit_blend64(c->flow, in_b ? c->mrg_keys2 : c->mrg_keys, in_b ? c->mrg_vals[1] : c->mrg_vals[0],
                                   static_cast<const float4 *>(colors_dev), c->mrg_colors, total, c->dep_total, c->stream);
    TH_HIP(hipGetLastError());
    uint32_t too_many = 0;
    if (th_status s = read_back(c, &too_many, c->dep_total, sizeof too_many)) return s;        // (a sync: the input buffers may be reused by the caller now)
    if (too_many) return fail(TH_ERR_UNSUPPORTED, "a texel received fragments of more than 32 source bands");
    return TH_OK;
}

th_status th_deposit_merge(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count)
{
    if (th_status s = use(c)) return s;
    return merge_parted(c, keys_dev, colors_dev, count, 0);
}

th_status th_view_merge(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    return merge_parted(c, keys_dev, colors_dev, count, 1);
}

th_status th_draw_merge(th_context *c, const void *keys_dev, const void *colors_dev, uint64_t count)
{
    if (th_status s = use(c, true)) return s;
    if (th_status s = view_storage(c)) return s;
    return merge_parted(c, keys_dev, colors_dev, count, 2);
}

// ---- draw() of a row-band shard, the exchange issued by the library over its own communicator ---------------------------------
// A rank that fails on its own (a line that needs a halo row nobody supplied, an allocation) must not leave the others
// waiting inside a collective: before every exchange whose size or success depends on something rank-local, the ranks
// agree on a status word, and all of them leave together.
constexpr unsigned long long kPeerFailed = 1ull << 62;       // (a count travels below bit 31)
constexpr unsigned long long kPeerRetries = 1ull << 61;      // ... a rank whose binned pass gave up: everybody takes the stream-ordered pass

// (TH_TESTING builds: TH_OPT_INJECT_FAILURE) this rank fails at `stage` once; a release build has no such switch
#ifdef TH_TESTING
static th_status injected(th_context *c, int stage)
{
    if (c->opt.inject_failure != stage) return TH_OK;
    c->opt.inject_failure = 0;
    return fail(TH_ERR_UNSUPPORTED, "injected failure at stage %d of the sharded draw (TH_OPT_INJECT_FAILURE)", stage);
}
static bool injected_give_up(th_context *c)       // value 4: this rank's binned pass gives up (everybody takes the stream-ordered pass)
{
    if (c->opt.inject_failure != 4) return false;
    c->opt.inject_failure = 0;
    return true;
}
#else
static th_status injected(th_context *, int) { return TH_OK; }
static bool injected_give_up(th_context *) { return false; }
#endif

static th_status peer_failure(th_context *c, int who, const char *stage)
{
    return fail(TH_ERR_UNSUPPORTED, "sharded draw: rank %d failed while %s (this rank, %d, had no error of its own); nothing was blended", who, stage, c->comm_rank);
}

// every rank hands in its status; all of them return TH_OK, or none does (a failing rank returns its own error)
static th_status agree_status(th_context *c, th_status mine, const char *stage)
{
    if (c->comm_world <= 1) return mine;
    const std::string why = mine != TH_OK ? last_error() : std::string();
    // the lowest failing rank wins the maximum; a failure (2) outranks a binned pass that only wants the stream-ordered pass (1)
    const uint32_t word = mine != TH_OK ? (((uint32_t)(c->comm_world - c->comm_rank)) | (mine == kRetryInStreamOrder ? 0x10000u : 0x20000u)) : 0u;
    th::launch_exchange_word(c->d_status, word, c->stream);         // (from the kernel arguments: no host buffer to outlive, no sync)
    const hipError_t e = hipGetLastError();
    if (c->transport->allreduce_max_u32(c->comm, c->d_status, c->stream)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    uint32_t worst = 0;
    if (th_status s = read_back(c, &worst, c->d_status, sizeof worst)) return s;
    TH_HIP(e);
    if ((worst >> 16) == 1u) return kRetryInStreamOrder;              // (every rank returns this together)
    if (mine != TH_OK) { last_error() = why; return mine; }
    if (worst) return peer_failure(c, c->comm_world - (int)(worst & 0xffffu), stage);
    return TH_OK;
}

// scratch of the owner's merge for `total` received fragments (grow-only); target 2: room for two varyings per fragment
static th_status merge_reserve(th_context *c, uint32_t total, int target)
{
    if (target == 2 && !c->mrg_pairs) {
        (void)hipFree(c->mrg_colors); c->mrg_colors = nullptr;
        if (c->mrg_capacity) TH_HIP(hipMalloc((void **)&c->mrg_colors, 2 * c->mrg_capacity * sizeof(float4)));
        c->mrg_pairs = true;
    }
    if (c->mrg_capacity < total) {
        (void)hipFree(c->mrg_keys); (void)hipFree(c->mrg_keys2); (void)hipFree(c->mrg_vals[0]); (void)hipFree(c->mrg_vals[1]);
        (void)hipFree(c->mrg_colors);
        c->mrg_keys = c->mrg_keys2 = nullptr; c->mrg_vals[0] = c->mrg_vals[1] = nullptr; c->mrg_colors = nullptr; c->mrg_capacity = 0;
        const size_t cap = (size_t)total + (size_t)total / 4 + 1024;
        TH_HIP(hipMalloc((void **)&c->mrg_keys, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->mrg_keys2, cap * sizeof(unsigned long long)));
        TH_HIP(hipMalloc((void **)&c->mrg_vals[0], cap * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->mrg_vals[1], cap * sizeof(uint32_t)));
        TH_HIP(hipMalloc((void **)&c->mrg_colors, (c->mrg_pairs ? 2 : 1) * cap * sizeof(float4)));
        c->mrg_capacity = cap;
    }
    if (th_status s = deposit_temp(c, th::radix_sort_temp_bytes(total, 32, 32 + deposit_texel_bits(c)))) return s;
    if (!c->dep_total) TH_HIP(hipMalloc((void **)&c->dep_total, th::kTotWords * sizeof(uint32_t)));
    return TH_OK;
}

// One pass: this band's fragments parted by owner -> all-to-all -> the owner's merge -> all-gather of the owned ranges.
// du alone: the flow pass; ru alone: the view pass; both: both passes over one rasterisation and one exchange (two all-gathers)
static th_status sharded_pass(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *fragments)
{
    const int world = c->comm_world, rank = c->comm_rank;
    const bool view = ru != nullptr, both = ru != nullptr && du != nullptr;
    const size_t color_bytes = both ? 2 * sizeof(float4) : sizeof(float4);
    uint64_t count = 0;
    void *keys = nullptr, *colors = nullptr;
    // stage 1 - rasterise and part by owner.  Whether it worked travels with the counts: every rank learns of every failure
    std::vector<unsigned long long> hb((size_t)world + 1, 0ull);
    th_status mine = injected(c, 2);
    if (mine == TH_OK)
        mine = both ? th_draw_emit(c, du, ru, &count, &keys, &colors)
                    : (view ? th_view_emit(c, ru, &count, &keys, &colors) : th_deposit_emit(c, du, &count, &keys, &colors));
    // (the owners' bounds stay on the device: what this rank holds for every owner is their differences, computed there, and
    // comes back to the host together with what the others hold for it - one trip where there were two.  A rank with
    // something to report, or nothing to send, uploads its words itself.)
    unsigned long long *bounds = c->x_counts, *sendc = c->x_counts + 33, *recvc = c->x_counts + 65;
    const std::string why = mine != TH_OK ? last_error() : std::string();
    if (fragments) *fragments = mine == TH_OK ? count : 0;
    c->last_draw.pipeline = TH_DRAW_STREAM; c->last_draw.fragments = mine == TH_OK ? count : 0; c->last_draw.crowded_fragments = 0;
    std::vector<size_t> scount((size_t)world), soff((size_t)world), rcount((size_t)world), roff((size_t)world), one((size_t)world, 1), idx((size_t)world);
    std::vector<unsigned long long> hs((size_t)world, mine != TH_OK ? kPeerFailed : 0ull), hr((size_t)world);
    for (int r = 0; r < world; ++r) idx[(size_t)r] = (size_t)r;
    const bool on_device = mine == TH_OK && count;
    if (on_device) {
        th::launch_owner_bounds(static_cast<const unsigned long long *>(keys), (uint32_t)count, (uint32_t)world, bounds, c->stream);
        th::launch_exchange_send_counts(bounds, (uint32_t)world, sendc, c->stream);
        TH_HIP(hipGetLastError());
    } else TH_HIP(hipMemcpyAsync(sendc, hs.data(), (size_t)world * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    if (c->transport->alltoallv(c->comm, sendc, one.data(), idx.data(), recvc, one.data(), idx.data(), sizeof(unsigned long long), world, c->stream))
        return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    {
        std::vector<unsigned long long> words(97);
        if (th_status s = read_back(c, words.data(), c->x_counts, words.size() * sizeof(unsigned long long))) return s;
        if (on_device) std::copy(words.begin(), words.begin() + world + 1, hb.begin());
        std::copy(words.begin() + 65, words.begin() + 65 + world, hr.begin());
    }
    for (int r = 0; r < world; ++r) { scount[(size_t)r] = (size_t)(hb[(size_t)r + 1] - hb[(size_t)r]); soff[(size_t)r] = (size_t)hb[(size_t)r]; }
    if (mine != TH_OK) { last_error() = why; return mine; }
    for (int r = 0; r < world; ++r) if (hr[(size_t)r] & kPeerFailed) return peer_failure(c, r, "rasterising its band's lines");
    size_t total = 0;
    for (int r = 0; r < world; ++r) { rcount[(size_t)r] = (size_t)hr[(size_t)r]; roff[(size_t)r] = total; total += rcount[(size_t)r]; }
    for (int r = 0; r < world; ++r) if (r != rank) {
        c->last_draw.sent_bytes += 8 + scount[(size_t)r] * (sizeof(unsigned long long) + color_bytes);
        c->last_draw.received_bytes += 8 + rcount[(size_t)r] * (sizeof(unsigned long long) + color_bytes);
    }
    // stage 2 - room for what arrives and for its merge: the last thing that can fail on one rank alone
    mine = total < ((size_t)1 << 31) ? TH_OK : fail(TH_ERR_UNSUPPORTED, "too many fragments for one owner");
    auto room = [&]() -> th_status {
        if (c->x_capacity < total || (both && !c->x_pairs)) {
            (void)hipFree(c->x_keys); (void)hipFree(c->x_colors);
            c->x_keys = nullptr; c->x_colors = nullptr;
            const size_t cap = std::max(total, c->x_capacity) + total / 4 + 1024;
            c->x_capacity = 0;
            c->x_pairs = c->x_pairs || both;
            TH_HIP(hipMalloc((void **)&c->x_keys, cap * sizeof(unsigned long long)));
            TH_HIP(hipMalloc((void **)&c->x_colors, (c->x_pairs ? 2 : 1) * cap * sizeof(float4)));
            c->x_capacity = cap;
        }
        return merge_reserve(c, (uint32_t)total, both ? 2 : (view ? 1 : 0));
    };
    if (mine == TH_OK) mine = injected(c, 3);
    if (mine == TH_OK) mine = room();
    if (th_status s = agree_status(c, mine, "making room for the fragments it owns")) return s;
    // stage 3 - the exchange, the merge, the owned ranges back to everybody
    if (c->transport->alltoallv(c->comm, keys, scount.data(), soff.data(), c->x_keys, rcount.data(), roff.data(), sizeof(unsigned long long), world, c->stream) ||
        c->transport->alltoallv(c->comm, colors, scount.data(), soff.data(), c->x_colors, rcount.data(), roff.data(), color_bytes, world, c->stream))
        return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    if (th_status s = both ? th_draw_merge(c, c->x_keys, c->x_colors, total)
                           : (view ? th_view_merge(c, c->x_keys, c->x_colors, total) : th_deposit_merge(c, c->x_keys, c->x_colors, total))) return s;
    // the owners' texel ranges of the target(s) to every rank, in place
    const size_t texels = (size_t)c->fw * c->fh, chunk = (texels + (size_t)world - 1) / (size_t)world;
    for (int plane_of = 0; plane_of < 2; ++plane_of) {          // 0: the flow texture, 1: the view buffer
        if (plane_of == 0 ? (view && !both) : !view) continue;
        const size_t elem = plane_of ? sizeof(uchar4) : sizeof(float4);
        std::vector<size_t> gb((size_t)world), go((size_t)world);
        for (int r = 0; r < world; ++r) {
            const size_t lo = std::min(texels, (size_t)r * chunk), hi = std::min(texels, ((size_t)r + 1) * chunk);
            gb[(size_t)r] = (hi - lo) * elem; go[(size_t)r] = lo * elem;
        }
        char *plane = plane_of ? reinterpret_cast<char *>(c->view) : reinterpret_cast<char *>(c->flow);
        if (c->transport->allgather_bytes(c->comm, plane + go[(size_t)rank], plane, gb.data(), go.data(), rank, world, c->stream))
            return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
        for (int r = 0; r < world; ++r) if (r != rank) c->last_draw.received_bytes += gb[(size_t)r];
        if (world > 1) c->last_draw.sent_bytes += gb[(size_t)rank];
    }
    return TH_OK;
}

// The same pass through the BINS (th_bins.hip "the bins travel to the ranks that own them"): every rank rasterises its band's
// lines into its own page store over whatever slot order its ring is held in (the integrator's tile-sorted one: no return to
// texel order, no 64-bit keys, no sort anywhere); an owner owns whole bin rows; the bins change hands bin by bin with their
// counts beside them and are laid out in the owner's store as if it had emitted them; plan and blend kernels as in a single
// context.  kRetryInStreamOrder (on every rank together): a store could not be had or a bin outgrew its lists - the caller
// runs the stream-ordered pass.
static th_status sharded_pass_bins(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *fragments)
{
    const int world = c->comm_world, rank = c->comm_rank;
    const bool view = ru != nullptr, both = ru != nullptr && du != nullptr;
    th_deposit_uniforms d{};
    if (du) d = *du; else { d.viewSize[0] = ru->viewSize[0]; d.viewSize[1] = ru->viewSize[1]; d.time = ru->time; d.speedLimit = ru->speedLimit; }
    // ---- stage 1: my band's lines into my store, then compacted owner by owner
    th::DepositParams p;
    th::OwnerParams o{};
    uint32_t *host = nullptr, emitted = 0;
    std::vector<unsigned long long> hb((size_t)world + 1, 0ull);
    auto stage1 = [&]() -> th_status {
        if (th_status s = injected(c, 2)) return s;
        if (injected_give_up(c)) return kRetryInStreamOrder;
        if (th_status s = deposit_prepare_bins(c, &d, p)) return s;
        p.mode = both ? 2 : (view ? 1 : 0);
        if (view) { view_fields(c, ru, p); p.view = c->view; if (!both) p.line_half = 0.5f * drawn_line_width(c, TH_PASS_VIEW); }
        if (th_status s = bins_pass_emit(c, p, false)) return s;
        host = c->bins_totals_host;
        o.pool_used = host[th::kTotPool];                     // (my own bins stay in the store: what arrives goes behind them)
        emitted = host[th::kTotFragments];
        const uint32_t bins_y = p.nbins / p.bins_x;
        o.world = (uint32_t)world; o.rank = (uint32_t)rank;
        for (int r = 0; r <= world; ++r) o.bin_lo[r] = (uint32_t)((unsigned long long)bins_y * (unsigned)r / (unsigned)world) * p.bins_x;
        o.nb = o.bin_lo[rank + 1] - o.bin_lo[rank];
        if (c->own_bins < p.nbins) {
            TH_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->own_mem); c->own_mem = nullptr; c->own_bins = 0;
            const size_t words = (size_t)p.nbins * (1 + 2 + 1 + 1 + 32 * 1 + 32 * 2) + 2 * (2 + 33 + 32) + 64;
            TH_HIP(hipMalloc(&c->own_mem, words * sizeof(uint32_t)));
            c->own_bins = p.nbins;
        }
        unsigned long long *q = static_cast<unsigned long long *>(c->own_mem);           // (the 8-byte arrays first)
        o.offsets = q; q += (size_t)p.nbins + 1;
        q += 33;                                               // (the owners' bounds lived here: they lie beside the counts now)
        o.owner_bounds = c->x_counts;
        unsigned long long *recv_base = q; q += 32;
        o.recv_base = recv_base;
        o.src_prefix = q; q += (size_t)32 * p.nbins;
        uint32_t *w = reinterpret_cast<uint32_t *>(q);
        o.counts = w; w += p.nbins;
        o.bin_total = w; w += p.nbins;
        o.bin_page = w; w += p.nbins;
        o.table = w;
        const uint32_t total = emitted;                        // (an upper bound of what leaves)
        if (total) {
            if (th_status s = deposit_reserve(c, total, true, both)) return s;
            o.out_keys = c->dep_u64[0]; o.out_colors = c->dep_colors;
        }
        th::launch_bins_owner_counts(p, o, c->stream);
        if (total) th::launch_bins_owner_extract(p, o, c->stream);
        th::launch_exchange_send_counts(o.owner_bounds, (uint32_t)world, c->x_counts + 33, c->stream);
        TH_HIP(hipGetLastError());
        return TH_OK;                   // (the owners' bounds come back with the counts the others send: one trip, below)
    };
    th_status mine = stage1();
    const std::string why = mine != TH_OK ? last_error() : std::string();
    if (fragments) *fragments = mine == TH_OK ? emitted : 0;          // (what leaves + what stays)
    unsigned long long *sendc = c->x_counts + 33, *recvc = c->x_counts + 65;
    std::vector<size_t> scount((size_t)world), soff((size_t)world), rcount((size_t)world), roff((size_t)world), one((size_t)world, 1), idx((size_t)world);
    std::vector<unsigned long long> hs((size_t)world, mine == kRetryInStreamOrder ? kPeerRetries : kPeerFailed), hr((size_t)world);
    for (int r = 0; r < world; ++r) idx[(size_t)r] = (size_t)r;
    // (a rank whose stage 1 went through left its counts on the device - launch_exchange_send_counts; one that has something to
    // report uploads its words)
    if (mine != TH_OK) TH_HIP(hipMemcpyAsync(sendc, hs.data(), (size_t)world * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    if (c->transport->alltoallv(c->comm, sendc, one.data(), idx.data(), recvc, one.data(), idx.data(), sizeof(unsigned long long), world, c->stream))
        return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    {
        std::vector<unsigned long long> words(97);
        if (th_status s = read_back(c, words.data(), c->x_counts, words.size() * sizeof(unsigned long long))) return s;
        if (mine == TH_OK) std::copy(words.begin(), words.begin() + world + 1, hb.begin());
        std::copy(words.begin() + 65, words.begin() + 65 + world, hr.begin());
    }
    for (int r = 0; r < world; ++r) { scount[(size_t)r] = (size_t)(hb[(size_t)r + 1] - hb[(size_t)r]); soff[(size_t)r] = (size_t)hb[(size_t)r]; }
    if (mine != TH_OK && mine != kRetryInStreamOrder) { last_error() = why; return mine; }
    for (int r = 0; r < world; ++r) if (hr[(size_t)r] & kPeerFailed) return peer_failure(c, r, "rasterising its band's lines");
    for (int r = 0; r < world; ++r) if (hr[(size_t)r] & kPeerRetries) return kRetryInStreamOrder;
    size_t total = 0;
    for (int r = 0; r < world; ++r) { rcount[(size_t)r] = (size_t)(hr[(size_t)r] & 0xffffffffull); roff[(size_t)r] = total; total += rcount[(size_t)r]; }
    {   // (counts word, the per-bin counts of the owner's bins, keys + varyings of what changes hands)
        const size_t frag_bytes = sizeof(unsigned long long) + (both ? 2 : 1) * sizeof(float4);
        for (int r = 0; r < world; ++r) if (r != rank) {
            c->last_draw.sent_bytes += 8 + (size_t)(o.bin_lo[r + 1] - o.bin_lo[r]) * sizeof(uint32_t) + scount[(size_t)r] * frag_bytes;
            c->last_draw.received_bytes += 8 + (size_t)o.nb * sizeof(uint32_t) + rcount[(size_t)r] * frag_bytes;
        }
    }
    // ---- stage 2: every source's counts for my bins; room for what arrives; my store laid out for it (the last things that can
    // fail on one rank alone: agreed on before the fragments travel)
    {
        std::vector<size_t> tc((size_t)world), to((size_t)world), rc((size_t)world, o.nb), ro((size_t)world);
        for (int r = 0; r < world; ++r) { tc[(size_t)r] = o.bin_lo[r + 1] - o.bin_lo[r]; to[(size_t)r] = o.bin_lo[r]; ro[(size_t)r] = (size_t)r * o.nb; }
        if (c->transport->alltoallv(c->comm, o.counts, tc.data(), to.data(), const_cast<uint32_t *>(o.table), rc.data(), ro.data(), sizeof(uint32_t), world, c->stream))
            return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    }
    auto stage2 = [&]() -> th_status {
        if (th_status s = injected(c, 3)) return s;
        if (total >= ((size_t)1 << 31)) return fail(TH_ERR_UNSUPPORTED, "too many fragments for one owner");
        if (world > 1 && (c->x_capacity < total || (both && !c->x_pairs))) {
            (void)hipFree(c->x_keys); (void)hipFree(c->x_colors);
            c->x_keys = nullptr; c->x_colors = nullptr;
            const size_t cap = std::max(total, c->x_capacity) + total / 4 + 1024;
            c->x_capacity = 0;
            c->x_pairs = c->x_pairs || both;
            TH_HIP(hipMalloc((void **)&c->x_keys, cap * sizeof(unsigned long long)));
            TH_HIP(hipMalloc((void **)&c->x_colors, (c->x_pairs ? 2 : 1) * cap * sizeof(float4)));
            c->x_capacity = cap;
        }
        // (where every source's part starts in what arrives: the running sum of the received counts, made on the device)
        th::launch_exchange_recv_base(recvc, (uint32_t)world, const_cast<unsigned long long *>(o.recv_base), c->stream);
        TH_HIP(hipGetLastError());
        return TH_OK;
    };
    mine = stage2();
    if (th_status s = agree_status(c, mine, "making room for the bins it owns")) return s;
    // ---- stage 3: the bins travel; laid out and blended where they are owned; the owned rows back to everybody
    if (world == 1) { o.in_keys = o.out_keys; o.in_colors = o.out_colors; }        // (nobody to send to: the compacted parts are the received ones)
    else {
        if (c->transport->alltoallv(c->comm, o.out_keys, scount.data(), soff.data(), c->x_keys, rcount.data(), roff.data(), sizeof(unsigned long long), world, c->stream) ||
            c->transport->alltoallv(c->comm, o.out_colors, scount.data(), soff.data(), c->x_colors, rcount.data(), roff.data(), both ? 2 * sizeof(float4) : sizeof(float4), world, c->stream))
            return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
        o.in_keys = c->x_keys; o.in_colors = c->x_colors;
    }
    th_status laid = TH_OK;
    for (int attempt = 0;; ++attempt) {
        TH_HIP(hipMemsetAsync(c->dep_total, 0, th::kTotWords * sizeof(uint32_t), c->stream));
        bins_pass_expect(c, p);
        th::launch_bins_owner_insert(p, o, c->stream);
        TH_HIP(hipGetLastError());
        if (th_status s = bins_pass_totals(c, p)) return s;
        const uint32_t flags = host[th::kTotFlags];
        if (flags == 0) break;
        // (nothing has been blended; a bin beyond its lists' reach cannot be drawn through the bins at all)
        if ((flags & ~(th::kBinsPoolExhausted | th::kBinsBinFull)) || attempt >= 6) { laid = fail(TH_ERR_UNSUPPORTED, "the bins this rank owns could not be laid out (flags %u)", flags); break; }
        if (flags & th::kBinsBinFull) {         // (a wider page table, with the entries of what this rank emitted itself)
            if ((laid = bins_table_widen(c, p, true)) == kRetryInStreamOrder)
                laid = fail(TH_ERR_UNSUPPORTED, "a bin of the target received more fragments than its lists hold (%u places)", c->bin_max_pages * th::kBinPage * th::kBinReplicas);
            if (laid != TH_OK) break;
        }
        if ((flags & th::kBinsPoolExhausted) && (laid = bins_store_grow_keep(c, p, host[th::kTotPool] + host[th::kTotPool] / 2u + 64u)) != TH_OK) break;
    }
    if (laid == TH_OK) laid = bins_pass_finish(c, p, nullptr, false);
    // (an owner that could not lay its bins out has blended nothing; the others have: the draw is lost, and everybody says so)
    if (th_status s = agree_status(c, laid == kRetryInStreamOrder ? fail(TH_ERR_HIP, "the owner's store could not be had") : laid, "laying out the bins it owns")) return s;
    const uint32_t bins_y = p.nbins / p.bins_x;
    for (int plane_of = 0; plane_of < 2; ++plane_of) {          // 0: the flow texture, 1: the view buffer
        if (plane_of == 0 ? (view && !both) : !view) continue;
        const size_t elem = plane_of ? sizeof(uchar4) : sizeof(float4);
        std::vector<size_t> gb((size_t)world), go((size_t)world);
        for (int r = 0; r < world; ++r) {
            const size_t row_lo = std::min<size_t>((size_t)c->fh, ((size_t)bins_y * (size_t)r / (size_t)world) << th::kBinShift);
            const size_t row_hi = std::min<size_t>((size_t)c->fh, ((size_t)bins_y * ((size_t)r + 1) / (size_t)world) << th::kBinShift);
            gb[(size_t)r] = (row_hi - row_lo) * (size_t)c->fw * elem; go[(size_t)r] = row_lo * (size_t)c->fw * elem;
        }
        char *plane = plane_of ? reinterpret_cast<char *>(c->view) : reinterpret_cast<char *>(c->flow);
        if (c->transport->allgather_bytes(c->comm, plane + go[(size_t)rank], plane, gb.data(), go.data(), rank, world, c->stream))
            return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
        for (int r = 0; r < world; ++r) if (r != rank) c->last_draw.received_bytes += gb[(size_t)r];
        if (world > 1) c->last_draw.sent_bytes += gb[(size_t)rank];
    }
    return TH_OK;
}

th_status th_draw_sharded(th_context *c, const th_deposit_uniforms *du, const th_render_uniforms *ru, uint64_t *fragments)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(du, "null uniforms");
    TH_REQUIRE(c->comm, "th_draw_sharded needs the job's communicator (th_comm_init)");
    TH_REQUIRE(c->comm_world <= 32, "the owners' merge handles up to 32 ranks");
    if (ru) TH_REQUIRE(memcmp(du->viewSize, ru->viewSize, sizeof du->viewSize) == 0 && memcmp(&du->time, &ru->time, sizeof du->time) == 0 &&
                       memcmp(&du->speedLimit, &ru->speedLimit, sizeof du->speedLimit) == 0,
                       "the two passes of one draw share viewSize, time and speedLimit");      // (the same on every rank: a host error, not a rank-local one)
    const int world = c->comm_world, rank = c->comm_rank, W = c->cfg.width;
    c->last_draw.sent_bytes = c->last_draw.received_bytes = 0;
    if (!c->sharded_draw_ready) {
        // the fixed buffers of the exchange, once - and the ranks make sure that every one of them has them
        auto fixed = [&]() -> th_status {
            if (!c->x_counts) TH_HIP(hipMalloc((void **)&c->x_counts, 97 * sizeof(unsigned long long)));
            if (world > 1 && !c->x_halo) TH_HIP(hipMalloc((void **)&c->x_halo, (size_t)4 * W * sizeof(float4)));
            if (th_status s = line_rows(c)) return s;
            if (world > 1 && c->rows_cross_bands && !c->edge_rows) TH_HIP(hipMalloc((void **)&c->edge_rows, (size_t)4 * W * sizeof(float4)));
            return ru ? view_storage(c) : TH_OK;
        };
        if (th_status s = agree_status(c, fixed(), "allocating the exchange's buffers")) return s;
        c->sharded_draw_ready = true;
    } else if (ru) if (th_status s = view_storage(c)) return s;       // (a resized target: the same on every rank)
    // the neighbouring bands' edge rows of both state buffers (the fp32 row lookup of the vertex stream can land one row
    // beside a line's own row for some texture heights): my first row to the rank below, my last row to the rank above.
    // A packed ring sends the rows of its f32 views - what the stored texels decode to, what the lines are made of.
    // A job that draws through the bins keeps its slot order: where every vertex of every line is the line's own particle it
    // needs none of this; where the row lookup drifts (rows_cross_bands: a property of the texture's shape, the same on every
    // rank) a band's two edge rows are picked out of the slot order (th::LineSources) into texel order, f32, and sent from there.
    if (th_status s = line_rows(c)) return s;
    const bool bins = binned_shards(c);
    c->halo_lo = c->halo_hi = nullptr;
    if (world > 1 && bins && c->rows_cross_bands) {
        th::DepositParams p;
        th_status mine = injected(c, 1);
        if (mine == TH_OK) mine = deposit_prepare_bins(c, du, p);
        if (mine == TH_OK) { th::launch_bins_edge_rows(p, c->edge_rows, c->stream); if (hipGetLastError() != hipSuccess) mine = fail(TH_ERR_HIP, "the edge rows' gather could not be launched"); }
        if (th_status s = agree_status(c, mine, "gathering its edge rows")) return s;
        std::vector<size_t> sc((size_t)world, 0), so((size_t)world, 0), rc((size_t)world, 0), ro((size_t)world, 0);
        // to rank - 1: my first row of cur and of prev (its `hi`); to rank + 1: my last rows (its `lo`); two rows of W texels each
        if (rank > 0) { sc[(size_t)rank - 1] = 2; so[(size_t)rank - 1] = 0; rc[(size_t)rank - 1] = 2; ro[(size_t)rank - 1] = 0; }
        if (rank + 1 < world) { sc[(size_t)rank + 1] = 2; so[(size_t)rank + 1] = 2; rc[(size_t)rank + 1] = 2; ro[(size_t)rank + 1] = 2; }
        if (c->transport->alltoallv(c->comm, c->edge_rows, sc.data(), so.data(), c->x_halo, rc.data(), ro.data(), (size_t)W * sizeof(float4), world, c->stream))
            return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
        const uint64_t rows_moved = 2ull * ((rank > 0 ? 1u : 0u) + (rank + 1 < world ? 1u : 0u)) * (uint64_t)W * sizeof(float4);
        c->last_draw.sent_bytes += rows_moved; c->last_draw.received_bytes += rows_moved;
        c->halo_lo = rank > 0 ? c->x_halo : nullptr;
        c->halo_hi = rank + 1 < world ? c->x_halo + (size_t)2 * W : nullptr;
    }
    if (world > 1 && !bins) {
        th_status mine = c->packed ? injected(c, 1) : TH_OK;
        if (mine == TH_OK) mine = ensure_identity(c);
        const float4 *state[2] = {nullptr, nullptr};
        for (int b = 0; b < 2 && mine == TH_OK; ++b) {
            float4 *v = nullptr;
            mine = unpacked_view(c, c->ring[(size_t)b], b, &v);
            state[b] = v;
        }
        // (an f32 ring: nothing here fails on one rank alone - no agreement, no extra round trip)
        if (c->packed) { if (th_status s = agree_status(c, mine, "decoding its packed edge rows")) return s; }
        else if (mine != TH_OK) return mine;
        std::vector<size_t> sc((size_t)world, 0), so((size_t)world, 0), rc((size_t)world, 0), ro((size_t)world, 0);
        for (int b = 0; b < 2; ++b) {           // ring buffer b: one exchange each (the rows lie in different allocations)
            std::fill(sc.begin(), sc.end(), 0); std::fill(rc.begin(), rc.end(), 0);
            // to rank - 1: my first row (its `hi`); to rank + 1: my last row (its `lo`).  Offsets are in rows of W texels from `state`.
            if (rank > 0) { sc[(size_t)rank - 1] = 1; so[(size_t)rank - 1] = 0; rc[(size_t)rank - 1] = 1; ro[(size_t)rank - 1] = (size_t)b; }
            if (rank + 1 < world) { sc[(size_t)rank + 1] = 1; so[(size_t)rank + 1] = (size_t)c->cfg.height - 1; rc[(size_t)rank + 1] = 1; ro[(size_t)rank + 1] = 2 + (size_t)b; }
            if (c->transport->alltoallv(c->comm, state[b], sc.data(), so.data(), c->x_halo, rc.data(), ro.data(), (size_t)W * sizeof(float4), world, c->stream))
                return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
        }
        {
            const uint64_t rows_moved = 2ull * ((rank > 0 ? 1u : 0u) + (rank + 1 < world ? 1u : 0u)) * (uint64_t)W * sizeof(float4);
            c->last_draw.sent_bytes += rows_moved; c->last_draw.received_bytes += rows_moved;
        }
        c->halo_lo = rank > 0 ? c->x_halo : nullptr;
        c->halo_hi = rank + 1 < world ? c->x_halo + (size_t)2 * W : nullptr;
    }
    c->dep_owners = (uint32_t)world;
    auto pass = [&](const th_deposit_uniforms *d, const th_render_uniforms *r, uint64_t *n) -> th_status {
        if (bins) {
            const th_status s = sharded_pass_bins(c, d, r, n);
            if (s != kRetryInStreamOrder) return s;          // (on every rank together: the stream-ordered pass instead)
        }
        return sharded_pass(c, d, r, n);
    };
    if (ru && drawn_line_width(c, TH_PASS_FLOW) == drawn_line_width(c, TH_PASS_VIEW)) {
        // both passes draw the same lines: one rasterisation, one exchange of fragments carrying both varyings
        if (th_status s = pass(du, ru, fragments)) return s;
    } else {
        if (th_status s = pass(du, nullptr, fragments)) return s;
        if (ru) if (th_status s = pass(nullptr, ru, nullptr)) return s;
    }
    // (no synchronize: like every other entry point the call enqueues and returns - the all-gathers of the owned ranges are
    // on the context's stream, and so is whatever the host asks for next)
    return TH_OK;
}

// ---- one process per GPU: the communicator of the job's ranks and the path's collective (th_comm.hip) --------------------
th_status th_comm_unique_id(void *id_out)
{
    TH_REQUIRE(id_out, "null output");
    if (th::comm_unique_id(id_out, TH_COMM_ID_BYTES)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    return TH_OK;
}

#ifdef TH_TESTING
th_status th_comm_loopback_id(void *id_out)
{
    TH_REQUIRE(id_out, "null output");
    if (th::loopback_unique_id(id_out, TH_COMM_ID_BYTES)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    return TH_OK;
}
#endif

th_status th_comm_init(th_context *c, const void *id, int32_t rank, int32_t world)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(id, "null communicator id");
    TH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d outside world %d", rank, world);
    TH_REQUIRE(!c->comm, "the context already holds a communicator (th_comm_destroy first)");
    if (!c->d_status) TH_HIP(hipMalloc((void **)&c->d_status, 2 * sizeof(uint32_t)));
    if (th::comm_init(&c->comm, &c->transport, id, TH_COMM_ID_BYTES, rank, world)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    c->comm_rank = rank; c->comm_world = world;
    c->sharded_draw_ready = false;
    return TH_OK;
}

th_status th_comm_destroy(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    if (!c->comm) return TH_OK;
    TH_HIP(hipStreamSynchronize(c->stream));
    const int bad = c->transport->destroy(c->comm);
    c->comm = nullptr; c->transport = nullptr; c->comm_rank = 0; c->comm_world = 1;
    if (bad) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    return TH_OK;
}

th_status th_comm_query(th_context *c, th_comm_info *out)
{
    TH_REQUIRE(c && out, "null argument");
    *out = th_comm_info{};
    out->rank = c->comm_rank; out->world = c->comm_world; out->active = c->comm ? (c->transport == nullptr || strcmp(c->transport->name, "rccl") == 0 ? 1 : 2) : 0;
    int v = 0;
    if (th::comm_available(&v) == 0) out->rccl_version = v;
    return TH_OK;
}

// ---- row-band shards: the whole particle texture on every rank, for the spawners that sample arbitrary particles ----------
static th_status gather_storage(th_context *c, int32_t buffer)
{
    TH_REQUIRE(buffer >= 0 && buffer < (int32_t)c->ring.size(), "bad buffer %d (ring has %zu)", buffer, c->ring.size());
    if (!c->gathered) TH_HIP(hipMalloc((void **)&c->gathered, (size_t)c->cfg.width * c->cfg.global_height * sizeof(float4)));
    c->gathered_of = nullptr;
    return TH_OK;
}

th_status th_state_gather_ptr(th_context *c, int32_t buffer, void **dptr)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(dptr, "null output");
    if (th_status s = gather_storage(c, buffer)) return s;
    if (th_status s = ensure_identity(c)) return s;          // (the association below is with the buffer as the host will see it)
    c->gathered_of = c->ring[(size_t)buffer];
    *dptr = c->gathered;
    return TH_OK;
}

th_status th_state_gather(th_context *c, int32_t buffer)
{
    if (th_status s = use(c, true)) return s;
    TH_REQUIRE(c->comm, "th_state_gather needs the job's communicator (th_comm_init)");
    if (th_status s = gather_storage(c, buffer)) return s;
    if (th_status s = ensure_identity(c)) return s;          // bands travel in texel order
    // the bands of sharding.shard_rows: contiguous, balanced (the first H % world ranks hold one row more)
    const int world = c->comm_world, H = c->cfg.global_height, base = H / world, extra = H % world;
    std::vector<size_t> bytes((size_t)world), offset((size_t)world);
    for (int r = 0; r < world; ++r) {
        const int rows = base + (r < extra ? 1 : 0), row0 = r * base + (r < extra ? r : extra);
        bytes[(size_t)r] = (size_t)rows * c->cfg.width * sizeof(float4);
        offset[(size_t)r] = (size_t)row0 * c->cfg.width * sizeof(float4);
        if (r == c->comm_rank)
            TH_REQUIRE(rows == c->cfg.height && row0 == c->cfg.row0, "this context holds rows %d..%d, rank %d of %d balanced bands holds %d..%d",
                       c->cfg.row0, c->cfg.row0 + c->cfg.height, r, world, row0, row0 + rows);
    }
    float4 *data = nullptr;
    if (th_status s = unpacked_view(c, c->ring[(size_t)buffer], 2, &data)) return s;
    if (c->transport->allgather_bytes(c->comm, data, c->gathered, bytes.data(), offset.data(), c->comm_rank, world, c->stream))
        return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    c->gathered_of = c->ring[(size_t)buffer];
    return TH_OK;
}

th_status th_stats_allreduce(th_context *c)
{
    if (th_status s = use(c, true)) return s;
    if (!c->comm) return TH_OK;                    // a single-rank job: the local block is the global one
    if (c->transport->allreduce_counters(c->comm, c->d_counters, c->stream)) return fail(TH_ERR_UNSUPPORTED, "%s", th::comm_error());
    return TH_OK;
}

th_status th_stats_global(th_context *c, float speed_limit, th_counters *out)
{
    TH_REQUIRE(out, "null output");
    if (th_status s = th_stats_async(c, speed_limit, nullptr)) return s;
    if (th_status s = th_stats_allreduce(c)) return s;
    return read_back(c, out, c->d_counters, sizeof *out);
}

}  // extern "C"
