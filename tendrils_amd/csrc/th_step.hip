// th_step.hip - Particles.step (src/particles.js:123-145) = one integrator pass over the state ring: th_step, and th_step_n =
// n fixed-step passes as one fused launch per <= 32 steps or a captured hipGraph.
#include "th_ctx.hpp"

using namespace thi;

namespace {

// Largest s2 with sqrt_rn(s2) <= limit (sqrt_rn monotonic), so that
// `0 < s2 <= cap` <=> `0 < speed <= speedLimit` <=> min(speed,limit)/speed == 1.
float s2_cap_for(float limit)
{
    if (!(limit > 0.0f)) return -1.0f;                       // never take the shortcut
    if (std::isinf(limit)) return std::numeric_limits<float>::max();
    double sq = (double)limit * (double)limit;
    if (sq >= (double)std::numeric_limits<float>::max()) return std::numeric_limits<float>::max();
    float c = (float)sq;
    while (sqrtf(c) > limit) c = nextafterf(c, 0.0f);
    for (;;) {
        float n = nextafterf(c, std::numeric_limits<float>::infinity());
        if (std::isinf(n) || sqrtf(n) > limit) break;
        c = n;
    }
    return c;
}

bool finite_uniforms(const th_logic_uniforms &u)
{
    const float *f = reinterpret_cast<const float *>(&u);
    for (size_t k = 0; k < sizeof(u) / sizeof(float); ++k)
        if (!std::isfinite(f[k])) return false;
    return true;
}

}  // namespace

// Build the launch parameters of one integrator pass and pick the kernel variant.
// ---- one integrator pass = plan (host decisions, may synchronise) + enqueue (launches only) -------
struct StepPlan {
    th::LogicParams p{};         // everything except in / out / perm / time_dev
    bool noise = false, use_targets = false, pow2 = false, decoded = false, generic = false;
    bool may_sort = false;       // this pass may run on (and produce) tile-sorted slots
};

// Pick the kernel variant and bring the slot layout up to date.  `u.time` must be the time of
// largest magnitude the plan will be used with (it only enters the domain checks here).
static th_status plan_step(th_context *c, const th_logic_uniforms &u, int32_t target, StepPlan &plan)
{
    const uint32_t W = (uint32_t)c->cfg.width, H = (uint32_t)c->cfg.global_height;
    th::LogicParams &p = plan.p;
    p = th::LogicParams{};
    p.flow = c->flow; p.flow_dec = c->flow_dec; p.targets = c->targets; p.lut = c->lut;
    p.count = (uint32_t)c->texels();
    p.width = W;
    p.row0 = (uint32_t)c->cfg.row0;
    p.wf = (float)W; p.hf = (float)H;
    plan.pow2 = is_pow2(W) && is_pow2(H);
    p.log2w = plan.pow2 ? ilog2(W) : 0;
    p.inv_w = 1.0f / p.wf; p.inv_h = 1.0f / p.hf; p.inv_wh = 1.0f / (p.wf * p.hf);
    p.fw = c->fw; p.fh = c->fh;
    p.fwf = (float)c->fw; p.fhf = (float)c->fh;
    p.half_fw = 0.5f * p.fwf; p.half_fh = 0.5f * p.fhf;
    p.fwm1 = (float)(c->fw - 1); p.fhm1 = (float)(c->fh - 1);
    p.u = u;
    p.s2_cap = s2_cap_for(u.speedLimit);

    // Preconditions of the specialised path (DESIGN.md "fast-path domain").
    plan.generic = c->opt.force_generic || !finite_uniforms(u);
    plan.noise = u.noiseWeight != 0.0f;
    plan.use_targets = u.target != 0.0f;
    if (!plan.generic) {
        // i = (x+.5 + (y+.5)W)/(WH) lies in (0, 1]; bound |vary(base, i, v)| <= |base|(1+|v|)
        double nscale = std::fabs((double)u.noiseScale) * (1.0 + std::fabs((double)u.varyNoiseScale)) * 1.001;
        double ntime = std::fabs((double)u.time) * std::fabs((double)u.noiseSpeed) *
                       (1.0 + std::fabs((double)u.varyNoiseSpeed)) * 1.001;
        if (ntime + 1237.0 >= (double)th::kNoiseDomain) plan.generic = true;  // z = uv + noiseTime (+1234.5678)
        double bound = nscale > 0.0 ? (double)th::kNoiseDomain / nscale : 3.0e38;
        // capped below |inert| = 1e6: a lane inside the bound cannot be inert, so the specialised path tests
        // the bound only and the inert pass-through lives on the (reference-order) fallback path
        p.pos_bound = (float)std::fmin(bound * 0.999, 999999.0);
        if (!(p.pos_bound > 0.0f)) plan.generic = true;
    }
    if (!plan.generic && !plan.use_targets) {
        // target == 0 multiplies (targets - pos) by an exact zero; dropping the read is only
        // value-preserving when the texture holds no NaN/Inf.
        if (!c->targets_checked) {
            unsigned int flag = 0;
            TH_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(unsigned int), c->stream));
            th::launch_finite_check(c->targets, c->texels(), c->d_flag, c->stream);
            TH_HIP(hipMemcpyAsync(&flag, c->d_flag, sizeof flag, hipMemcpyDeviceToHost, c->stream));
            TH_HIP(hipStreamSynchronize(c->stream));
            c->targets_nonfinite = flag != 0;
            c->targets_checked = true;
        }
        plan.use_targets = c->targets_nonfinite;
    }
    // Decode the flow once per step when that is cheaper than decoding per particle: it shrinks the
    // random-gather footprint (the L2/Infinity-Fabric miss traffic is what bounds this kernel).
    const size_t flow_texels = (size_t)c->fw * c->fh;
    plan.decoded = !plan.generic && c->texels() >= 2 * flow_texels;

    // Slot layout (texel order or a tile-sorted order): only ring -> ring passes of the specialised f32 kernels run on
    // sorted slots; the callers bring the layout up to date.
    plan.may_sort = plan.decoded && target == TH_TARGET_RING && sorting_possible(c) &&
                    c->total_steps >= c->hold_texel_order_until;
    return TH_OK;
}

// what a captured th_step_n sequence depends on besides the ring order and the kernel flags (`time` excluded: it lives in
// device memory); an explicit field list - the struct has padding and fields the captured launches never read
static bool same_key(const th::LogicParams &a, const th::LogicParams &b)
{
    th_logic_uniforms ua = a.u, ub = b.u;
    ua.time = ub.time = 0.0f;
    return a.flow == b.flow && a.flow_dec == b.flow_dec && a.targets == b.targets && a.lut == b.lut &&
           a.count == b.count && a.width == b.width && a.log2w == b.log2w && a.row0 == b.row0 &&
           a.wf == b.wf && a.hf == b.hf && a.fw == b.fw && a.fh == b.fh &&
           memcmp(&ua, &ub, sizeof ua) == 0 && a.s2_cap == b.s2_cap && a.pos_bound == b.pos_bound;
}

static uint32_t plan_flags(const StepPlan &plan)
{
    return (plan.noise ? 1u : 0u) | (plan.use_targets ? 2u : 0u) | (plan.pow2 ? 4u : 0u) | (plan.decoded ? 8u : 0u) |
           (plan.generic ? 16u : 0u);
}

static th_status timing_events(th_context *c, hipEvent_t *k0, hipEvent_t *k1)
{
    if (c->kt_used + 2 > c->kt_events.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        TH_HIP(hipEventCreate(&a)); TH_HIP(hipEventCreate(&b));
        c->kt_events.push_back(a); c->kt_events.push_back(b);
    }
    *k0 = c->kt_events[c->kt_used]; *k1 = c->kt_events[c->kt_used + 1];
    c->kt_used += 2;
    return TH_OK;
}

// Rotate / resolve the render target and launch (flow decode +) the integrator.  Launches only:
// safe inside a stream capture.  `time_dev` (optional) overrides plan.p.u.time on the device.
// `sorted`: the pass may read and write tile-sorted slots (else every ring buffer is in texel order already).
static th_status enqueue_step(th_context *c, const StepPlan &plan, int32_t target, float time, const float *time_dev,
                              bool timing, bool sorted = false)
{
    th::LogicParams p = plan.p;
    float4 *out = nullptr;
    if (th_status s = resolve_target(c, target, true, &out)) return s;
    // A packed (TH_STATE_F16) ring runs the packed kernel on the default path (ring -> ring, specialised
    // kernel); explicit targets and the generic kernel go through f32 staging.
    const bool packed_kernel = c->packed && target == TH_TARGET_RING && !plan.generic;
    float4 *in = c->ring[1], *rt = out;     // Particles.step binds buffers[1] as `particles` (src/particles.js:139)
    if (c->packed && !packed_kernel) {
        if (th_status s = unpacked_view(c, c->ring[1], 1, &in)) return s;
        if (th_status s = render_target(c, out, 0, &rt)) return s;
    }
    p.in = in;
    p.out = rt;
    p.u.time = time;
    p.time_dev = time_dev;

    // Sorted slots.  The input keeps its order; the output is written either at the same slots or - every
    // c->opt.resort_steps steps, and when the input is not sorted yet or was sorted for another view / field shape - at
    // the slots of a new sort keyed on the input positions (counted just before the launch).
    int in_order = sorted ? order_of(c, in) : -1, out_order = -1;
    bool use_sorted = false, scatter = false, count = false, gather = false;
    if (c->asort.pending && (!sorted || packed_kernel)) if (th_status s = asort_drop(c)) return s;
    bool async = false;
    if (sorted && packed_kernel) {
        // packed ring: the plain grid-stride kernel over the sorted slots; a re-sort is a plain move of the input
        // (tile_hist, scan, tile_scatter into the spare buffer, which then takes the input's place in the ring)
        const th::TileGeom g = tile_geom(c, p.u);
        const bool stale = in_order >= 0 && (!same_geom(c->orders[(size_t)in_order].geom, g) ||
                                             c->orders[(size_t)in_order].fw != c->fw || c->orders[(size_t)in_order].fh != c->fh);
        if (in_order < 0 || stale || c->steps_since_sort >= c->opt.resort_steps) {
            int fresh = -1;
            th::TileSortParams b;
            if (th_status s = begin_sort(c, g, in, in_order >= 0 ? c->orders[(size_t)in_order].perm : nullptr, &fresh, &b)) return s;
            b.state_out = c->spare;
            th::launch_tile_scatter(b, c->stream);
            TH_HIP(hipGetLastError());
            clear_graphs(c);               // captured sequences name the ring buffers: one of them changes places with the spare
            float4 *old = in;
            state_moved(c, old, c->spare);
            for (float4 *&r : c->ring) if (r == old) r = c->spare;
            c->spare = old;
            set_order(c, old, -1);
            in = c->ring[1];
            set_order(c, in, fresh);
            p.in = in;
            in_order = fresh;
        }
        p.perm = c->orders[(size_t)in_order].perm;
        out_order = in_order;
    } else if (sorted) {
        const th::TileGeom g = tile_geom(c, p.u);
        const bool stale = in_order >= 0 && (!same_geom(c->orders[(size_t)in_order].geom, g) ||
                                             c->orders[(size_t)in_order].fw != c->fw || c->orders[(size_t)in_order].fh != c->fh);
        if (stale) {                       // the chunk table no longer describes the field: start over from texel order
            if (th_status s = ensure_identity(c)) return s;
            in = c->ring[1]; out = rt = c->ring[0];
            p.in = in; p.out = rt;
            in_order = -1;
        }
        // The re-sort of a frame loop (th_order.hip: asort_start): while draws over the slot order are going on, no step counts
        // or scatters - the order laid out beside the last draw is taken up here, its copy of this step's input in the input's
        // place - and the next one is started behind the step that is `resort_steps` launches on.
        async = c->opt.async_sort && in_order >= 0 && plan.decoded && c->side && in == c->ring[1] && target == TH_TARGET_RING &&
                c->total_steps - c->last_binned_draw <= 2ll * c->opt.resort_steps;
        if (c->asort.pending) {
            const bool take = async && c->asort.valid && c->asort.src == in && c->asort.src_order == in_order && c->asort.at_step == c->total_steps &&
                              same_geom(c->orders[(size_t)c->asort.order].geom, g);
            if (!take) { if (th_status s = asort_drop(c)) return s; }
            else {
                TH_HIP(hipStreamWaitEvent(c->stream, c->asort.done, 0));
                clear_graphs(c);               // captured sequences name the ring buffers: one of them changes places with the copy
                float4 *old = in, *copy = c->asort.dst;
                const int fresh = c->asort.order;
                c->asort.pending = c->asort.valid = false;
                c->asort.src = nullptr;
                set_order(c, old, -1);
                state_moved(c, old, copy);
                c->ring[1] = copy; c->asort.dst = old;
                set_order(c, copy, fresh);
                in = copy; p.in = in; in_order = fresh;
                c->steps_since_sort = 0;
            }
        }
        scatter = !async && (in_order < 0 || c->steps_since_sort >= c->opt.resort_steps);
        use_sorted = true;
        // between two sorts the pass is the plain grid-stride kernel over the sorted slots (taps gathered from the
        // decoded plane: a wave's taps fall into one neighbourhood); the chunk kernel counts and scatters around a re-sort
        gather = !scatter && plan.decoded && (async || c->steps_since_sort + 1 < c->opt.resort_steps);
        p.geom = g;
        if (in_order >= 0) {
            const th_context::SlotOrder &o = c->orders[(size_t)in_order];
            p.perm = o.perm; p.chunks = o.chunks; p.nchunks = o.nchunks; p.records = o.records;
        }
        if (scatter) {
            // counted by the pass that wrote `in`?  Then the histogram is complete and every chunk has its table.
            const bool counted = in_order >= 0 && c->counted.buf == in && c->counted.order == in_order &&
                                 same_geom(c->counted.geom, g) && c->counted.at_step == c->total_steps;
            set_order(c, out, -1);         // the output buffer's old content (and order) dies here
            th::TileSortParams b;
            if (th_status s = begin_sort(c, g, in, in_order >= 0 ? c->orders[(size_t)in_order].perm : nullptr, &out_order, &b, counted)) return s;
            p.cursor = b.cursor; p.perm_out = b.perm_out;
            p.use_records = counted ? 1u : 0u;
            // draws over the slot order are going on (th_bins.hip): the pass moves its INPUT along to the new slots, so
            // that buffers[0] and buffers[1] - the two ends of every line - stay in one order
            if (in == c->ring[1] && c->total_steps - c->last_binned_draw <= 2ll * c->opt.resort_steps) p.in_moved = c->spare;
        } else {
            out_order = in_order;
            count = !gather && c->steps_since_sort + 1 >= c->opt.resort_steps;      // the next pass will re-sort: count for it
            if (count) {
                if (th_status s = sort_storage(c)) return s;
                p.hist = c->tile_mem;
                TH_HIP(hipMemsetAsync(p.hist, 0, kTileWords / 2 * sizeof(uint32_t), c->stream));
            }
        }
    }

    if (plan.decoded)
        th::launch_flow_decode(c->flow, c->flow_dec, (size_t)c->fw * c->fh, time, time_dev, p.u.flowDecay, c->stream);

    // A frame loop: the plain kernel notes per 64 slots whether any of their lines - input position to output position - may
    // touch the view (LogicParams::seen); the draw that follows skips the blocks of 256 slots of which none may (44 % of the
    // bench's particles live outside the view, and the tile order keeps them together).  Hidden for sure = both ends beyond one
    // edge by more than 2 texels: more than a line of width <= 2 reaches (its diamonds: one texel) and its snapping moves.
    // Never inside a stream capture (th_step_n's graphs: time_dev set): the bytes would be allocated on a capturing thread, the
    // captured launch would keep writing them at every replay, and `seen` would describe a launch that has not run.
    bool seeing = false;
    const float vx = p.u.viewSize[0], vy = p.u.viewSize[1];
    if (c->opt.skip_unseen && !time_dev && target == TH_TARGET_RING && !c->packed && !plan.generic && (gather || !use_sorted) && rt == out &&
        c->total_steps - c->last_binned_draw <= 2ll * c->opt.resort_steps && vx > 0.0f && vy > 0.0f && std::isfinite(vx) && std::isfinite(vy)) {
        if (!c->seen.bytes) {
            const size_t bytes = ((c->texels() + 63) / 64 + 7) & ~(size_t)3;
            TH_HIP(hipMalloc((void **)&c->seen.bytes, bytes));
            TH_HIP(hipMemsetAsync(c->seen.bytes, 0, bytes, c->stream));
        }
        const float mx = 4.0f / (float)c->fw, my = 4.0f / (float)c->fh;
        p.seen = c->seen.bytes;
        p.seen_xlo = (-1.0f - mx) / vx; p.seen_xhi = (1.0f + mx) / vx;
        p.seen_ylo = (-1.0f - my) / vy; p.seen_yhi = (1.0f + my) / vy;
        seeing = true;
    }

    hipEvent_t k0 = nullptr, k1 = nullptr;
    if (timing && c->kernel_timing) {
        if (th_status s = timing_events(c, &k0, &k1)) return s;
        TH_HIP(hipEventRecord(k0, c->stream));
    }
    if (gather) {
        th::launch_logic(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, plan.decoded, plan.generic, packed_kernel, c->stream);
    } else if (use_sorted)
        th::launch_logic_sorted(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, in_order >= 0, scatter, count, c->max_chunks, c->stream);
    else
        th::launch_logic(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, plan.decoded, plan.generic, packed_kernel,
                         c->stream);
    if (k1) TH_HIP(hipEventRecord(k1, c->stream));
    TH_HIP(hipGetLastError());
    if (target == TH_TARGET_RING || (target >= 0 && target < (int32_t)c->ring.size())) set_order(c, out, out_order);
    if (p.in_moved) {                       // the moved copy takes the input's place in the ring
        clear_graphs(c);
        float4 *old = c->ring[1];
        set_order(c, old, -1);
        state_moved(c, old, c->spare);
        c->ring[1] = c->spare; c->spare = old;
        set_order(c, c->ring[1], out_order);
    }
    if (c->packed && !packed_kernel)
        if (th_status s = commit_target(c, out, rt)) return s;
    if (seeing) {
        c->seen.cur = out; c->seen.prev = in; c->seen.order = out_order;
        c->seen.stamp = out_order >= 0 ? c->orders[(size_t)out_order].stamp : 0ull;
        c->seen.view_x = vx; c->seen.view_y = vy; c->seen.fw = c->fw; c->seen.fh = c->fh;
    }
    ++c->steps_since_sort; ++c->total_steps;
    if (count) { c->counted.buf = out; c->counted.order = out_order; c->counted.geom = p.geom; c->counted.at_step = c->total_steps; }
    else c->counted.buf = nullptr;
    if (async && !c->asort.pending && out_order >= 0 && c->steps_since_sort >= c->opt.resort_steps)
        if (th_status s = asort_start(c, p.geom, out, out_order)) return s;
    return TH_OK;
}

extern "C" {

th_status th_step(th_context *c, const th_logic_uniforms *u, int32_t target)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(u, "null uniforms");
    // Particles.step reads this.buffers[1] (src/particles.js:139): needs >= 2 buffers
    TH_REQUIRE(c->ring.size() >= 2, "step needs at least 2 state buffers (have %zu)", c->ring.size());
    StepPlan plan;
    if (th_status s = plan_step(c, *u, target, plan)) return s;
    const bool sorted = plan.may_sort && !plan.generic;
    if (!sorted) { if (th_status s = asort_drop(c)) return s; if (th_status s = ensure_identity(c)) return s; }
    return enqueue_step(c, plan, target, u->time, nullptr, true, sorted);
}

// n fixed-step Tendrils.step() calls.  The launch sequence (2 kernels per step) is captured once into
// a hipGraph per (n, uniforms, ring order, layout) and replayed; the per-step `time` values live in a
// small device array refreshed before every replay, so replays need no node updates.
th_status th_step_n(th_context *c, const th_logic_uniforms *u, double time0, double dt_ms, int32_t n)
{
    if (th_status s = use(c)) return s;
    TH_REQUIRE(u && n >= 0, "bad arguments");
    TH_REQUIRE(c->ring.size() >= 2, "step needs at least 2 state buffers (have %zu)", c->ring.size());
    if (n == 0) return TH_OK;
    if (th_status s = asort_drop(c)) return s;          // (a frame loop's re-sort under way: these launches lay their own orders out)
    th_logic_uniforms v = *u;
    v.dt = (float)dt_ms;
    std::vector<float> times((size_t)n);
    double t = time0, tmax = 0.0;
    for (int32_t k = 0; k < n; ++k) {
        t += dt_ms;                                   // src/timer.js:28-31: time accumulates in double
        times[(size_t)k] = (float)t;
        if (std::fabs(t) > std::fabs(tmax)) tmax = t;
    }
    v.time = (float)tmax;
    StepPlan plan;
    if (th_status s = plan_step(c, v, TH_TARGET_RING, plan)) return s;

    // Temporal fusion (logic_fused_kernel): all n steps of a particle in one pass, <= kMaxFusedSteps per launch.
    // Needs the plain 2-buffer ring (only the last two states survive n rotations) and the specialised kernel;
    // both ring formats.  th_options::fuse = 0 turns it off (the tests compare both paths).
    if (c->opt.fuse && n >= 2 && c->ring.size() == 2 && !plan.generic) {
        // Slot layout of the fused passes: the newest state (ring[0]) may be in a tile-sorted order; both outputs of a
        // pass keep the slots of its input.  (Re)sorted every c->opt.rebucket_steps steps by a plain move into the other
        // buffer, whose content (state n-1 of the previous call) the pass overwrites anyway.
        if (plan.may_sort) {
            const th::TileGeom g = tile_geom(c, plan.p.u);
            int o = order_of(c, c->ring[0]);
            const bool stale = o >= 0 && (!same_geom(c->orders[(size_t)o].geom, g) || c->orders[(size_t)o].fw != c->fw ||
                                          c->orders[(size_t)o].fh != c->fh);
            if (o < 0 || stale || c->steps_since_sort >= c->opt.rebucket_steps) {
                float4 *cur = c->ring[0], *other = c->ring[1];
                set_order(c, other, -1);
                int fresh = -1;
                th::TileSortParams b;
                if (th_status s = begin_sort(c, g, cur, o >= 0 ? c->orders[(size_t)o].perm : nullptr, &fresh, &b)) return s;
                b.state_out = other;
                th::launch_tile_scatter(b, c->stream);
                TH_HIP(hipGetLastError());
                set_order(c, other, fresh);
                set_order(c, cur, -1);                 // (its content is dead: the sorted copy is the newest state now)
                state_written(c, other); state_moved(c, cur, other);
                c->ring[0] = other; c->ring[1] = cur;
            }
        } else if (th_status s = ensure_identity(c)) return s;
        // The field does not change inside the call.  Without the noise the pass waits for its taps (a dependent gather per
        // step): the field's x, y, z packed 12 B apart once per call - three quarters of the footprint, and the band one
        // XCD taps fits its L2 (0.574 -> 0.546 ms per 20-step launch at C3; with the noise on the pass is bound by its
        // arithmetic and the packing pass only costs: 1.829 against 1.818 + 0.01)
        const bool pack3 = !plan.noise;
        if (pack3) {
            if (!c->flow3) TH_HIP(hipMalloc((void **)&c->flow3, (size_t)c->fw * c->fh * 3 * sizeof(float)));
            th::launch_flow_pack3(c->flow, c->flow3, (size_t)c->fw * c->fh, c->stream);
        }
        {
            int32_t done = 0;
            while (done < n) {
                const int32_t m = std::min<int32_t>(n - done, (int32_t)th::kMaxFusedSteps);
                th::LogicParams p = plan.p;
                p.flow3 = pack3 ? c->flow3 : nullptr;
                float4 *cur = c->ring[0], *other = c->ring[1];
                const int order = order_of(c, cur);
                state_written(c, cur); state_written(c, other);
                p.in = cur;
                // a lane only ever touches its own texel, so one of the two outputs may overwrite the input;
                // after m rotations of [cur, other]: m even -> [cur, other], m odd -> [other, cur]
                p.out = (m & 1) ? other : cur;             // state m     (ends up in buffers[0])
                p.out_prev = (m & 1) ? cur : other;        // state m - 1 (ends up in buffers[1])
                p.perm = order >= 0 ? c->orders[(size_t)order].perm : nullptr;
                p.nsteps = (uint32_t)m;
                for (int32_t k = 0; k < m; ++k) p.times[k] = times[(size_t)(done + k)];
                // the last launch of the call takes the statistics of the state it leaves in buffers[0] (a packed ring's: of
                // what the stored texels decode to)
                const bool takes_stats = done + m == n;
                if (takes_stats) {
                    const uint32_t parts = th::fused_stats_parts(p.count, p.perm != nullptr), need = parts + (parts + 255u) / 256u + 16u;
                    if (c->fused_parts_cap < need) {
                        TH_HIP(hipStreamSynchronize(c->stream));
                        (void)hipFree(c->fused_parts); c->fused_parts = nullptr; c->fused_parts_cap = 0;
                        TH_HIP(hipMalloc((void **)&c->fused_parts, (size_t)need * sizeof(th::StatsPartial)));
                        // (no memset, here or in front of a launch: every wave writes its partial - an empty one where it met no particle)
                        c->fused_parts_cap = need;
                    }
                    p.stats_part = c->fused_parts;
                    c->fused_stats.nparts = parts; c->fused_stats.limit = p.u.speedLimit;
                }
                hipEvent_t k0 = nullptr, k1 = nullptr;
                if (c->kernel_timing) {
                    if (th_status s = timing_events(c, &k0, &k1)) return s;
                    TH_HIP(hipEventRecord(k0, c->stream));
                }
                th::launch_logic_fused(p, c->cfg.mode, plan.noise, plan.use_targets, plan.pow2, c->packed, c->stream);
                if (k1) TH_HIP(hipEventRecord(k1, c->stream));
                TH_HIP(hipGetLastError());
                set_order(c, other, order);                // both outputs sit at the input's slots
                c->counted.buf = nullptr;
                if (m & 1) { c->ring[0] = other; c->ring[1] = cur; }
                c->steps_since_sort += m; c->total_steps += m;
                done += m;
                if (takes_stats) { c->fused_stats.valid = true; c->fused_stats.buf = c->ring[0]; }
            }
            return TH_OK;
        }
    }

    // everything below runs in texel order
    if (th_status s = ensure_identity(c)) return s;
    if (!c->opt.graph || n < 2 || (c->packed && plan.generic)) {
        for (int32_t k = 0; k < n; ++k) {
            if (k) { v.time = times[(size_t)k]; if (th_status s = plan_step(c, v, TH_TARGET_RING, plan)) return s; }
            if (th_status s = enqueue_step(c, plan, TH_TARGET_RING, times[(size_t)k], nullptr, true)) return s;
        }
        return TH_OK;
    }

    // cache lookup: same n, same parameters (time excluded), same ring order and layout
    th::LogicParams key = plan.p;
    key.u.time = 0.0f;
    GraphEntry *hit = nullptr;
    for (GraphEntry &g : c->graphs)
        if (g.n == n && g.mode == c->cfg.mode && g.ring == c->ring && g.flags == plan_flags(plan) && same_key(g.key, key)) { hit = &g; break; }
    if (!hit) {
        if (c->graphs.size() >= 8) { destroy_graph(c->graphs.front()); c->graphs.erase(c->graphs.begin()); }
        GraphEntry g;
        g.n = n; g.mode = c->cfg.mode; g.ring = c->ring;
        g.flags = plan_flags(plan); g.key = key;
        TH_HIP(hipMalloc((void **)&g.times_dev, (size_t)n * sizeof(float)));
        TH_HIP(hipHostMalloc((void **)&g.times_host, (size_t)n * sizeof(float)));
        TH_HIP(hipEventCreate(&g.copied));
        const std::vector<float4 *> ring_before = c->ring;
        const int since_before = c->steps_since_sort;
        const long long total_before = c->total_steps;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        th_status st = TH_OK;
        if (e == hipSuccess) {
            for (int32_t k = 0; k < n && st == TH_OK; ++k)
                st = enqueue_step(c, plan, TH_TARGET_RING, 0.0f, g.times_dev + k, false);
            hipGraph_t graph = nullptr;
            e = hipStreamEndCapture(c->stream, &graph);
            if (e == hipSuccess && st == TH_OK) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
        }
        c->ring = ring_before;                         // the capture only recorded; nothing ran yet
        c->steps_since_sort = since_before;
        c->total_steps = total_before;
        if (e != hipSuccess || st != TH_OK) {
            destroy_graph(g);
            if (st != TH_OK) return st;
            return fail(TH_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e));
        }
        c->graphs.push_back(g);
        hit = &c->graphs.back();
    }
    TH_HIP(hipEventSynchronize(hit->copied));          // previous replay's copy out of times_host is done
    memcpy(hit->times_host, times.data(), (size_t)n * sizeof(float));
    TH_HIP(hipMemcpyAsync(hit->times_dev, hit->times_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    TH_HIP(hipGraphLaunch(hit->exec, c->stream));
    for (int32_t k = 0; k < n; ++k) {                  // host-side ring bookkeeping of the n rotations
        float4 *last = c->ring.back();
        c->ring.pop_back();
        c->ring.insert(c->ring.begin(), last);
    }
    // the replay wrote the buffers the capture's resolve_target() calls named - at capture time only: what is remembered of
    // their content (a step's `seen` bytes, a gathered copy, a re-sort's copy) ends here, as it does behind a plain step
    for (float4 *r : c->ring) state_written(c, r);
    c->steps_since_sort += n; c->total_steps += n;
    // times_host must stay untouched until the copy has run; a later replay of this entry waits here
    TH_HIP(hipEventRecord(hit->copied, c->stream));
    return TH_OK;
}

}  // extern "C"
