"""The side legs of a bench line: config 4 beside the headline, the frame loop step() + draw(), the CPU baseline."""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import workload as W
from .job import Job, median, repetition_block
from .workload import HBM_PEAK_GBS, ROOT, synth_rows


def fixed_costs(job, groups=24):
    """What one launch group of the timed region costs beside its integrator kernel - the part of a strong-scaling point that
    does not shrink with the band: the host's time to enqueue a th_step_n, the statistics fold and the counter all-reduce
    (an event pair on the context's stream around each), and a group's wall time against its kernel's."""
    import numpy as np
    capi, ctx, t, L = job.capi, job.ctx, job.t, job.launch_len
    ms = C.c_float()
    limit = C.c_float(t.state["speedLimit"])

    def timed(fn):
        capi.call("th_timer_start", ctx)
        fn()
        capi.call("th_timer_stop", ctx, C.byref(ms))
        return ms.value
    fold, reduce_, enqueue = [], [], []
    for _ in range(groups):
        job.t.particles.sync()
        h0 = time.perf_counter()
        t.step_n(L)
        enqueue.append((time.perf_counter() - h0) * 1e3)
        fold.append(timed(lambda: capi.call("th_stats_async", ctx, limit, None)))
        if job.comm is not None:
            reduce_.append(timed(lambda: capi.call("th_stats_allreduce", ctx)))
    k_ms, k_n = job.timed_kernels(lambda: job.run_kernel_only(groups * L, L))
    job.sync_all()
    w0 = time.perf_counter()
    job.run(groups * L, every=L, refresh=False)
    job.t.particles.sync()
    wall = (time.perf_counter() - w0) / groups * 1e3
    empty = timed(lambda: None)
    brackets = []
    for _ in range(7):                    # what the timed region's bracket alone costs: [barrier + sync | nothing | barrier + sync]
        job.sync_all()
        b0 = time.perf_counter()
        job.sync_all()
        brackets.append((time.perf_counter() - b0) * 1e3)
    return {"steps_per_launch": L, "bracket_ms": float(np.median(brackets)), "launch_ms": k_ms, "group_wall_ms": wall, "beside_the_kernel_ms": wall - k_ms,
            "host_enqueue_ms": float(np.median(enqueue)), "stats_fold_ms": float(np.median(fold)),
            "allreduce_ms": float(np.median(reduce_)) if reduce_ else None, "empty_event_pair_ms": empty,
            "note": "per launch group of the timed region: th_step_n(%d) + th_stats_async (the fold of the statistics the launch took) + "
                    "th_stats_allreduce; fold / all-reduce: an event pair on the stream around the call, enqueue to done (an empty pair "
                    "beside them); group_wall_ms: %d groups back to back against the wall clock" % (L, groups)}


def short_frame_loop(job, frames=12, reupload=True):
    """tick(); step(); draw() on a strong-scaling configuration's particles (at N = 1 the whole texture: the local draw(); at
    N > 1 the band's step and th_draw_sharded): medians of an event pair around step and draw, the loop against the wall clock,
    which pipeline drew, per-particle cost - max over ranks."""
    import numpy as np
    from tendrils_amd import _capi
    t, ctx = job.t, job.ctx
    if job.world > 1 and job.comm is None:
        return {"skipped": "the library's communicator is not up: Tendrils.draw() of a band needs it"}
    ms, info = C.c_float(), _capi.DrawInfo()

    def timed(fn):
        _capi.call("th_timer_start", ctx)
        fn()
        _capi.call("th_timer_stop", ctx, C.byref(ms))
        return ms.value
    if reupload:                      # (config 5: the state the timed region left - generating 268 M particles again takes half a minute)
        job.upload_synthetic()
        t.timer.time = 1000.0
    t.renderView = True
    for _ in range(3):
        t.timer.tick(); t.step(); t.draw()
    step_ms, draw_ms, frags, pipes = [], [], [], []
    for _ in range(frames):
        t.timer.tick()
        step_ms.append(timed(t.step))
        draw_ms.append(timed(t.draw))
        _capi.call("th_draw_query", ctx, C.byref(info))
        frags.append(t.fragments); pipes.append(info.pipeline)
    job.sync_all()
    t0 = time.perf_counter()
    for _ in range(frames):
        t.timer.tick(); t.step(); t.draw()
    job.sync_all()
    wall = (time.perf_counter() - t0) / frames * 1e3
    med = job.max_over_ranks([float(np.median(step_ms)), float(np.median(draw_ms)), wall])
    particles = job.particles_rank * job.world
    return {"frames": frames, "step_ms": med[0], "draw_both_ms": med[1], "wall_ms_per_frame": med[2],
            "fragments_per_draw_this_rank": float(np.mean(frags)),
            "ns_per_particle_frame": med[2] * 1e6 / particles,
            "pipeline": "bins" if all(p == 1 for p in pipes) else ("stream" if not any(p == 1 for p in pipes) else "mixed"),
            "note": "first frames from the synthetic state; draw_both_ms: Tendrils.draw() with both passes (a band: th_draw_sharded); "
                    "ns_per_particle_frame = wall per frame / all particles of the job (C3's frame loop on one GPU: 1.28 ms / 16.8 M = 0.076)"}


def strong_leg(args, name, rank, local_rank, world, dist, costs=False, loop=False):
    """A strong-scaling configuration beside the metric's own: ONE texture row-sharded over the ranks, whatever N - `c4`
    (BASELINE.json config 4: 8192 x 8192, 64 M particles, counters reduced after every 16-step launch) and `c3_strong` (the
    metric's 16 M particles read as one 4096 x 4096 texture over the GPUs: 4096 // N rows per rank)."""
    job = Job(args, name, rank, local_rank, world, dist)
    job.run(args.warmup)
    job.preroll()
    job.reductions = 0
    reps = max(args.reps // 2, 3)
    walls = job.timed_region(args.steps, reps)
    reductions = job.reductions // reps
    walls = job.max_over_ranks(walls)
    stats = job.global_stats()
    particles = job.particles_rank * world
    mid = median(walls)
    out = {"value": particles * args.steps / mid, "unit": "particle-steps/s", "ms_per_step": mid / args.steps * 1e3,
           "scaling": "strong", "n_gpus": world, "steps": args.steps, "particles": particles,
           "particles_per_gpu": job.particles_rank, "rows_per_gpu": job.rows, "repetitions": repetition_block(walls, args.steps),
           "rccl": job.rccl_block(stats, reductions),
           "workload": (job.cfg["label"] % "RGBA32F") + ", same flow and uniforms as the headline, fused launches of <= %d steps, "
                       "statistics + counter all-reduce after every launch" % job.launch_len,
           "note": "K = %d steps run as %s: a short trailing launch streams 48 / n bytes per particle-step like any other and costs "
                   "its own statistics" % (args.steps, " + ".join(str(min(job.launch_len, args.steps - d)) for d in range(0, args.steps, job.launch_len)) + " step launches")}
    if job.share != world:
        out["pretend_world"] = job.share
    if costs:
        try:
            out["fixed_costs"] = fixed_costs(job)
        except Exception as e:            # noqa: BLE001
            out["fixed_costs"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if loop and not args.no_frame_loop:
        try:
            out["frame_loop"] = short_frame_loop(job)
        except Exception as e:            # noqa: BLE001
            out["frame_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
    job.dispose()
    return out


def c4_leg(args, rank, local_rank, world, dist):
    return strong_leg(args, "c4", rank, local_rank, world, dist, loop=True)


def c3_strong_leg(args, rank, local_rank, world, dist):
    return strong_leg(args, "c3_strong", rank, local_rank, world, dist, costs=True)



def c5_leg(args, rank, local_rank, world, dist):
    """BASELINE.json config 5 beside the metric's own configuration: 16384 x 16384 particles in the packed 8-byte ring
    (SNORM16 position + fp16 velocity: 16 B per particle-step) row-sharded over the ranks (strong scaling: 268 M particles in
    all, whatever N), 16-step fused launches - th_step_n: a captured sequence where it does not fuse -, the trail / history
    double-buffer being the two-buffer ring itself, counters reduced after every launch.  Carries its own roofline entry."""
    from .roofline import bind, roofline_entry
    job = Job(args, "c5", rank, local_rank, world, dist)
    job.run(args.warmup)
    job.preroll()
    job.reductions = 0
    reps = max(args.reps // 2, 3)
    walls = job.timed_region(args.steps, reps)
    reductions = job.reductions // reps
    job.run_kernel_only(2 * job.launch_len, job.launch_len)
    k_ms, k_n = job.timed_kernels(lambda: job.run_kernel_only(args.steps, job.launch_len))
    walls = job.max_over_ranks(walls)
    (kern_s,) = job.max_over_ranks([k_ms / 1e3])
    stats = job.global_stats()
    particles = job.particles_rank * world
    mid = median(walls)
    steps_per_launch = args.steps / max(int(k_n), 1)
    e = roofline_entry(job, kern_s, steps_per_launch, None, W.BYTES_PER_PARTICLE_STEP // 2)
    e = bind(e, False)          # (no counters in this leg: the algorithmic figure against the HBM peak, named as such)
    e.update(kernel="logic_fused_packed_kernel" if steps_per_launch > 1 else "logic_packed_kernel", launches=int(k_n),
             frac_is="algorithmic bytes (16 B per particle-step: 8 B packed texel read + 8 B written) per launch / mean launch "
                     "duration / 8 TB/s - an equivalent bandwidth: a fused launch streams 24/n B per particle-step and is "
                     "limited by VALU issue (DESIGN.md 5)")
    out = {"value": particles * args.steps / mid, "unit": "particle-steps/s", "ms_per_step": mid / args.steps * 1e3,
           "scaling": "strong", "n_gpus": world, "steps": args.steps, "particles": particles, "dtype": "f32 arithmetic on packed 8-B texels",
           "particles_per_gpu": job.particles_rank, "repetitions": repetition_block(walls, args.steps),
           "rccl": job.rccl_block(stats, reductions), "roofline": e,
           "workload": (job.cfg["label"] % "packed 8-B (SNORM16 pos + fp16 vel)") + ", same flow and uniforms as the headline, fused "
                       "launches of <= %d steps, statistics + counter all-reduce after every launch" % job.launch_len}
    if not args.no_frame_loop:
        try:
            out["frame_loop"] = short_frame_loop(job, frames=8, reupload=False)
        except Exception as e:            # noqa: BLE001
            out["frame_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
    job.dispose()
    return out


def frame_loop_sharded(job, frames=12):
    """The frame loop of a row-band job - timer.tick(); step(); draw() on every rank together - with the path's one real data
    exchange inside: Tendrils.draw() of a band context is th_draw_sharded (the bins - or, on the fallback, the fragments - to
    the ranks that own them over the library's communicator, then the all-gather of the owned rows of both targets).
    Per frame, max over ranks: the band's step, the draw with its exchange, the payload bytes this rank sent / received, and
    the loop against the wall clock between two barriers."""
    import torch
    from tendrils_amd import _capi
    t, ctx, dist = job.t, job.ctx, job.dist
    if job.comm is None:
        return {"skipped": "the library's communicator is not up (%s): Tendrils.draw() of a band needs it" % (job.comm_fallback or "world 1")}
    ms = C.c_float()

    def timed(fn):
        _capi.call("th_timer_start", ctx)
        fn()
        _capi.call("th_timer_stop", ctx, C.byref(ms))
        return ms.value

    st = synth_rows(job.width, job.rows, 777 + job.rank)
    t.particles.upload_texels(st)
    st = None
    t.timer.time = 1000.0
    info = _capi.DrawInfo()
    for _ in range(3):
        t.timer.tick(); t.step(); t.draw()
    step_ms, draw_ms, sent, recv, frags, pipes = [], [], [], [], [], []
    for _ in range(frames):
        t.timer.tick()
        step_ms.append(timed(t.step))
        draw_ms.append(timed(t.draw))              # (an event pair on the context's stream: the exchange's collectives are on it too)
        _capi.call("th_draw_query", ctx, C.byref(info))
        sent.append(info.sent_bytes); recv.append(info.received_bytes); frags.append(t.fragments); pipes.append(info.pipeline)
    job.sync_all()
    t0 = time.perf_counter()
    for _ in range(frames):
        t.timer.tick(); t.step(); t.draw()
    job.sync_all()
    wall = (time.perf_counter() - t0) / frames * 1e3
    med = [float(np.median(step_ms)), float(np.median(draw_ms)), wall, float(np.median(sent)), float(np.median(recv))]
    med = job.max_over_ranks(med)
    total = torch.tensor([float(np.mean(frags))], dtype=torch.float64, device="cuda")
    dist.all_reduce(total)
    return {"frames": frames, "n_gpus": job.world, "step_ms": med[0], "draw_both_ms": med[1], "wall_ms_per_frame": med[2],
            "sent_bytes_per_draw": med[3], "received_bytes_per_draw": med[4], "fragments_per_draw_all_ranks": float(total.item()),
            "pipeline": "bins" if all(p == 1 for p in pipes) else ("stream" if not any(p == 1 for p in pipes) else "mixed"),
            "note": "max over ranks of each rank's median; step_ms / draw_both_ms: an event pair on the context's stream around the call "
                    "(Tendrils.draw() of a band = th_draw_sharded: both passes, one exchange, its collectives on that stream); "
                    "wall_ms_per_frame: the loop between two barriers; bytes: th_draw_query (payload handed to / taken from the other "
                    "ranks per draw)"}


def frame_loop(t, ctx, state, frames=20):
    """SURVEY.md 8f-1/8f-2 beside the headline: the reference's frame loop - timer.tick(), step(), draw() - on the same
    particles: one single-step launch, the flow pass of draw() (the particle lines blended into the flow field in GL
    primitive order) and the view pass (the same lines into the RGBA8 view buffer), each timed with a HIP event pair
    on the context's stream."""
    from tendrils_amd import _capi
    ms = C.c_float()

    def timed(fn):
        _capi.call("th_timer_start", ctx)
        fn()
        _capi.call("th_timer_stop", ctx, C.byref(ms))
        return ms.value

    t.particles.upload_texels(state)
    t.timer.time = 1000.0
    keep = t.renderView
    for _ in range(5):
        t.timer.tick(); t.step(); t.draw()
    step_ms, flow_ms, view_ms, frags = [], [], [], []
    for _ in range(frames):
        t.timer.tick()
        step_ms.append(timed(t.step))
        t.renderView = False
        flow_ms.append(timed(t.draw))
        frags.append(t.fragments)
        t.renderView = True
        u, n = t.render_uniforms(), C.c_uint64(0)
        view_ms.append(timed(lambda: _capi.call("th_view_draw", ctx, C.byref(u), C.byref(n))))
    both_ms = []                          # Tendrils.draw() as it runs with renderView: both passes in one call (th_draw)
    for _ in range(5):
        t.timer.tick(); t.step()
        both_ms.append(timed(t.draw))
    frame_ms = []                         # ... and the frame as one piece: step() + draw() inside one event pair
    for _ in range(10):
        t.timer.tick()
        frame_ms.append(timed(lambda: (t.step(), t.draw())))
    # ... and the same loop once the wake has crowded the target (the reference's loop runs for minutes: after ~60 frames
    # at this size most fragments fall into texels with hundreds and thousands of them, and a draw waits for the
    # longest run of one texel): `settle` more frames untimed, then 50 timed
    def wall(n):
        """n frames of the loop as a host runs it - no event, no sync but the draw's own read-back - against the wall clock"""
        _capi.call("th_sync", ctx)
        t0 = time.perf_counter()
        for _ in range(n):
            t.timer.tick(); t.step(); t.draw()
        _capi.call("th_sync", ctx)
        return (time.perf_counter() - t0) / n * 1e3
    wall_ms = wall(20)
    settle = 230
    for _ in range(settle):
        t.timer.tick(); t.step(); t.draw()
    c_step, c_both, c_frags = [], [], []
    for _ in range(50):
        t.timer.tick()
        c_step.append(timed(t.step))
        c_both.append(timed(t.draw))
        c_frags.append(t.fragments)
    crowded = {"after_frames": 5 + frames + 5 + 20 + settle, "frames": 50, "step_ms": float(np.median(c_step)), "draw_both_ms": float(np.median(c_both)),
               "wall_ms_per_frame": wall(50),
               "slowest_frame": {"step_ms": float(np.max(c_step)), "draw_both_ms": float(np.max(c_both))},
               "fragments_per_draw": float(np.mean(c_frags)),
               "frame_ms_reference_loop": float(np.median(c_step)) + float(np.median(c_both))}
    t.renderView = keep
    lines, f = state.shape[0] * state.shape[1], float(np.mean(frags))
    texels = W.FLOW_W * W.FLOW_H
    # the binned pipeline (th_bins.hip; what `auto` runs over tile-sorted slots).  Per slot: the particle id (4 B); per line that
    # can draw (half of the rows: state-at-frame.glsl reads `current` twice in the others): two state texels (32 B); per
    # fragment: key + varying written (24 B) and read once where its bin is put in order (24 B); the target read and written (32 B
    # per texel).  Round 2's pipeline (three radix passes + a gather between emit and blend) moved 88 B per line + 124 B per
    # fragment: its model is kept beside for the comparison across rounds.
    alg = lines * 4.0 + lines * 0.5 * 32.0 + f * 48.0 + texels * 32.0
    alg_r2 = lines * 88.0 + f * 124.0
    # medians over the frames (a frame in which a store grows - a hipMalloc inside the pass - would otherwise own the mean);
    # the slowest frame is reported beside
    d, s_ms, b_ms = float(np.median(flow_ms)), float(np.median(step_ms)), float(np.median(both_ms))
    return {"frames": frames, "step_ms": s_ms, "draw_flow_ms": d, "draw_view_ms": float(np.median(view_ms)), "draw_both_ms": b_ms,
            "slowest_frame": {"step_ms": float(np.max(step_ms)), "draw_flow_ms": float(np.max(flow_ms)), "draw_view_ms": float(np.max(view_ms)),
                              "draw_both_ms": float(np.max(both_ms))},
            "fragments_per_draw": f, "frames_per_s": 1e3 / (s_ms + d),
            "frame_ms_reference_loop": s_ms + b_ms,
            "frame_ms": float(np.median(frame_ms)),
            "wall_ms_per_frame": wall_ms,
            "crowded": crowded,
            "pipeline": "binned (th_bins.hip): particles stay in the integrator's tile-sorted slot order; one fused rasterise + emit pass into "
                        "16x16-texel bins of the target, per-bin ordering by (texel, stream index) and blending in LDS",
            "roofline": {"bound": "hbm", "kernel": "flow pass of draw(): bins_fused_kernel + per-bin blend kernels", "achieved": alg / d / 1e6,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / d / 1e6 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_draw": alg,
                         "achieved_is": "4 B per slot + 32 B per drawable line + 48 B per fragment + 32 B per target texel / median duration of the pass "
                                        "(the pass is bound by the rasteriser's integer arithmetic and by latency, not by bytes: DESIGN.md 3.4)",
                         "r2_model": {"algorithmic_bytes_per_draw": alg_r2, "achieved": alg_r2 / d / 1e6, "frac": alg_r2 / d / 1e6 / HBM_PEAK_GBS,
                                      "note": "round 2's byte model (88 B per line + 124 B per fragment: what the stream-ordered pipeline moves) over "
                                              "this round's duration - comparable with round 2's frame_loop.roofline.frac"}},
            "note": "timer.tick(); step(); draw(): one single-step launch over tile-sorted slots + the flow pass; the view pass timed separately "
                    "(th_view_draw after th_flow_deposit: a full pass of its own in the binned pipeline), and both passes in one call "
                    "(th_draw, what Tendrils.draw() runs with renderView: one rasterisation, two varyings per fragment) over 5 more frames; "
                    "wall_ms_per_frame: the loop as a host runs it (no events, no sync but the draw's own read-back) against the wall clock, 20 frames"}


def cpu_baseline(t, width, rows_avail):
    """The oracle (CPU restatement, bit-equal to the reference shader) timed on this host's cores
    on a bounded sample of the same workload: whole steps of a row band until ~12 s have been spent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    st = synth_rows(width, rows_avail, 12345)
    fl = t.flow.read()
    gh = t.particles._global_height or t.particles.shape[1]
    u = O.logic_uniforms(width, gh, 1000.0 + 1000 / 60, 1000 / 60, view_size=t.viewSize,
                         **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
    # threads: the host may expose more CPUs than this process can run on (cgroup quota, SMT) - probe a few
    # OpenMP team sizes on a quarter sample and time the baseline with the best one
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gomp = C.CDLL("libgomp.so.1")
    O.logic_step(u, st[:64], fl)                 # warm (library load)
    scratch = np.zeros_like(st)                  # output buffer, touched once here
    best, cores = 0.0, ncpu
    q = rows_avail // 4
    for cand in sorted({ncpu, max(ncpu // 2, 1), max(ncpu // 4, 1), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}):
        gomp.omp_set_num_threads(int(cand))
        probe = st[:q]
        O.logic_step(u, probe[:cand * 2], fl)    # spin the team up
        p0 = time.perf_counter()
        O.logic_step(u, probe, fl, out=scratch[:q])
        rate = probe.shape[0] * width / (time.perf_counter() - p0)
        if rate > best:
            best, cores = rate, int(cand)
    gomp.omp_set_num_threads(cores)
    rows = rows_avail if best >= 20e6 else q     # keep the leg within ~10-30 s on small hosts
    sample = st[:rows]
    done, t0 = 0, time.perf_counter()
    while True:
        O.logic_step(u, sample, fl, out=scratch[:rows])
        done += 1
        el = time.perf_counter() - t0
        if el > 12.0 or done >= 8:
            break
    return {"value": rows * width * done / el, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": "%d step(s) of rows [0,%d) x %d of the same state/flow (oracle/tendrils_oracle.c, "
                      "OpenMP over rows with the best of the probed team sizes, strict fp32; %d CPUs visible)" % (done, rows, width, ncpu),
            # the reference itself (JS + GLSL) cannot run on the GPU box: /root/reference does not travel and the box has
            # no GL.  Its own CPU figure, measured in the build container, carried here for the record:
            "reference_on_cpu": {"value": 5.86e6, "unit": "particle-steps/s", "cores": 8, "kind": "reference",
                                 "sample": "1 step() of 4096^2 particles, the reference's own bundle (docs/js/index.js) on "
                                           "SwiftShader software WebGL in kaleido's headless Chromium, 8-core Xeon 2.1 GHz, "
                                           "measured in the build container, not on this host (BASELINE.md section 2)"}}


