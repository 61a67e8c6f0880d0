#!/usr/bin/env python3
"""bench.py - particle-steps/s of the Tendrils integrator on MI355X (BASELINE.json metric).

Workload (default: config C3 of BASELINE.json / SURVEY.md 8d): 4096 x 4096 state texture
(16,777,216 particles, RGBA32F) per GPU, flow field 1920 x 1080 produced by the optical-flow
pass from a synthetic 1080p frame pair, reference default uniforms (noise on), fixed 60 Hz timer.
One "step" = one Tendrils.step() = one pass of the integrator over every particle this rank
holds.  State, flow and frames are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode exact|fast] [--config c3|c4|c5]

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself: the parent - before it touches
the GPU - runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's
JSON line and exits with the child's status.  Under a launcher (WORLD_SIZE set) every process is one rank.

Timed region: W warm-up steps, then an untimed clock pre-roll (>= 50 ms of the same launches,
disclosed as `preroll_ms`: a GPU that has idled for a few ms runs its first launches ~10 %
slower, profiles/r2_a_*), then `--reps` (5) repetitions of: barrier + synchronize, EXACTLY K steps,
barrier + synchronize (the barrier: the ranks of one node meeting in shared memory - benchlib/job.py NodeBarrier; `barrier` in the
line says which); max over ranks per repetition; `value` / `ms_per_step` are the MEDIAN repetition
(`repetitions` carries all of them and their spread).  Every timed repetition contains the path's
collective: after every fused launch the statistics pass and - world > 1 - the library's own RCCL
all-reduce of the counter block (th_stats_allreduce, on the context's stream); `rccl` reports the ranks
the reduction saw.

Roofline block.  The step loop runs as fused launches (th_step_n: <= 32 steps of a
particle back to back in registers), which stream 48/n bytes per particle-step instead of 32 and are
bound by VALU issue, not by HBM.  The line therefore carries
  * roofline.achieved / peak / hbm_equivalent_frac (= equivalent_frac): the SURVEY.md 8d figure - ALGORITHMIC
    bytes (32 B x particles x steps of one launch) / mean launch duration, against 8 TB/s.  An equivalent
    single-step bandwidth, not a physical one (for a register-resident fused launch it can exceed the peak);
  * roofline.bound / frac : ONE rule for every entry, the headline included (benchlib/roofline.bind(); `frac_is`
    names it): the fraction of the resource the entry names and runs out of - VALU issue or physical HBM traffic
    for a fused launch (PMC child runs; unknown without them), HBM (algorithmic bytes, which a single-step launch
    really streams) for one step per launch - never above 1.  The headline's fused launch reads `"bound": "valu"`;
  * roofline.hbm_physical       : rocprofv3 PMC bytes per launch (2 x FETCH_SIZE + WRITE_SIZE KiB,
    as MI355X_MICROARCH.md prescribes) / the same duration;
  * roofline.valu               : SQ_INSTS_VALU per launch / duration against the chip's VALU issue
    peak (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction), plus the clock the
    chip actually held (GRBM_GUI_ACTIVE) - "bound": "valu";
  * roofline.flow_only          : the same for the second uniform set of BASELINE.md 3
    (noiseWeight = 0), roofline.single_step_kernel: one step per launch (what a
    step() + draw() frame loop runs).
PMC numbers come from short child runs of this script under rocprofv3 with launches of the SAME
length as the timed ones.
frame_loop (N = 1, c3): the reference's frame loop on the same particles - timer.tick(), step(),
draw() - after everything else: single-step launch, flow pass and view pass of draw() in ms,
fragments per draw, and the flow pass's own HBM roofline; `crowded`: the same loop
280 frames on, when the wake has crowded the target (step and draw with both passes).

Multi-GPU: one process per GPU.  c3: every rank holds a 4096-row band of a 4096 x (4096 N) texture
(weak scaling; the N = 1 line is the single-GPU bench).  c3_strong: the metric's own 16 M particles as ONE
4096 x 4096 texture row-sharded over the ranks (4096 // N rows each: strong scaling; key `c3_strong` of the
default invocation, with what a launch group costs beside its kernel under `fixed_costs`).  c4: 8192 x 8192
row-sharded (64 M particles in all, strong scaling), counters reduced every 16 steps (and, reported beside it, every step) - also run as
a second leg of the default c3 invocation (key `c4`), so that a scaling sweep of the driver's command
carries both curves.  c5: 16384 x 16384 packed fp16 state row-sharded, 16-step fused groups.  Flow
replicated; no data-path collective; the counter block is reduced by the library's RCCL all-reduce
(th_comm_init / th_stats_allreduce; torch.distributed only carries the 128-byte id, the barriers and
the max over ranks of the bench itself).

`--dry-run`: the launcher, the rank plumbing, the timed-region protocol and the collective on CPU
(gloo), stepping a small band with the CPU restatement - what tests/test_bench_launch.py runs at world
size 2; it measures nothing.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import workload as W  # noqa: E402
from benchlib.job import Job, median, repetition_block  # noqa: E402
from benchlib.launcher import dry_run, self_launch  # noqa: E402
from benchlib.legs import c3_strong_leg, c4_leg, c5_leg, cpu_baseline, frame_loop, frame_loop_sharded  # noqa: E402
from benchlib.pmc import measure_pmc, pmc_bytes  # noqa: E402,F401
from benchlib.roofline import bind, roofline_entry  # noqa: E402
from benchlib.sidelegs import SideLegs, Stages, comm_deadline  # noqa: E402
from benchlib.workload import (BYTES_PER_PARTICLE_STEP, CONFIGS, HBM_PEAK_GBS, MAX_FUSED, synth_rows, synth_state)  # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed K-step region (value = the median)")
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS), help="BASELINE.json config (default c3: the metric's)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child passes")
    ap.add_argument("--no-frame-loop", action="store_true", help="skip the step() + draw() frame-loop leg")
    ap.add_argument("--no-c3-strong", action="store_true", help="skip the strong-scaling leg of the metric's own particles (ONE 4096 x 4096 texture row-sharded) of a c3 run")
    ap.add_argument("--no-c4", action="store_true", help="skip the config-4 leg (8192 x 8192 row-sharded, strong scaling) of a c3 run")
    ap.add_argument("--no-c5", action="store_true", help="skip the config-5 leg (16384 x 16384 packed state row-sharded, strong scaling) of a c3 run")
    ap.add_argument("--dry-run", action="store_true", help="CPU plumbing check: gloo ranks stepping the CPU restatement")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)   # PMC child: launches of this length only
    ap.add_argument("--force-dist", action="store_true", help="init RCCL even at world size 1 (path check)")
    ap.add_argument("--no-library-comm", action="store_true", help="reduce the counters through torch.distributed instead of the library's own communicator (what the bench falls back to when th_comm_init fails on any rank)")
    ap.add_argument("--pretend-world", type=int, default=0, help="experiment: every rank holds the band rank 0 of a job of this many ranks would hold (the per-GPU share of a strong-scaling point on one GPU: tools/band_sweep.py)")
    ap.add_argument("--flow-size", default=None, help="experiment: WxH of the flow/view instead of 1920x1080")
    ap.add_argument("--state", default=None, choices=["f32", "f16"], help="state ring storage (f16 = packed 8 B/particle); default: the config's")
    ap.add_argument("--in-view", action="store_true", help="experiment: keep every particle inside the view (|y*viewSize.y| < 1)")
    ap.add_argument("--flow-only", action="store_true", help="noiseWeight = 0 (preset 'Flow Only') in the timed region")
    args = ap.parse_args()
    args.reps = max(args.reps, 1)

    if args.flow_size:
        W.FLOW_W, W.FLOW_H = (int(v) for v in args.flow_size.lower().split("x"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one (before anything touches the GPU) and relay rank 0's line
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if rank != 0:           # one JSON line on the job's stdout: the other ranks' (and their libraries') go nowhere
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if args.dry_run:
        return dry_run(args, rank, world)
    cfg = CONFIGS[args.config]
    stages = Stages(rank, world, {"metric": "particle-steps/sec (16M particles per GPU)" if args.config == "c3" else "particle-steps/sec (%s)" % args.config,
                                  "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup})
    state_fmt = args.state or cfg["state"]
    group = cfg["group"]
    launch_len = args.pmc_child or min(group, args.steps)

    pmc, pmc_note = {}, "not measured"
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    if rank == 0 and world == 1 and not args.no_traffic and not args.force_dist and not under_profiler and not args.pmc_child:
        extra = ["--mode", args.mode, "--config", args.config, "--state", state_fmt] + \
                (["--in-view"] if args.in_view else []) + (["--flow-size", args.flow_size] if args.flow_size else [])
        stages.at("PMC child passes (rocprofv3)")
        pmc, pmc_note = measure_pmc(extra, launch_len)

    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        stages.at("torch.distributed init (RCCL)")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import tendrils_amd as ta
    from tendrils_amd import _capi

    stages.at("context, state upload, the library's communicator")
    job = Job(args, args.config, rank, local_rank, world, dist, launch_len=launch_len,
              comm_guard=comm_deadline(rank, sys.argv[1:], os.path.abspath(__file__)))
    t, ctx = job.t, job.ctx
    width, rows = job.width, job.rows
    particles_rank = job.particles_rank
    sync_all, run, run_kernel_only, timed_kernels = job.sync_all, job.run, job.run_kernel_only, job.timed_kernels

    if args.pmc_child:
        # PMC child (under rocprofv3): the launches the parent times, nothing else.  Counters are read per kernel.
        L = args.pmc_child
        run_kernel_only(2 * L, L)
        run_kernel_only(4 * L, L)
        for _ in range(8):
            t.timer.tick()
            t.step()
        t.state["noiseWeight"] = 0
        run_kernel_only(3 * L, L)           # (few: see the flow-only leg of the parent)
        sync_all()
        job.dispose()
        return

    stages.at("warm-up")
    run(args.warmup)
    preroll_ms = job.preroll()
    stages.at("timed region")

    job.reductions = 0
    walls = job.timed_region(args.steps, args.reps)
    reductions_per_rep = job.reductions // args.reps

    # c4: the same K steps with the counters reduced after EVERY step (BASELINE.md config 4 names both cadences)
    walls_every_step = None
    if args.config == "c4":
        walls_every_step = job.timed_region(args.steps, max(args.reps // 2, 2), every=1, refresh=False)

    stages.at("kernel-only passes (roofline)")
    # kernel-only pass for the roofline: the same K steps in the same launches as the timed region, a HIP event
    # pair around every launch on the context's own stream
    run_kernel_only(2 * launch_len, launch_len)
    k_ms, k_n = timed_kernels(lambda: run_kernel_only(args.steps, launch_len))
    # and the single-step kernel (one Tendrils.step() per launch: what a step() + draw() frame loop runs)
    def singles():
        for _ in range(64):             # one whole re-sort period of the tile-sorted slot order
            t.timer.tick()
            t.step()
    singles()
    s_ms, s_n = timed_kernels(singles)
    # the other arithmetic mode on the same fused launches (exact <-> fast; tolerance of fast mode: DESIGN.md 4)
    other_mode = "exact" if args.mode == "fast" else "fast"
    _capi.call("th_set_mode", ctx, ta.TH_MODE_EXACT if other_mode == "exact" else ta.TH_MODE_FAST)
    run_kernel_only(2 * launch_len, launch_len)
    o_ms, o_n = timed_kernels(lambda: run_kernel_only(8 * launch_len, launch_len))
    _capi.call("th_set_mode", ctx, ta.TH_MODE_FAST if args.mode == "fast" else ta.TH_MODE_EXACT)
    sync_all()

    stages.at("statistics, flow-only launches")
    stats = job.global_stats()          # (local pass + the library's all-reduce: every rank holds the job's counters)

    # second uniform set of BASELINE.md 3: flow only (noiseWeight = 0), same launches, after everything else
    # (it changes the state: without the wander term velocities decay towards 0/0 = NaN, as in the reference)
    f_ms = f_n = fs_ms = flow_only_nan = 0
    if not args.flow_only:
        keep = t.state["noiseWeight"]
        t.state["noiseWeight"] = 0
        # few launches, straight from the live state: without the wander term the velocities decay by 0.72 per step and
        # underflow to 0/0 = NaN after a few hundred steps (the reference's behaviour) - dead particles cost nothing and
        # would flatter the number
        f_ms, f_n = timed_kernels(lambda: run_kernel_only(3 * launch_len, launch_len))
        fs_ms, _ = timed_kernels(singles)
        flow_only_nan = t.particles.stats(t.state["speedLimit"])["nan"]
        t.state["noiseWeight"] = keep
        sync_all()

    stages.at("max over ranks")
    walls = job.max_over_ranks(walls)
    kern_s, single_s = job.max_over_ranks([k_ms / 1e3, s_ms / 1e3])
    if walls_every_step is not None:
        walls_every_step = job.max_over_ranks(walls_every_step)
    wall = median(walls)

    particles = particles_rank * world
    packed = state_fmt == "f16"
    bytes_per_step = BYTES_PER_PARTICLE_STEP // (2 if packed else 1)
    value = particles * args.steps / wall
    launches = max(int(k_n), 1)
    steps_per_launch = args.steps / launches
    fused = steps_per_launch > 1

    def rl(launch_s, steps_in_launch, counters):
        return roofline_entry(job, launch_s, steps_in_launch, counters, bytes_per_step)

    main_cls = "fused" if fused else "single"
    if args.flow_only:
        main_cls += "_flow_only"
    head = bind(rl(kern_s, steps_per_launch, pmc.get(main_cls)), fused)
    kernel_name = ("logic_fused_packed_kernel" if packed else "logic_fused_kernel") if fused else \
        ("logic_packed_kernel" if packed else "logic_kernel")
    roofline = {"kernel": kernel_name, "launches": launches,
                "uniform_set": "flow only (noiseWeight = 0)" if args.flow_only else "default (simplex noise on)",
                "achieved_is": "algorithmic bytes (SURVEY.md 8d: %d B per particle-step) / mean launch duration - an equivalent "
                               "single-step bandwidth (hbm_equivalent_frac = equivalent_frac = achieved / peak); the fused launch streams "
                               "48/n B per particle-step (hbm_physical) and runs out of VALU issue: `bound` / `frac` name THAT resource "
                               "(`frac_is`)" % bytes_per_step,
                "pmc_note": pmc_note}
    # `bound` / `frac` are what bind() found - the resource the launch runs out of (VALU issue for a fused launch), never above 1;
    # the contract's figure for the dominant kernel (algorithmic bytes / launch duration against the HBM peak) stands
    # beside it under its own name: achieved / peak / hbm_equivalent_frac
    head["hbm_equivalent_frac"] = head["equivalent_frac"]
    roofline.update(head)
    single = bind(rl(single_s, 1, pmc.get("single")), False)
    single["kernel"] = "logic_packed_kernel" if packed else "logic_kernel over tile-sorted slots (gathered taps; every 64th launch re-sorts through logic_sorted_kernel)"
    roofline["single_step_kernel"] = single
    roofline["other_mode"] = {"mode": other_mode, "avg_launch_ms": o_ms, "steps_per_launch": launch_len,
                              "achieved": bytes_per_step * particles_rank * launch_len / max(o_ms, 1e-9) / 1e6,
                              "note": "fast mode is toleranced for ONE step only (DESIGN.md 4): not a headline"}
    if f_n:
        fo = bind(rl(f_ms / 1e3, launch_len, pmc.get("fused_flow_only" if launch_len > 1 else "single_flow_only")), launch_len > 1)
        fo["uniform_set"] = "flow only (noiseWeight = 0)"
        fo["single_step_kernel_ms"] = fs_ms
        fo["nan_particles_after"] = flow_only_nan
        roofline["flow_only"] = fo

    storage = "packed 8-B (SNORM16 pos + fp16 vel)" if packed else "RGBA32F"
    line = {
        "metric": "particle-steps/sec (16M particles per GPU)" if args.config == "c3" else
                  "particle-steps/sec (%s)" % args.config,
        "value": value, "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": cfg["scaling"],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",     # arithmetic is fp32 in both storage formats
        "preroll_ms": preroll_ms,
        "repetitions": repetition_block(walls, args.steps),
        "config": {"workload": (cfg["label"] % storage) + ", flow %dx%d from " % (W.FLOW_W, W.FLOW_H)
                               + job.flow_source + ", reference default uniforms"
                               + (" with noiseWeight=0 (flow-only)" if args.flow_only else " (simplex noise on)")
                               + ", 60 Hz fixed timer; step loop as fused launches of <= %d steps, statistics "
                                 "(+ the library's RCCL all-reduce of the counters) after every launch" % launch_len,
                   "mode": args.mode, "state_storage": state_fmt, "particles_per_gpu": particles_rank,
                   "parallelism": "row-band shard x%d, flow replicated" % world},
        "roofline": roofline,
        "rccl": job.rccl_block(stats, reductions_per_rep),
        "counters": stats,
        "barrier": job.barrier.kind if job.barrier is not None else "none (one rank)",
    }
    if walls_every_step is not None:
        wes = median(walls_every_step)
        line["counters_every_step"] = {"value": particles * args.steps / wes, "ms_per_step": wes / args.steps * 1e3,
                                       "note": "same K steps, one launch and one counter reduction per step"}

    if cfg["scaling"] == "strong":
        # what a launch group costs beside its kernel: the part of a strong-scaling point that does not shrink with the band
        from benchlib.legs import fixed_costs, short_frame_loop
        try:
            line["fixed_costs"] = fixed_costs(job)
        except Exception as e:            # noqa: BLE001
            line["fixed_costs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.config in ("c4", "c5") and not args.no_frame_loop:      # (a run of the configuration itself: its frame loop beside its headline)
            try:
                line["frame_loop"] = short_frame_loop(job, frames=12, reupload=args.config == "c4")
            except Exception as e:        # noqa: BLE001
                line["frame_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # (the side legs must not cost the line: whatever goes wrong in them is reported in their place, and one that never comes
    # back - a collective some rank does not reach - ends the job with the line as it stands: benchlib/sidelegs.py)
    stages.at("side legs")
    stages.done()                       # (the headline stands: from here on every leg has its own deadline)
    legs = SideLegs(line, rank)
    c3 = args.config == "c3"
    want_c3s = c3 and not args.no_c3_strong and not args.flow_size and not under_profiler
    want_c4 = c3 and not args.no_c4 and not args.flow_size and not under_profiler
    want_c5 = c3 and not args.no_c5 and not args.flow_size and not under_profiler
    after = (["c3_strong"] if want_c3s else []) + (["c4"] if want_c4 else []) + (["c5"] if want_c5 else [])
    if world == 1 and c3 and not args.no_frame_loop:
        legs.run("frame_loop", lambda: frame_loop(t, ctx, synth_state(rank)), after)
    if (world > 1 or args.force_dist) and c3 and not args.no_frame_loop:
        # (collective: every rank runs it; a rank-local failure inside th_draw_sharded ends the draw on every rank - th_shard.hip)
        legs.run("frame_loop_sharded", lambda: frame_loop_sharded(job), after)
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            line["cpu_baseline"] = cpu_baseline(t, width, min(rows, W.N))
        except Exception as e:            # noqa: BLE001
            line["cpu_baseline"] = {"value": None, "unit": "particle-steps/s", "cores": 0, "kind": "port",
                                    "sample": "failed: %s: %s" % (type(e).__name__, e)}
    job.dispose()
    if want_c3s:
        legs.run("c3_strong", lambda: c3_strong_leg(args, rank, local_rank, world, dist), after[1:])
    if under_profiler and c3 and not args.no_c4:
        # (a kernel-trace of this command should average the headline's launches, not mix them with config 4's)
        line["c4"] = {"skipped": "under a profiler: run without it (or --config c4) for the config-4 leg"}
    elif want_c4:
        # (every rank takes part; the leg runs the same code path as the headline, so what it can still fail on - memory -
        # fails on every rank alike)
        legs.run("c4", lambda: c4_leg(args, rank, local_rank, world, dist), ["c5"] if want_c5 else [])
    if under_profiler and c3 and not args.no_c5:
        line["c5"] = {"skipped": "under a profiler: run without it (or --config c5) for the config-5 leg"}
    elif want_c5:
        legs.run("c5", lambda: c5_leg(args, rank, local_rank, world, dist))
    if dist is not None:
        def finish():
            dist.barrier()
            dist.destroy_process_group()
        legs.run("shutdown", finish, record=False)
    if rank == 0:
        # RCCL prints its version banner through C stdio (flushed at exit when stdout is a pipe): push it out first, so
        # that the JSON line is the last thing on stdout
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

