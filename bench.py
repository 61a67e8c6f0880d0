#!/usr/bin/env python3
"""bench.py - particle-steps/s of the Tendrils integrator on MI355X (BASELINE.json metric).

Workload (config C3 of BASELINE.json / SURVEY.md 8d): 4096 x 4096 state texture
(16,777,216 particles, RGBA32F), flow field 1920 x 1080 produced by the optical-flow
pass from a synthetic 1080p frame pair (falls back to a seeded synthetic field while
that pass is unavailable), reference default uniforms (noise on), fixed 60 Hz timer.
One "step" = one Tendrils.step() = one pass of the integrator kernel over every
particle this rank holds.  State, flow and frames are resident in HBM before the
timed region starts.

Multi-GPU (weak scaling): one process per GPU, each holds a 4096-row band of a
4096 x (4096*N) global texture; flow replicated; no data-path collective.  The
statistics counters are reduced with one small RCCL all-reduce per 16 steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode exact|fast] [--no-cpu]
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

N = 4096                        # particles per rank = N*N
FLOW_W, FLOW_H = 1920, 1080
BYTES_PER_PARTICLE_STEP = 32    # 16 B state read + 16 B written; 8 + 8 with --state f16
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
STATS_EVERY = 32        # steps per fused launch and per statistics reduction (= th::kMaxFusedSteps)


def synth_state(rank):
    rng = np.random.default_rng(12345 + rank)
    st = np.empty((N, N, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (N, N, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (N, N, 2))
    return st


def synth_frames():
    """frame0 = seeded band-limited pattern, frame1 = frame0 translated by (1.5, 0.7) px."""
    rng = np.random.default_rng(777)
    yy, xx = np.mgrid[0:FLOW_H, 0:FLOW_W].astype(np.float64)

    def pattern(dx, dy):
        img = np.zeros((FLOW_H, FLOW_W, 3))
        r = np.random.default_rng(778)
        for _ in range(24):
            fx, fy = r.uniform(-0.08, 0.08, 2)
            ph = r.uniform(0, 2 * np.pi, 3)
            amp = r.uniform(0.2, 1.0)
            for c in range(3):
                img[..., c] += amp * np.sin((xx - dx) * fx + (yy - dy) * fy + ph[c])
        img = (img - img.min()) / (img.max() - img.min())
        out = np.empty((FLOW_H, FLOW_W, 4), np.uint8)
        out[..., :3] = np.clip(np.rint(img * 255), 0, 255).astype(np.uint8)
        out[..., 3] = 255
        return out
    del rng
    return pattern(0.0, 0.0), pattern(1.5, 0.7)


def synth_flow(time_ms):
    """Divergence-free seeded field in reference flow format (Fx, Fy, t_deposit, alpha)."""
    yy, xx = np.mgrid[0:FLOW_H, 0:FLOW_W].astype(np.float32)
    r = np.random.default_rng(4242)
    psi_x = np.zeros((FLOW_H, FLOW_W), np.float32)
    psi_y = np.zeros((FLOW_H, FLOW_W), np.float32)
    for _ in range(12):
        fx, fy = r.uniform(-0.05, 0.05, 2).astype(np.float32)
        ph = np.float32(r.uniform(0, 2 * np.pi))
        a = np.float32(r.uniform(0.3, 1.0))
        c = a * np.cos(xx * fx + yy * fy + ph)
        psi_x += c * fy        # d(psi)/dy
        psi_y += -c * fx       # -d(psi)/dx
    s = np.float32(0.01) / max(np.abs(psi_x).max(), np.abs(psi_y).max())
    fl = np.empty((FLOW_H, FLOW_W, 4), np.float32)
    fl[..., 0] = psi_x * s
    fl[..., 1] = psi_y * s
    fl[..., 2] = time_ms
    fl[..., 3] = 1.0
    return fl


def measure_traffic(extra_args):
    """HBM traffic of one launch of the dominant integrator kernel from rocprofv3 PMC counters, as
    MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes of the
    same workload (short child runs of this script), FETCH_SIZE doubled (gfx950 tallies the 128-B
    requests of a wide coalesced stream at 64 B), both in KiB.  Runs before this process touches the
    GPU.  Returns bytes per launch, or None when the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="th_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.abspath(__file__), "--steps", "64", "--warmup", "32", "--no-cpu",
               "--no-traffic", "--traffic-child"] + extra_args
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=240, check=True)
            by_kernel = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        for kname in ("logic_fused_kernel", "logic_packed_kernel", "logic_kernel"):
                            if kname in row.get("Kernel_Name", ""):
                                by_kernel.setdefault(kname, []).append(float(row["Counter_Value"]))
                                break
            rows = next((by_kernel[k] for k in ("logic_fused_kernel", "logic_packed_kernel", "logic_kernel")
                         if k in by_kernel), [])
            if not rows:
                return None, "no %s rows for the integrator kernel" % counter
            vals[counter] = sum(rows) / len(rows)
        except (subprocess.SubprocessError, OSError) as e:
            return None, "%s pass failed: %s" % (counter, type(e).__name__)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        "rocprofv3 PMC: (2*FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE=%.0f WRITE_SIZE=%.0f" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that measure HBM traffic")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # PMC child: fused launches only
    ap.add_argument("--force-dist", action="store_true", help="init RCCL even at world size 1 (path check)")
    ap.add_argument("--flow-size", default=None, help="experiment: WxH of the flow/view instead of 1920x1080")
    ap.add_argument("--state", default="f32", choices=["f32", "f16"], help="state ring storage (f16 = packed 8 B/particle, config C5)")
    ap.add_argument("--in-view", action="store_true", help="experiment: keep every particle inside the view (|y*viewSize.y| < 1)")
    ap.add_argument("--flow-only", action="store_true", help="noiseWeight = 0 (preset 'Flow Only')")
    args = ap.parse_args()

    global FLOW_W, FLOW_H
    if args.flow_size:
        FLOW_W, FLOW_H = (int(v) for v in args.flow_size.lower().split("x"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world

    traffic, traffic_note = None, "not measured"
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    if rank == 0 and world == 1 and not args.no_traffic and not args.force_dist and not under_profiler:
        extra = ["--mode", args.mode, "--state", args.state] + (["--flow-only"] if args.flow_only else []) + \
                (["--in-view"] if args.in_view else []) + (["--flow-size", args.flow_size] if args.flow_size else [])
        traffic, traffic_note = measure_traffic(extra)

    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import tendrils_amd as ta
    from tendrils_amd import _capi
    from tendrils_amd.sharding import DeviceCounters, reduce_counters, shard_rows
    from tendrils_amd.tendrils import View

    opts = ta.defaults()
    opts.update(device=local_rank, mode=ta.TH_MODE_FAST if args.mode == "fast" else ta.TH_MODE_EXACT,
                row0=shard_rows(N * world, world, rank)[0], rows=N, globalHeight=N * world,
                stateFormat=ta.TH_STATE_F16 if args.state == "f16" else ta.TH_STATE_F32)
    t = ta.Tendrils(View(FLOW_W, FLOW_H), opts)
    t.resize()                       # viewRes 1920x1080 -> viewSize [1, 1.7778]; flow.shape = viewRes
    t.setup(N)
    if args.flow_only:
        t.state["noiseWeight"] = 0
    ctx = t.particles._ctx
    st0 = synth_state(rank)
    if args.in_view:
        st0[..., 1] *= np.float32(0.56)
    t.particles.upload_texels(st0)
    del st0

    # flow field: optical-flow pass over the synthetic frame pair (C3), else a seeded field
    time0 = 1000.0
    flow_source = "optical-flow(synthetic 1080p frame pair)"
    of = None
    try:
        from tendrils_amd.optical_flow import OpticalFlow
        f0, f1 = synth_frames()
        of = OpticalFlow(t, uniforms=dict(speed=0.08, offset=0.1, scaleUV=[-1, -1]))   # src/demo.main.js:526-530
        of.resize([FLOW_W, FLOW_H])
        of.set_pixels(f0)
        of.step()
        of.set_pixels(f1)
        of.update(dict(speedLimit=t.state["speedLimit"], time=time0, viewSize=t.viewSize))
        of.render()
    except (ImportError, ta.TendrilsHipError):
        of = None
        flow_source = "synthetic divergence-free field (optical-flow pass unavailable)"
        t.flow.set_pixels(synth_flow(time0))

    def sync_all():
        t.particles.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    t.timer.time = time0
    counters_dev = C.c_void_p()
    pending, dev_counters, ext_stream = [], None, None
    if dist is not None:
        sp = C.c_void_p()
        _capi.call("th_stream", ctx, C.byref(sp))
        ext_stream = torch.cuda.ExternalStream(sp.value)

    def stats_tick():
        """statistics of buffers[0], reduced over the ranks in place by RCCL on the context's stream"""
        nonlocal pending, dev_counters
        if pending:        # the context's stream waits (on the device) for the previous reduction before the
            with torch.cuda.stream(ext_stream):     # counters are overwritten; the host does not block
                for w in pending:
                    w.wait()
        pending = []
        _capi.call("th_stats_async", ctx, C.c_float(t.state["speedLimit"]), C.byref(counters_dev))
        if dist is not None:
            if dev_counters is None:
                dev_counters = DeviceCounters(counters_dev.value)
            with torch.cuda.stream(ext_stream):
                pending = dev_counters.all_reduce_async(dist)

    def run(k_steps):
        # the step loop runs as fused launches (Tendrils.step_n -> th_step_n), STATS_EVERY steps each;
        # between launches: statistics (+ their RCCL reduction) and the optical-flow refresh
        done = 0
        while done < k_steps:
            n = min(STATS_EVERY, k_steps - done)
            t.step_n(n)
            done += n
            if n == STATS_EVERY:
                stats_tick()
                if of is not None:      # keep the field alive: re-stamp it from the frame pair (blended)
                    of.update(dict(speedLimit=t.state["speedLimit"], time=t.timer.time, viewSize=t.viewSize))
                    of.render()

    run(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    run(args.steps)
    sync_all()
    wall = time.perf_counter() - t0

    # kernel-only pass for the roofline: the same K steps in the same launches as the timed region
    # (th_step_n: STATS_EVERY steps fused per logic_fused_kernel launch), a HIP event pair around every launch
    # on the context's own stream
    def run_kernel_only(k_steps):
        done = 0
        while done < k_steps:
            n = min(STATS_EVERY, k_steps - done)
            t.step_n(n)
            done += n
    ev_ms, k_ms, k_n = C.c_float(), C.c_float(), C.c_int32()
    _capi.call("th_kernel_timing", ctx, 1)
    _capi.call("th_timer_start", ctx)
    run_kernel_only(args.steps)
    _capi.call("th_timer_stop", ctx, C.byref(ev_ms))
    _capi.call("th_kernel_timing_read", ctx, C.byref(k_ms), C.byref(k_n))
    # and the single-step kernel (one Tendrils.step() per launch) for reference
    s_ms, s_n = C.c_float(), C.c_int32()
    for _ in range(32):
        t.timer.tick()
        t.step()
    _capi.call("th_kernel_timing_read", ctx, C.byref(s_ms), C.byref(s_n))
    # and the other arithmetic mode on the same fused launches (exact <-> fast; tolerance of fast mode: DESIGN.md 4)
    o_ms, o_n = C.c_float(), C.c_int32()
    other_mode = "exact" if args.mode == "fast" else "fast"
    _capi.call("th_set_mode", ctx, ta.TH_MODE_EXACT if other_mode == "exact" else ta.TH_MODE_FAST)
    run_kernel_only(8 * STATS_EVERY)
    _capi.call("th_kernel_timing_read", ctx, C.byref(o_ms), C.byref(o_n))
    _capi.call("th_set_mode", ctx, ta.TH_MODE_FAST if args.mode == "fast" else ta.TH_MODE_EXACT)
    _capi.call("th_kernel_timing", ctx, 0)
    sync_all()

    if pending:
        with torch.cuda.stream(ext_stream):
            for w in pending:
                w.wait()
    stats = t.particles.stats(t.state["speedLimit"])
    if dist is not None:
        tmax = torch.tensor([wall, ev_ms.value / 1e3, k_ms.value / 1e3, s_ms.value / 1e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, ev_s, kern_s, single_s = (float(v) for v in tmax)
        stats = reduce_counters(dist, stats, device="cuda")
    else:
        ev_s, kern_s, single_s = ev_ms.value / 1e3, k_ms.value / 1e3, s_ms.value / 1e3

    particles = N * N * world
    bytes_per_step = BYTES_PER_PARTICLE_STEP // (2 if args.state == "f16" else 1)
    value = particles * args.steps / wall
    # `value` includes the statistics reductions and the optical-flow refresh (every STATS_EVERY steps); the roofline
    # uses the kernel-only pass: mean duration of the launches that did the K steps (k_n launches)
    launches = max(int(k_n.value), 1)
    per_launch_s = kern_s                       # mean launch duration (event pair per launch)
    steps_per_launch = args.steps / launches
    alg_bytes_per_launch = bytes_per_step * N * N * steps_per_launch
    achieved = alg_bytes_per_launch / per_launch_s / 1e9
    packed = args.state == "f16"
    kernel_name = ("logic_fused_packed_kernel" if packed else "logic_fused_kernel") if steps_per_launch > 1 else \
        ("logic_packed_kernel" if packed else "logic_kernel")

    line = {
        "metric": "particle-steps/sec (16M particles per GPU)", "value": value, "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",     # arithmetic is fp32 in both storage formats
        "config": {"workload": "C3: 4096x4096 " + ("packed 8-B (SNORM16 pos + fp16 vel)" if args.state == "f16" else "RGBA32F")
                               + " state (16.8M particles) per GPU, flow 1920x1080 from "
                               + flow_source + ", reference default uniforms"
                               + (" with noiseWeight=0 (flow-only)" if args.flow_only else " (simplex noise on)")
                               + ", 60 Hz fixed timer",
                   "mode": args.mode, "state_storage": args.state, "particles_per_gpu": N * N,
                   "parallelism": "row-band shard x%d, flow replicated" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                     "kernel": kernel_name, "avg_launch_ms": per_launch_s * 1e3, "launches": launches,
                     "steps_per_launch": steps_per_launch, "particle_steps_per_launch": N * N * steps_per_launch,
                     "avg_step_ms_on_stream": ev_s / args.steps * 1e3,
                     "single_step_kernel": {"kernel": "logic_packed_kernel" if args.state == "f16" else "logic_kernel", "avg_launch_ms": single_s * 1e3,
                                            "achieved": bytes_per_step * N * N / single_s / 1e9},
                     "other_mode": {"mode": other_mode, "avg_launch_ms": o_ms.value, "steps_per_launch": STATS_EVERY,
                                    "achieved": bytes_per_step * N * N * STATS_EVERY / max(o_ms.value, 1e-9) / 1e6},
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch},
        "counters": stats,
    }

    if rank == 0 and world == 1 and not args.no_cpu:
        line["cpu_baseline"] = cpu_baseline(t)
    t.dispose()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


def cpu_baseline(t):
    """The oracle (CPU restatement, bit-equal to the reference shader) timed on this host's cores
    on a bounded sample of the same workload: whole 4096^2 steps until ~12 s have been spent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    st = synth_state(0)
    fl = t.flow.read()
    u = O.logic_uniforms(N, N, 1000.0 + 1000 / 60, 1000 / 60, view_size=t.viewSize,
                         **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
    # threads: the host may expose more CPUs than this process can run on (cgroup quota, SMT) - probe a few
    # OpenMP team sizes on a quarter sample and time the baseline with the best one
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gomp = C.CDLL("libgomp.so.1")
    O.logic_step(u, st[:64], fl)                 # warm (library load)
    scratch = np.zeros_like(st)                  # output buffer, touched once here
    best, cores = 0.0, ncpu
    for cand in sorted({ncpu, max(ncpu // 2, 1), max(ncpu // 4, 1), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}):
        gomp.omp_set_num_threads(int(cand))
        probe = st[:N // 4]
        O.logic_step(u, probe[:cand * 2], fl)    # spin the team up
        p0 = time.perf_counter()
        O.logic_step(u, probe, fl, out=scratch[:N // 4])
        rate = probe.shape[0] * N / (time.perf_counter() - p0)
        if rate > best:
            best, cores = rate, int(cand)
    gomp.omp_set_num_threads(cores)
    rows = N if best >= 20e6 else N // 4         # keep the leg within ~10-30 s on small hosts
    sample = st[:rows]
    done, t0 = 0, time.perf_counter()
    while True:
        O.logic_step(u, sample, fl, out=scratch[:rows])
        done += 1
        el = time.perf_counter() - t0
        if el > 12.0 or done >= 8:
            break
    return {"value": rows * N * done / el, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": "%d step(s) of rows [0,%d) x %d of the same state/flow (oracle/tendrils_oracle.c, "
                      "OpenMP over rows with the best of the probed team sizes, strict fp32; %d CPUs visible)" % (done, rows, N, ncpu)}


if __name__ == "__main__":
    main()
