#!/usr/bin/env python3
"""bench.py - particle-steps/s of the Tendrils integrator on MI355X (BASELINE.json metric).

Workload (default: config C3 of BASELINE.json / SURVEY.md 8d): 4096 x 4096 state texture
(16,777,216 particles, RGBA32F) per GPU, flow field 1920 x 1080 produced by the optical-flow
pass from a synthetic 1080p frame pair, reference default uniforms (noise on), fixed 60 Hz timer.
One "step" = one Tendrils.step() = one pass of the integrator over every particle this rank
holds.  State, flow and frames are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode exact|fast] [--config c3|c4|c5]

Timed region: W warm-up steps, then an untimed clock pre-roll (>= 50 ms of the same launches,
disclosed as `preroll_ms`: a GPU that has idled for a few ms runs its first launches ~10 %
slower, profiles/r2_a_*), barrier + synchronize, EXACTLY K steps, barrier + synchronize; max
over ranks.

Roofline block.  The step loop runs as fused launches (th_step_n: <= 32 steps of a
particle back to back in registers), which stream 48/n bytes per particle-step instead of 32 and are
bound by VALU issue, not by HBM.  The line therefore carries
  * roofline.achieved/peak/frac : the SURVEY.md 8d figure - ALGORITHMIC bytes (32 B x particles x
    steps of one launch) / mean launch duration, against 8 TB/s.  An equivalent single-step
    bandwidth, not a physical one (it may exceed 1 for the flow-only uniform set);
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
fragments per draw, and the flow pass's own HBM roofline.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  c3: every rank holds a
4096-row band of a 4096 x (4096 N) texture (weak scaling).  c4: 8192 x 8192 row-sharded (64 M
particles in all, strong scaling), counters reduced every 16 steps (and, reported beside it, every
step).  c5: 16384 x 16384 packed fp16 state row-sharded, 16-step fused groups.  Flow replicated; no
data-path collective; the statistics counters are reduced by small RCCL all-reduces.
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

N = 4096                        # C3: particles per rank = N*N
FLOW_W, FLOW_H = 1920, 1080
BYTES_PER_PARTICLE_STEP = 32    # 16 B state read + 16 B written; 8 + 8 with packed state
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK = 256 * 4 * 2.4e9 / 2     # wave64 VALU instructions per second: 2 cycles each on a SIMD-32
MAX_FUSED = 32                  # th::kMaxFusedSteps
PREROLL_MS = 60.0

CONFIGS = {
    # name: (width, global height as a function of world, rows per rank, state, steps per group, scaling)
    "c3": dict(width=N, rows=lambda w: N, gheight=lambda w: N * w, state="f32", group=32, scaling="weak",
               label="C3: 4096x4096 %s state (16.8M particles) per GPU"),
    "c4": dict(width=8192, rows=lambda w: 8192 // w, gheight=lambda w: 8192, state="f32", group=16, scaling="strong",
               label="C4: 8192x8192 %s state (67.1M particles) row-sharded over the GPUs"),
    "c5": dict(width=16384, rows=lambda w: 16384 // w, gheight=lambda w: 16384, state="f16", group=16, scaling="strong",
               label="C5: 16384x16384 %s state (268M particles) row-sharded over the GPUs"),
}


def synth_rows(width, rows, seed):
    rng = np.random.default_rng(seed)
    st = np.empty((rows, width, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (rows, width, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (rows, width, 2))
    return st


def synth_state(rank):
    return synth_rows(N, N, 12345 + rank)


def synth_frames():
    """frame0 = seeded band-limited pattern, frame1 = frame0 translated by (1.5, 0.7) px."""
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


# ---- rocprofv3 PMC child passes ----------------------------------------------------------------------------
PMC_PASSES = (("FETCH_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"), ("WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))


def kernel_class(name):
    """Which of the measured launches a kernel-trace row belongs to (template arguments: <FAST, NOISE, ...>)."""
    for base in ("logic_fused_packed_kernel", "logic_fused_kernel", "logic_packed_kernel", "logic_sorted_kernel", "logic_kernel"):
        if "th::" + base + "<" in name:
            args = name.split(base + "<", 1)[1].split(",")
            noise = len(args) > 1 and args[1].strip().startswith("true")
            fused = "fused" in base
            return ("fused" if fused else "single") + ("" if noise else "_flow_only")
    return None


def measure_pmc(extra_args, launch_len):
    """PMC counters of the integrator launches from rocprofv3, as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE
    and WRITE_SIZE in separate --pmc passes of the same workload (short child runs of this script with launches of
    `launch_len` steps like the timed region), FETCH_SIZE doubled when turned into bytes (gfx950 tallies the 128-B
    requests of a wide coalesced stream at 64 B), both in KiB.  Runs before this process touches the GPU.
    Returns {class: {counter: mean per launch}} and a note."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {}, "rocprofv3 not found"
    out_all = {}
    notes = []
    for group in PMC_PASSES:
        out = tempfile.mkdtemp(prefix="th_pmc_", dir="/tmp")
        cmd = [prof, "--pmc"] + list(group) + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", str(launch_len), "--no-cpu", "--no-traffic"] + extra_args
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=300, check=True)
            dur = {}
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            per_dispatch = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    cls = kernel_class(row.get("Kernel_Name", ""))
                    if cls:
                        per_dispatch.setdefault((cls, row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
            for (cls, did), cs in per_dispatch.items():
                d = out_all.setdefault(cls, {})
                for k, v in cs.items():
                    d.setdefault(k, []).append(v)
                if cs.get("GRBM_GUI_ACTIVE", 0) > 0:
                    cyc = cs["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
                    if dur.get(did, 0) > 0:
                        d.setdefault("clock_ghz", []).append(cyc / dur[did])          # cycles per ns
                        d.setdefault("profiled_launch_ms", []).append(dur[did] * 1e-6)
                    if "SQ_INSTS_VALU" in cs:      # issue slots used: 2 cycles per wave64 instruction on each of 1024 SIMDs
                        d.setdefault("valu_issue_utilization", []).append(cs["SQ_INSTS_VALU"] * 2.0 / (cyc * 1024.0))
        except (subprocess.SubprocessError, OSError) as e:
            notes.append("pass %s failed: %s" % ("+".join(group), type(e).__name__))
        finally:
            shutil.rmtree(out, ignore_errors=True)
    res = {cls: {k: sum(v) / len(v) for k, v in cs.items()} for cls, cs in out_all.items()}
    note = "rocprofv3 PMC, child runs with %d-step launches; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB" % launch_len
    if notes:
        note += "; " + "; ".join(notes)
    return res, note


def pmc_bytes(c):
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS), help="BASELINE.json config (default c3: the metric's)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child passes")
    ap.add_argument("--no-frame-loop", action="store_true", help="skip the step() + draw() frame-loop leg")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)   # PMC child: launches of this length only
    ap.add_argument("--force-dist", action="store_true", help="init RCCL even at world size 1 (path check)")
    ap.add_argument("--flow-size", default=None, help="experiment: WxH of the flow/view instead of 1920x1080")
    ap.add_argument("--state", default=None, choices=["f32", "f16"], help="state ring storage (f16 = packed 8 B/particle); default: the config's")
    ap.add_argument("--in-view", action="store_true", help="experiment: keep every particle inside the view (|y*viewSize.y| < 1)")
    ap.add_argument("--flow-only", action="store_true", help="noiseWeight = 0 (preset 'Flow Only') in the timed region")
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
    if rank != 0:           # one JSON line on the job's stdout: the other ranks' (and their libraries') go nowhere
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    cfg = CONFIGS[args.config]
    state_fmt = args.state or cfg["state"]
    group = cfg["group"]                      # steps per fused launch and per statistics reduction
    width, rows, gheight = cfg["width"], cfg["rows"](world), cfg["gheight"](world)
    particles_rank = width * rows
    launch_len = args.pmc_child or min(group, args.steps)

    pmc, pmc_note = {}, "not measured"
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    if rank == 0 and world == 1 and not args.no_traffic and not args.force_dist and not under_profiler and not args.pmc_child:
        extra = ["--mode", args.mode, "--config", args.config, "--state", state_fmt] + \
                (["--in-view"] if args.in_view else []) + (["--flow-size", args.flow_size] if args.flow_size else [])
        pmc, pmc_note = measure_pmc(extra, launch_len)

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
                row0=shard_rows(gheight, world, rank)[0], rows=rows, globalHeight=gheight,
                stateFormat=ta.TH_STATE_F16 if state_fmt == "f16" else ta.TH_STATE_F32)
    t = ta.Tendrils(View(FLOW_W, FLOW_H), opts)
    t.resize()                       # viewRes 1920x1080 -> viewSize [1, 1.7778]; flow.shape = viewRes
    t.setup(width)
    ctx = t.particles._ctx
    band = 1024                      # generated and uploaded in row bands (bounded host memory at C5)
    full = synth_state(rank) if args.config == "c3" else None
    for r0 in range(0, rows, band):
        r1 = min(rows, r0 + band)
        st = full[r0:r1] if full is not None else synth_rows(width, r1 - r0, 12345 + rank * 1000003 + r0)
        if args.in_view:
            st = st.copy()
            st[..., 1] *= np.float32(0.56)
        _capi.call("th_upload_state", ctx, -1, np.ascontiguousarray(st).ctypes.data_as(_capi._fp), 0, r0, width, r1 - r0)
    full = st = None

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

    def run(k_steps, every=None, refresh=True):
        # the step loop runs as fused launches (Tendrils.step_n -> th_step_n), `every` steps each; between
        # launches: statistics (+ their RCCL reduction) and the optical-flow refresh (every full group)
        every = every or group
        done = 0
        while done < k_steps:
            n = min(every, k_steps - done)
            t.step_n(n)
            done += n
            if n == every:
                stats_tick()
                if of is not None and refresh and done % group == 0:      # keep the field alive: re-stamp it from the frame pair (blended)
                    of.update(dict(speedLimit=t.state["speedLimit"], time=t.timer.time, viewSize=t.viewSize))
                    of.render()

    def run_kernel_only(k_steps, length):
        done = 0
        while done < k_steps:
            n = min(length, k_steps - done)
            t.step_n(n)
            done += n

    def timed_kernels(fn):
        """mean launch duration (HIP event pair around every integrator launch on the context's stream)"""
        ms, n = C.c_float(), C.c_int32()
        _capi.call("th_kernel_timing", ctx, 1)
        fn()
        _capi.call("th_kernel_timing_read", ctx, C.byref(ms), C.byref(n))
        _capi.call("th_kernel_timing", ctx, 0)
        return ms.value, n.value

    if args.flow_only:
        t.state["noiseWeight"] = 0

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
        t.dispose()
        return

    run(args.warmup)
    # clock pre-roll: the launches of the timed region, untimed, until >= PREROLL_MS have run on the device
    sync_all()
    p0 = time.perf_counter()
    run_kernel_only(launch_len, launch_len)
    sync_all()
    est = max(time.perf_counter() - p0, 1e-4)
    pre_launches = int(min(max(PREROLL_MS * 1e-3 / est, 1), 4096))
    p0 = time.perf_counter()
    run_kernel_only(pre_launches * launch_len, launch_len)
    sync_all()
    preroll_ms = (time.perf_counter() - p0) * 1e3

    t0 = time.perf_counter()
    run(args.steps)
    sync_all()
    wall = time.perf_counter() - t0

    # c4: the same K steps with the counters reduced after EVERY step (BASELINE.md config 4 names both cadences)
    wall_every_step = None
    if args.config == "c4":
        sync_all()
        t1 = time.perf_counter()
        run(args.steps, every=1, refresh=False)
        sync_all()
        wall_every_step = time.perf_counter() - t1

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

    if pending:
        with torch.cuda.stream(ext_stream):
            for w in pending:
                w.wait()
    stats = t.particles.stats(t.state["speedLimit"])

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

    if dist is not None:
        tmax = torch.tensor([wall, k_ms / 1e3, s_ms / 1e3, wall_every_step or 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, kern_s, single_s, wes = (float(v) for v in tmax)
        wall_every_step = wes if wall_every_step is not None else None
        stats = reduce_counters(dist, stats, device="cuda")
    else:
        kern_s, single_s = k_ms / 1e3, s_ms / 1e3

    particles = particles_rank * world
    packed = state_fmt == "f16"
    bytes_per_step = BYTES_PER_PARTICLE_STEP // (2 if packed else 1)
    value = particles * args.steps / wall
    launches = max(int(k_n), 1)
    steps_per_launch = args.steps / launches
    fused = steps_per_launch > 1

    def rl(launch_s, steps_in_launch, counters):
        """roofline entries of one kind of launch"""
        alg = bytes_per_step * particles_rank * steps_in_launch
        e = {"avg_launch_ms": launch_s * 1e3, "steps_per_launch": steps_in_launch,
             "ms_per_step": launch_s * 1e3 / steps_in_launch,
             "achieved": alg / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / launch_s / 1e9 / HBM_PEAK_GBS,
             "algorithmic_bytes_per_launch": alg}
        tb = pmc_bytes(counters)
        e["traffic"] = tb
        if tb is not None:
            e["hbm_physical"] = {"achieved": tb / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": tb / launch_s / 1e9 / HBM_PEAK_GBS, "bytes_over_algorithmic": tb / alg}
        if counters and "SQ_INSTS_VALU" in counters:
            v = counters["SQ_INSTS_VALU"]
            e["valu"] = {"wave_insts_per_launch": v, "per_wave_step": v / (particles_rank / 64.0 * steps_in_launch),
                         "achieved": v / launch_s, "peak": VALU_PEAK, "unit": "wave-instr/s", "frac": v / launch_s / VALU_PEAK}
            if "clock_ghz" in counters:      # under the profiler (launches run a few % slower there)
                e["valu"]["clock_ghz"] = counters["clock_ghz"]
                e["valu"]["profiled_launch_ms"] = counters.get("profiled_launch_ms")
                e["valu"]["issue_utilization_at_held_clock"] = counters.get("valu_issue_utilization")
                e["valu"]["note"] = "peak = 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 instruction; clock_ghz = GRBM_GUI_ACTIVE / 8 / launch " \
                                    "duration and issue_utilization = 2 x SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), both per " \
                                    "dispatch in the PMC child run: the clock the chip held under this load and the share of its issue " \
                                    "slots the launch used at that clock"
        if counters and "SQ_LDS_IDX_ACTIVE" in counters and counters["SQ_LDS_IDX_ACTIVE"] > 0:
            e["lds"] = {"bank_conflict_share": counters.get("SQ_LDS_BANK_CONFLICT", 0.0) / counters["SQ_LDS_IDX_ACTIVE"]}
        return e

    main_cls = "fused" if fused else "single"
    if args.flow_only:
        main_cls += "_flow_only"
    head = rl(kern_s, steps_per_launch, pmc.get(main_cls))
    kernel_name = ("logic_fused_packed_kernel" if packed else "logic_fused_kernel") if fused else \
        ("logic_packed_kernel" if packed else "logic_kernel")
    roofline = {"bound": "valu" if (fused and not args.flow_only) else "hbm", "kernel": kernel_name, "launches": launches,
                "uniform_set": "flow only (noiseWeight = 0)" if args.flow_only else "default (simplex noise on)",
                "achieved_is": "algorithmic bytes (SURVEY.md 8d: %d B per particle-step) / mean launch duration - an equivalent "
                               "single-step bandwidth; the fused launch streams 48/n B per particle-step (hbm_physical) and is "
                               "bound by VALU issue (valu)" % bytes_per_step,
                "pmc_note": pmc_note}
    roofline.update(head)
    single = rl(single_s, 1, pmc.get("single"))
    single["kernel"] = "logic_packed_kernel" if packed else "logic_kernel over tile-sorted slots (gathered taps; every 64th launch re-sorts through logic_sorted_kernel)"
    single["bound"] = "hbm"
    roofline["single_step_kernel"] = single
    roofline["other_mode"] = {"mode": other_mode, "avg_launch_ms": o_ms, "steps_per_launch": launch_len,
                              "achieved": bytes_per_step * particles_rank * launch_len / max(o_ms, 1e-9) / 1e6}
    if f_n:
        fo = rl(f_ms / 1e3, launch_len, pmc.get("fused_flow_only" if launch_len > 1 else "single_flow_only"))
        fo["uniform_set"] = "flow only (noiseWeight = 0)"
        fo["bound"] = "hbm"
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
        "config": {"workload": (cfg["label"] % storage) + ", flow %dx%d from " % (FLOW_W, FLOW_H)
                               + flow_source + ", reference default uniforms"
                               + (" with noiseWeight=0 (flow-only)" if args.flow_only else " (simplex noise on)")
                               + ", 60 Hz fixed timer; step loop as fused launches of <= %d steps, statistics "
                                 "(+ RCCL reduction) every %d steps" % (launch_len, group),
                   "mode": args.mode, "state_storage": state_fmt, "particles_per_gpu": particles_rank,
                   "parallelism": "row-band shard x%d, flow replicated" % world},
        "roofline": roofline,
        "counters": stats,
    }
    if wall_every_step is not None:
        line["counters_every_step"] = {"value": particles * args.steps / wall_every_step,
                                       "ms_per_step": wall_every_step / args.steps * 1e3,
                                       "note": "same K steps, one launch and one counter reduction per step"}

    # (the side legs must not cost the line: whatever goes wrong in them is reported in their place)
    if world == 1 and args.config == "c3" and not args.no_frame_loop and not args.pmc_child:
        try:
            line["frame_loop"] = frame_loop(t, ctx, synth_state(rank))
        except Exception as e:            # noqa: BLE001
            line["frame_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            line["cpu_baseline"] = cpu_baseline(t, width, min(rows, N))
        except Exception as e:            # noqa: BLE001
            line["cpu_baseline"] = {"value": None, "unit": "particle-steps/s", "cores": 0, "kind": "port",
                                    "sample": "failed: %s: %s" % (type(e).__name__, e)}
    t.dispose()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio (flushed at exit when stdout is a pipe): push it out first, so
        # that the JSON line is the last thing on stdout
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


def frame_loop(t, ctx, state, frames=20):
    """SURVEY.md 8f-1/8f-2 beside the headline: the reference's frame loop - timer.tick(), step(), draw() - on the same
    particles: one single-step launch, the flow pass of draw() (the particle lines blended into the flow field in GL
    primitive order: rasterise, scan, emit, stable sort by texel, gather, blend) and the view pass (the same lines into
    the RGBA8 view buffer), each timed with a HIP event pair on the context's stream."""
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
    t.renderView = keep
    lines, f = state.shape[0] * state.shape[1], float(np.mean(frags))
    # per line: two state texels in the rasterising and in the emitting pass (64 B), count / offset / scan (24 B);
    # per fragment: key + varying written (20 B), three radix passes over key + position (48 B), gather (36 B), blend (20 B)
    alg = lines * 88.0 + f * 124.0
    d = float(np.mean(flow_ms))
    return {"frames": frames, "step_ms": float(np.mean(step_ms)), "draw_flow_ms": d, "draw_view_ms": float(np.mean(view_ms)), "draw_both_ms": float(np.mean(both_ms)),
            "fragments_per_draw": f, "frames_per_s": 1e3 / (float(np.mean(step_ms)) + d),
            "roofline": {"bound": "hbm", "kernel": "flow pass of draw() (9 kernels + 3 sort passes)", "achieved": alg / d / 1e6,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / d / 1e6 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_draw": alg,
                         "achieved_is": "88 B per line + 124 B per fragment (DESIGN.md 3.4) / mean duration of the pass"},
            "note": "timer.tick(); step(); draw(): one single-step launch + the flow pass; the view pass timed separately "
                    "(th_view_draw after th_flow_deposit: it reuses the flow pass's rasterisation and sort), and both passes in "
                    "one call (th_draw, what Tendrils.draw() runs with renderView) over 5 more frames"}


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


if __name__ == "__main__":
    main()
