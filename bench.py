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
barrier + synchronize; max over ranks per repetition; `value` / `ms_per_step` are the MEDIAN repetition
(`repetitions` carries all of them and their spread).  Every timed repetition contains the path's
collective: after every fused launch the statistics pass and - world > 1 - the library's own RCCL
all-reduce of the counter block (th_stats_allreduce, on the context's stream); `rccl` reports the ranks
the reduction saw.

Roofline block.  The step loop runs as fused launches (th_step_n: <= 32 steps of a
particle back to back in registers), which stream 48/n bytes per particle-step instead of 32 and are
bound by VALU issue, not by HBM.  The line therefore carries
  * roofline.achieved/peak/equivalent_frac : the SURVEY.md 8d figure - ALGORITHMIC bytes (32 B x
    particles x steps of one launch) / mean launch duration, against 8 TB/s.  An equivalent
    single-step bandwidth, not a physical one (for a register-resident fused launch it can exceed
    the peak); roofline.frac is the fraction of the bound the entry NAMES (`bound`): VALU issue for
    the fused noise-on launch, the larger of physical HBM and VALU for the fused flow-only one,
    HBM (algorithmic bytes, which a single-step launch really streams) for one step per launch -
    never above 1;
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
(weak scaling; the N = 1 line is the single-GPU bench).  c4: 8192 x 8192 row-sharded (64 M particles in
all, strong scaling), counters reduced every 16 steps (and, reported beside it, every step) - also run as
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
import socket
import subprocess
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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(gpus, argv):
    """`bench.py --gpus N` with no launcher around it: start the N ranks as a child job - from a process that has not
    touched the GPU (nothing here imports torch) - relay rank 0's JSON line, return the child's status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what this pool's driver supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in child.stdout:
        text = out.strip()
        if text.startswith("{") and '"metric"' in text:
            line = text
        elif text:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc else (0 if line is not None else 1)


def median(xs):
    ys = sorted(xs)
    return ys[len(ys) // 2] if len(ys) % 2 else 0.5 * (ys[len(ys) // 2 - 1] + ys[len(ys) // 2])


def repetition_block(walls, steps):
    ms = [w / steps * 1e3 for w in walls]
    mid = median(ms)
    return {"n": len(ms), "ms_per_step": ms, "median": mid, "min": min(ms), "max": max(ms),
            "spread": (max(ms) - min(ms)) / mid if mid > 0 else None,
            "note": "each repetition: barrier + synchronize, K steps, barrier + synchronize; max over ranks; value = median"}


class Job:
    """One rank's share of a configuration: the Tendrils object with its synthetic state and flow, the step loop of the
    timed region (fused launches + statistics + the counter all-reduce + optical-flow refresh) and its timing."""

    def __init__(self, args, config, rank, local_rank, world, dist, launch_len=None):
        import tendrils_amd as ta
        from tendrils_amd import _capi
        from tendrils_amd.sharding import comm_init, comm_query, shard_rows
        from tendrils_amd.tendrils import View
        self.ta, self.capi, self.dist, self.world, self.rank = ta, _capi, dist, world, rank
        cfg = CONFIGS[config]
        self.cfg, self.config = cfg, config
        self.state_fmt = args.state or cfg["state"]
        self.group = cfg["group"]                    # steps per fused launch and per statistics reduction
        self.width, self.rows, self.gheight = cfg["width"], cfg["rows"](world), cfg["gheight"](world)
        self.particles_rank = self.width * self.rows
        self.launch_len = launch_len or min(self.group, args.steps)
        opts = ta.defaults()
        opts.update(device=local_rank, mode=ta.TH_MODE_FAST if args.mode == "fast" else ta.TH_MODE_EXACT,
                    row0=shard_rows(self.gheight, world, rank)[0], rows=self.rows, globalHeight=self.gheight,
                    stateFormat=ta.TH_STATE_F16 if self.state_fmt == "f16" else ta.TH_STATE_F32)
        t = self.t = ta.Tendrils(View(FLOW_W, FLOW_H), opts)
        t.resize()                       # viewRes 1920x1080 -> viewSize [1, 1.7778]; flow.shape = viewRes
        t.setup(self.width)
        ctx = self.ctx = t.particles._ctx
        band = 1024                      # generated and uploaded in row bands (bounded host memory at C5)
        full = synth_state(rank) if config == "c3" else None
        for r0 in range(0, self.rows, band):
            r1 = min(self.rows, r0 + band)
            st = full[r0:r1] if full is not None else synth_rows(self.width, r1 - r0, 12345 + rank * 1000003 + r0)
            if args.in_view:
                st = st.copy()
                st[..., 1] *= np.float32(0.56)
            _capi.call("th_upload_state", ctx, -1, np.ascontiguousarray(st).ctypes.data_as(_capi._fp), 0, r0, self.width, r1 - r0)
        full = st = None

        # flow field: optical-flow pass over the synthetic frame pair (C3), else a seeded field
        self.time0 = 1000.0
        self.flow_source = "optical-flow(synthetic 1080p frame pair)"
        self.of = None
        try:
            from tendrils_amd.optical_flow import OpticalFlow
            f0, f1 = synth_frames()
            of = OpticalFlow(t, uniforms=dict(speed=0.08, offset=0.1, scaleUV=[-1, -1]))   # src/demo.main.js:526-530
            of.resize([FLOW_W, FLOW_H])
            of.set_pixels(f0)
            of.step()
            of.set_pixels(f1)
            of.update(dict(speedLimit=t.state["speedLimit"], time=self.time0, viewSize=t.viewSize))
            of.render()
            self.of = of
        except (ImportError, ta.TendrilsHipError):
            self.flow_source = "synthetic divergence-free field (optical-flow pass unavailable)"
            t.flow.set_pixels(synth_flow(self.time0))
        t.timer.time = self.time0
        # the job's communicator inside the library: the counter all-reduce of the timed region is th_stats_allreduce
        self.comm, self.comm_fallback = None, None
        if dist is not None:
            # (should the library's own communicator not come up - librccl not loadable beside torch's, say - on any rank,
            # every rank falls back to reducing the counter block through torch.distributed, and the line says so: a
            # scaling run is not lost to it)
            import torch
            why = "--no-library-comm" if args.no_library_comm else ""
            try:
                if not why:
                    comm_init(ctx, dist)
                    self.comm = comm_query(ctx)
            except ta.TendrilsHipError as e:
                why = str(e)
            ok = torch.tensor([0 if why else 1], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if self.comm is not None:
                    _capi.call("th_comm_destroy", ctx)
                self.comm = None
                self.comm_fallback = why or "another rank could not join the library's communicator"
                self._counters_view = None
        self.reductions = 0
        if args.flow_only:
            t.state["noiseWeight"] = 0

    def sync_all(self):
        import torch
        self.t.particles.sync()
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize()

    def stats_tick(self):
        """statistics of buffers[0] and - world > 1 - their reduction over the ranks, both enqueued on the context's
        stream (the library's RCCL all-reduce: no host sync, no second stream)"""
        dev = C.c_void_p()
        self.capi.call("th_stats_async", self.ctx, C.c_float(self.t.state["speedLimit"]), C.byref(dev))
        if self.comm is not None:
            self.capi.call("th_stats_allreduce", self.ctx)
            self.reductions += 1
        elif self.comm_fallback:
            self.fallback_reduce(dev.value)
            self.reductions += 1

    def fallback_reduce(self, dev_ptr):
        """(fallback) the th_counters block - five u64 and a f64 summed, a f64 maximised - reduced in place through
        torch.distributed on the context's stream"""
        import torch
        from tendrils_amd.sharding import device_view
        if self._counters_view is None:
            sp = C.c_void_p()
            self.capi.call("th_stream", self.ctx, C.byref(sp))
            self._ext = torch.cuda.ExternalStream(sp.value)
            self._counters_view = (device_view(dev_ptr, (5,), "<i8"), device_view(dev_ptr + 40, (1,), "<f8"), device_view(dev_ptr + 48, (1,), "<f8"))
        counts, total, peak = self._counters_view
        with torch.cuda.stream(self._ext):
            self.dist.all_reduce(counts)
            self.dist.all_reduce(total)
            self.dist.all_reduce(peak, op=self.dist.ReduceOp.MAX)

    def run(self, k_steps, every=None, refresh=True):
        # the step loop runs as fused launches (Tendrils.step_n -> th_step_n), `every` steps each; after EVERY launch
        # (a trailing partial one included): statistics + their reduction over the ranks; after every full group the
        # optical-flow refresh
        t, of = self.t, self.of
        every = min(every or self.group, max(k_steps, 1))
        done = 0
        while done < k_steps:
            n = min(every, k_steps - done)
            t.step_n(n)
            done += n
            self.stats_tick()
            if of is not None and refresh and done % self.group == 0:      # keep the field alive: re-stamp it from the frame pair (blended)
                of.update(dict(speedLimit=t.state["speedLimit"], time=t.timer.time, viewSize=t.viewSize))
                of.render()

    def run_kernel_only(self, k_steps, length):
        done = 0
        while done < k_steps:
            n = min(length, k_steps - done)
            self.t.step_n(n)
            done += n

    def timed_kernels(self, fn):
        """mean launch duration (HIP event pair around every integrator launch on the context's stream)"""
        ms, n = C.c_float(), C.c_int32()
        self.capi.call("th_kernel_timing", self.ctx, 1)
        fn()
        self.capi.call("th_kernel_timing_read", self.ctx, C.byref(ms), C.byref(n))
        self.capi.call("th_kernel_timing", self.ctx, 0)
        return ms.value, n.value

    def preroll(self):
        """clock pre-roll: the launches of the timed region, untimed, until >= PREROLL_MS have run on the device"""
        self.sync_all()
        p0 = time.perf_counter()
        self.run_kernel_only(self.launch_len, self.launch_len)
        self.sync_all()
        est = max(time.perf_counter() - p0, 1e-4)
        pre_launches = int(min(max(PREROLL_MS * 1e-3 / est, 1), 4096))
        p0 = time.perf_counter()
        self.run_kernel_only(pre_launches * self.launch_len, self.launch_len)
        self.sync_all()
        return (time.perf_counter() - p0) * 1e3

    def timed_region(self, steps, reps, **kw):
        """`reps` x [barrier + synchronize, `steps` steps, barrier + synchronize] -> wall seconds of each (this rank)"""
        walls = []
        for _ in range(reps):
            self.sync_all()
            t0 = time.perf_counter()
            self.run(steps, **kw)
            self.sync_all()
            walls.append(time.perf_counter() - t0)
        return walls

    def max_over_ranks(self, values):
        if self.dist is None:
            return [float(v) for v in values]
        import torch
        v = torch.tensor(list(values), dtype=torch.float64, device="cuda")
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in v]

    def global_stats(self):
        """th_stats_global: the job's counters (local pass + the library's all-reduce + download)"""
        if self.comm_fallback:
            import torch
            dev = C.c_void_p()
            self.capi.call("th_stats_async", self.ctx, C.c_float(self.t.state["speedLimit"]), C.byref(dev))
            self.fallback_reduce(dev.value)
            self.t.particles.sync()
            torch.cuda.synchronize()
            counts, total, peak = self._counters_view
            names = [k for k, _ in self.capi.Counters._fields_]
            vals = [int(v) for v in counts.cpu().tolist()] + [float(total.cpu()[0]), float(peak.cpu()[0])]
            return dict(zip(names, vals))
        c = self.capi.Counters()
        self.capi.call("th_stats_global", self.ctx, C.c_float(self.t.state["speedLimit"]), C.byref(c))
        return {k: getattr(c, k) for k, _ in self.capi.Counters._fields_}

    def rccl_block(self, stats, reductions_per_rep):
        seen = stats["particles"] / float(self.particles_rank)
        b = {"world": self.world, "nranks_seen": seen, "reductions_per_timed_repetition": reductions_per_rep,
             "note": "nranks_seen = the all-reduced `particles` counter / this rank's particles: the ranks whose blocks the "
                     "library's RCCL all-reduce (th_stats_allreduce, on the context's stream) added up; at world 1 the "
                     "context holds no communicator and the local block is the global one"}
        if self.comm is not None:
            b.update(version=self.comm["rccl_version"], in_library=True, rank=self.comm["rank"])
        elif self.comm_fallback:
            b.update(in_library=False, fallback="torch.distributed all-reduce of the counter block: " + self.comm_fallback)
        return b

    def dispose(self):
        self.t.dispose()


def roofline_entry(job, launch_s, steps_in_launch, counters, bytes_per_step):
    """roofline entries of one kind of launch"""
    alg = bytes_per_step * job.particles_rank * steps_in_launch
    eq = alg / launch_s / 1e9 / HBM_PEAK_GBS
    e = {"avg_launch_ms": launch_s * 1e3, "steps_per_launch": steps_in_launch,
         "ms_per_step": launch_s * 1e3 / steps_in_launch,
         "achieved": alg / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "equivalent_frac": eq,
         "algorithmic_bytes_per_launch": alg}
    tb = pmc_bytes(counters)
    e["traffic"] = tb
    if tb is not None:
        e["hbm_physical"] = {"achieved": tb / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": tb / launch_s / 1e9 / HBM_PEAK_GBS, "bytes_over_algorithmic": tb / alg}
    if counters and "SQ_INSTS_VALU" in counters:
        v = counters["SQ_INSTS_VALU"]
        e["valu"] = {"wave_insts_per_launch": v, "per_wave_step": v / (job.particles_rank / 64.0 * steps_in_launch),
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


def bind(e, fused):
    """`bound` and `frac` of an entry: the fraction of the bound it names, never the equivalent bandwidth of a
    register-resident launch.  One step per launch streams its algorithmic bytes: HBM, frac = algorithmic / peak.
    A fused launch is bound by whichever of VALU issue and physical HBM traffic it uses more of (PMC child runs);
    without counters the bound is not known and frac stays null."""
    if not fused:
        e["bound"], e["frac"] = "hbm", e["equivalent_frac"]
        e["frac_is"] = "algorithmic bytes / launch duration / HBM peak (a single-step launch streams them)"
        return e
    v = (e.get("valu") or {}).get("frac")
    h = (e.get("hbm_physical") or {}).get("frac")
    if v is None and h is None:
        e["bound"], e["frac"] = "valu", None
        e["frac_is"] = "unknown: the PMC child runs gave no counters (equivalent_frac is the SURVEY.md 8d figure)"
    elif h is None or (v is not None and v >= h):
        e["bound"], e["frac"] = "valu", v
        e["frac_is"] = "valu.frac: wave64 VALU instructions per second / the chip's issue peak at 2.4 GHz"
    else:
        e["bound"], e["frac"] = "hbm", h
        e["frac_is"] = "hbm_physical.frac: PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) / launch duration / HBM peak"
    return e


def c4_leg(args, rank, local_rank, world, dist):
    """BASELINE.json config 4 beside the metric's own configuration: 8192 x 8192 particles row-sharded over the ranks
    (strong scaling: 64 M particles in all, whatever N), counters reduced after every 16-step launch."""
    job = Job(args, "c4", rank, local_rank, world, dist)
    job.run(args.warmup)
    job.preroll()
    job.reductions = 0
    walls = job.timed_region(args.steps, max(args.reps // 2, 3))
    reductions = job.reductions // max(args.reps // 2, 3)
    walls = job.max_over_ranks(walls)
    stats = job.global_stats()
    particles = job.particles_rank * world
    mid = median(walls)
    out = {"value": particles * args.steps / mid, "unit": "particle-steps/s", "ms_per_step": mid / args.steps * 1e3,
           "scaling": "strong", "n_gpus": world, "steps": args.steps, "particles": particles,
           "particles_per_gpu": job.particles_rank, "repetitions": repetition_block(walls, args.steps),
           "rccl": job.rccl_block(stats, reductions),
           "workload": (job.cfg["label"] % "RGBA32F") + ", same flow and uniforms as the headline, fused launches of <= %d steps, "
                       "statistics + counter all-reduce after every launch" % job.launch_len,
           "note": "K = %d steps run as %s: a short trailing launch streams 48 / n bytes per particle-step like any other and costs "
                   "its own statistics" % (args.steps, " + ".join(str(min(job.launch_len, args.steps - d)) for d in range(0, args.steps, job.launch_len)) + " step launches")}
    job.dispose()
    return out


def dry_run(args, rank, world):
    """The launcher path, the rank plumbing, the timed-region protocol and the collective on CPU: gloo ranks stepping a
    small row band each with the CPU restatement (test infrastructure - this measures nothing and says so)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from tendrils_amd.sharding import reduce_counters, shard_rows
    n = 48
    gheight = n * world
    row0, rows = shard_rows(gheight, world, rank)
    band = synth_rows(n, rows, 12345 + rank)
    fl = np.zeros((27, 48, 4), np.float32)
    fl[..., :2] = np.random.default_rng(5).uniform(-.01, .01, (27, 48, 2))
    fl[..., 2] = 990.0
    group = min(4, max(args.steps, 1))
    tm = {"time": 1000.0}

    def counters(b):
        live = (b[..., 0] != -1e6) | (b[..., 1] != -1e6)
        sp = np.hypot(b[..., 2].astype(np.float64), b[..., 3].astype(np.float64))[live]
        return dict(particles=b.shape[0] * n, live=int(live.sum()), nan=int(np.isnan(b).any(-1).sum()),
                    capped=int((sp >= 0.01 * (1 - 2 ** -20)).sum()), respawned=0, sum_speed=float(np.nansum(sp)),
                    max_speed=float(np.nanmax(sp)) if sp.size else 0.0)

    state = {"band": band, "red": None, "reductions": 0}

    def run(k):
        done = 0
        while done < k:
            m = min(group, k - done)
            for _ in range(m):
                tm["time"] += 1000.0 / 60.0
                u = O.logic_uniforms(n, gheight, tm["time"], 1000.0 / 60.0, view_size=(1, 48 / 27))
                state["band"] = O.logic_step(u, state["band"], fl, y0=row0)
            done += m
            state["red"] = reduce_counters(dist, counters(state["band"]))
            state["reductions"] += 1

    run(args.warmup)
    walls = []
    for _ in range(args.reps):
        dist.barrier()
        t0 = time.perf_counter()
        state["reductions"] = 0
        run(args.steps)
        dist.barrier()
        walls.append(time.perf_counter() - t0)
    v = torch.tensor(walls, dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    walls = [float(x) for x in v]
    mid = median(walls)
    line = {"metric": "particle-steps/sec (dry run: CPU restatement over gloo, plumbing only)", "dry_run": True,
            "value": n * gheight * args.steps / mid, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": mid / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dry run: %d x %d particles per rank, CPU restatement, gloo" % (n, rows)},
            "repetitions": repetition_block(walls, args.steps),
            "rccl": {"world": world, "nranks_seen": state["red"]["particles"] / float(rows * n), "backend": "gloo",
                     "reductions_per_timed_repetition": state["reductions"]},
            "counters": state["red"]}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


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
    ap.add_argument("--no-c4", action="store_true", help="skip the config-4 leg (8192 x 8192 row-sharded, strong scaling) of a c3 run")
    ap.add_argument("--dry-run", action="store_true", help="CPU plumbing check: gloo ranks stepping the CPU restatement")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)   # PMC child: launches of this length only
    ap.add_argument("--force-dist", action="store_true", help="init RCCL even at world size 1 (path check)")
    ap.add_argument("--no-library-comm", action="store_true", help="reduce the counters through torch.distributed instead of the library's own communicator (what the bench falls back to when th_comm_init fails on any rank)")
    ap.add_argument("--flow-size", default=None, help="experiment: WxH of the flow/view instead of 1920x1080")
    ap.add_argument("--state", default=None, choices=["f32", "f16"], help="state ring storage (f16 = packed 8 B/particle); default: the config's")
    ap.add_argument("--in-view", action="store_true", help="experiment: keep every particle inside the view (|y*viewSize.y| < 1)")
    ap.add_argument("--flow-only", action="store_true", help="noiseWeight = 0 (preset 'Flow Only') in the timed region")
    args = ap.parse_args()
    args.reps = max(args.reps, 1)

    global FLOW_W, FLOW_H
    if args.flow_size:
        FLOW_W, FLOW_H = (int(v) for v in args.flow_size.lower().split("x"))
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
    state_fmt = args.state or cfg["state"]
    group = cfg["group"]
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

    job = Job(args, args.config, rank, local_rank, world, dist, launch_len=launch_len)
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

    run(args.warmup)
    preroll_ms = job.preroll()

    job.reductions = 0
    walls = job.timed_region(args.steps, args.reps)
    reductions_per_rep = job.reductions // args.reps

    # c4: the same K steps with the counters reduced after EVERY step (BASELINE.md config 4 names both cadences)
    walls_every_step = None
    if args.config == "c4":
        walls_every_step = job.timed_region(args.steps, max(args.reps // 2, 2), every=1, refresh=False)

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
                               "single-step bandwidth (equivalent_frac = achieved / peak); the fused launch streams 48/n B per "
                               "particle-step (hbm_physical); frac = achieved / peak (the resource the launch runs out of: `limited_by`)" % bytes_per_step,
                "pmc_note": pmc_note}
    if fused and head["equivalent_frac"] <= 1.0:
        # the contract's figure for the dominant kernel: algorithmic bytes / launch duration against the HBM peak.
        # What the launch actually runs out of is named beside it: a register-resident launch streams 48/n B per particle-step
        # and is limited by VALU issue (`limited_by`, `limited_by_frac`; never reported as more than 1)
        head["limited_by"], head["limited_by_frac"], head["limited_by_frac_is"] = head["bound"], head["frac"], head["frac_is"]
        head["bound"], head["frac"] = "hbm", head["equivalent_frac"]
        head["frac_is"] = "achieved / peak: algorithmic bytes (SURVEY.md 8d) per launch / mean launch duration / 8 TB/s"
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
        "config": {"workload": (cfg["label"] % storage) + ", flow %dx%d from " % (FLOW_W, FLOW_H)
                               + job.flow_source + ", reference default uniforms"
                               + (" with noiseWeight=0 (flow-only)" if args.flow_only else " (simplex noise on)")
                               + ", 60 Hz fixed timer; step loop as fused launches of <= %d steps, statistics "
                                 "(+ the library's RCCL all-reduce of the counters) after every launch" % launch_len,
                   "mode": args.mode, "state_storage": state_fmt, "particles_per_gpu": particles_rank,
                   "parallelism": "row-band shard x%d, flow replicated" % world},
        "roofline": roofline,
        "rccl": job.rccl_block(stats, reductions_per_rep),
        "counters": stats,
    }
    if walls_every_step is not None:
        wes = median(walls_every_step)
        line["counters_every_step"] = {"value": particles * args.steps / wes, "ms_per_step": wes / args.steps * 1e3,
                                       "note": "same K steps, one launch and one counter reduction per step"}

    # (the side legs must not cost the line: whatever goes wrong in them is reported in their place)
    if world == 1 and args.config == "c3" and not args.no_frame_loop:
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
    job.dispose()
    if under_profiler and args.config == "c3" and not args.no_c4:
        # (a kernel-trace of this command should average the headline's launches, not mix them with config 4's)
        line["c4"] = {"skipped": "under a profiler: run without it (or --config c4) for the config-4 leg"}
    elif args.config == "c3" and not args.no_c4 and not args.flow_size:
        # (every rank takes part; a failure on one rank would hang the others in a collective: the leg runs the same
        # code path as the headline, so what it can still fail on - memory - fails on every rank alike)
        try:
            line["c4"] = c4_leg(args, rank, local_rank, world, dist)
        except Exception as e:            # noqa: BLE001
            line["c4"] = {"error": "%s: %s" % (type(e).__name__, e)}
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
    texels = FLOW_W * FLOW_H
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


if __name__ == "__main__":
    main()
