"""One rank's share of a configuration and the timed region of bench.py."""
import ctypes as C
import time

import numpy as np

from . import workload as W
from .workload import CONFIGS, PREROLL_MS, synth_flow, synth_frames, synth_rows, synth_state


def median(xs):
    ys = sorted(xs)
    return ys[len(ys) // 2] if len(ys) % 2 else 0.5 * (ys[len(ys) // 2 - 1] + ys[len(ys) // 2])


def repetition_block(walls, steps):
    ms = [w / steps * 1e3 for w in walls]
    mid = median(ms)
    return {"n": len(ms), "ms_per_step": ms, "median": mid, "min": min(ms), "max": max(ms),
            "spread": (max(ms) - min(ms)) / mid if mid > 0 else None,
            "note": "each repetition: barrier + synchronize, K steps, barrier + synchronize; max over ranks; value = median"}


class NodeBarrier:
    """The barrier of the timed region's bracket.  The ranks of a bench run are the GPUs of ONE node (the contract), so they
    meet in shared memory: every rank writes the epoch it has reached into a cache line of its own and waits until nobody's is
    older - a few microseconds, where dist.barrier() on the RCCL backend launches an all-reduce and waits for it (29 us at world
    1 on this pool, inside EVERY repetition's wall time: 10 % of a 2 M-particle band's 20 steps).  One per process (the legs'
    jobs share it); dist.barrier() itself where shared memory cannot be had."""
    _one = None

    @classmethod
    def of(cls, dist, rank, world):
        if cls._one is None:
            cls._one = cls(dist, rank, world)
        return cls._one

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world, self.epoch, self.slots, self.shm = dist, rank, world, 0, None, None
        self.kind = "torch.distributed barrier"
        # (every rank takes every collective below whatever failed on it: a rank that skipped one would leave the others inside it)
        slots, ok, name = None, 1, [None]
        try:
            from multiprocessing import resource_tracker, shared_memory
        except Exception:                  # noqa: BLE001
            ok = 0
        if rank == 0 and ok:
            try:                           # (a fresh segment is zero-filled: epoch 0 everywhere)
                self.shm = shared_memory.SharedMemory(create=True, size=64 * world)
                name[0] = self.shm.name
            except Exception:              # noqa: BLE001
                name[0] = None
        try:
            dist.broadcast_object_list(name, src=0)
        except Exception:                  # noqa: BLE001
            name[0] = None
        if name[0] is None:
            ok = 0
        if ok:
            try:
                import os
                if os.environ.get("TH_BENCH_TEST_NOSHM") == str(rank):      # (tests: this rank cannot have the segment)
                    raise OSError("TH_BENCH_TEST_NOSHM")
                if rank != 0:
                    self.shm = shared_memory.SharedMemory(name=name[0])
                    try:                   # (the creator unlinks it; an attaching process's tracker must not)
                        resource_tracker.unregister(self.shm._name, "shared_memory")
                    except Exception:      # noqa: BLE001
                        pass
                slots = np.ndarray((world, 8), dtype=np.int64, buffer=self.shm.buf)
            except Exception:              # noqa: BLE001
                ok = 0
        try:                               # every rank or none
            import torch
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        except Exception:                  # noqa: BLE001
            ok = 0
        if ok:
            self.slots = slots
            self.kind = "shared-memory epoch barrier (the ranks of one node)"
            import atexit
            atexit.register(self.close)
        else:
            self.close()

    def __call__(self):
        if self.slots is None:
            self.dist.barrier()
            return
        self.epoch += 1
        self.slots[self.rank, 0] = self.epoch
        mine, col = self.epoch, self.slots[:, 0]
        while int(col.min()) < mine:
            pass

    def close(self):
        shm, self.slots, self.shm = self.shm, None, None
        if shm is not None:
            try:
                shm.close()
                if self.rank == 0:
                    shm.unlink()
            except Exception:              # noqa: BLE001
                pass


class Job:
    """One rank's share of a configuration: the Tendrils object with its synthetic state and flow, the step loop of the
    timed region (fused launches + statistics + the counter all-reduce + optical-flow refresh) and its timing."""

    def __init__(self, args, config, rank, local_rank, world, dist, launch_len=None, comm_guard=None):
        import tendrils_amd as ta
        from tendrils_amd import _capi
        from tendrils_amd.sharding import comm_init, comm_query, shard_rows
        from tendrils_amd.tendrils import View
        self.ta, self.capi, self.dist, self.world, self.rank = ta, _capi, dist, world, rank
        cfg = CONFIGS[config]
        self.cfg, self.config = cfg, config
        self.state_fmt = args.state or cfg["state"]
        self.group = cfg["group"]                    # steps per fused launch and per statistics reduction
        # `--pretend-world P` (experiment): the band rank 0 of a P-rank job would hold, on however many ranks there are - the per-GPU
        # share of a strong-scaling point measured on one GPU (tools/band_sweep.py)
        share = getattr(args, "pretend_world", 0) or world
        self.share = share
        self.width, self.rows, self.gheight = cfg["width"], cfg["rows"](share), cfg["gheight"](share)
        self.particles_rank = self.width * self.rows
        self.launch_len = launch_len or min(self.group, args.steps)
        opts = ta.defaults()
        opts.update(device=local_rank, mode=ta.TH_MODE_FAST if args.mode == "fast" else ta.TH_MODE_EXACT,
                    row0=shard_rows(self.gheight, share, rank if share == world else 0)[0], rows=self.rows, globalHeight=self.gheight,
                    stateFormat=ta.TH_STATE_F16 if self.state_fmt == "f16" else ta.TH_STATE_F32)
        t = self.t = ta.Tendrils(View(W.FLOW_W, W.FLOW_H), opts)
        t.resize()                       # viewRes 1920x1080 -> viewSize [1, 1.7778]; flow.shape = viewRes
        t.setup(self.width)
        ctx = self.ctx = t.particles._ctx
        self.args, self.rank_seed = args, rank
        self.upload_synthetic()

        # flow field: optical-flow pass over the synthetic frame pair (C3), else a seeded field
        self.time0 = 1000.0
        self.flow_source = "optical-flow(synthetic 1080p frame pair)"
        self.of = None
        try:
            from tendrils_amd.optical_flow import OpticalFlow
            f0, f1 = synth_frames()
            of = OpticalFlow(t, uniforms=dict(speed=0.08, offset=0.1, scaleUV=[-1, -1]))   # src/demo.main.js:526-530
            of.resize([W.FLOW_W, W.FLOW_H])
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
            # `comm_guard` (the headline's job: benchlib/sidelegs.py comm_deadline): a communicator that never comes up - a rank
            # that does not arrive - ends in a fresh child process that reduces through torch.distributed, not in silence
            import contextlib
            import os
            import torch
            why = (os.environ.get("TH_BENCH_COMM_FALLBACK") or "--no-library-comm") if args.no_library_comm else ""
            with (comm_guard if comm_guard is not None and not why else contextlib.nullcontext()):
                try:
                    if not why:
                        comm_init(ctx, dist)
                        self.comm = comm_query(ctx)
                except ta.TendrilsHipError as e:
                    why = str(e)
                ok = torch.tensor([0 if why else 1], dtype=torch.int32, device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                agreed = int(ok.item())
            if agreed == 0:
                if self.comm is not None:
                    _capi.call("th_comm_destroy", ctx)
                self.comm = None
                self.comm_fallback = why or "another rank could not join the library's communicator"
                self._counters_view = None
        self.reductions = 0
        self.barrier = NodeBarrier.of(dist, rank, world) if dist is not None else None
        if args.flow_only:
            t.state["noiseWeight"] = 0

    def upload_synthetic(self):
        """the configuration's seeded state into ring buffer 0 ... (generated and uploaded in row bands: bounded host memory at C5)"""
        band, rank, t = 1024, self.rank_seed, self.t
        full = synth_state(rank) if self.config == "c3" else None
        for r0 in range(0, self.rows, band):
            r1 = min(self.rows, r0 + band)
            st = full[r0:r1] if full is not None else synth_rows(self.width, r1 - r0, 12345 + rank * 1000003 + r0)
            if self.args.in_view:
                st = st.copy()
                st[..., 1] *= np.float32(0.56)
            self.capi.call("th_upload_state", self.ctx, -1, np.ascontiguousarray(st).ctypes.data_as(self.capi._fp), 0, r0, self.width, r1 - r0)

    def sync_all(self):
        import torch
        self.t.particles.sync()
        torch.cuda.synchronize()
        if self.dist is not None:
            self.barrier()               # (every rank's device is idle: nothing to synchronize behind it)

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


