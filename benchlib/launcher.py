"""`bench.py --gpus N` with no launcher around it, and the CPU dry run of the rank plumbing."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .workload import ROOT, synth_rows
from .job import median, repetition_block

BENCH = os.path.join(ROOT, "bench.py")


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
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH] + list(argv)
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


