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
    hang = os.environ.get("TH_BENCH_TEST_HANG", "")
    # where the GPU run brings the library's communicator up (benchlib/job.py), under the same deadline: a rank that never
    # arrives (TH_BENCH_TEST_HANG=comm_init:<rank>) sends every rank into a fresh child with --no-library-comm
    fallback = None
    if args.no_library_comm:
        fallback = os.environ.get("TH_BENCH_COMM_FALLBACK") or "--no-library-comm"
    else:
        from .sidelegs import comm_deadline
        with comm_deadline(rank, sys.argv[1:], BENCH):
            if hang == "comm_init:%d" % rank:
                time.sleep(1e6)
            dist.barrier()

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

    from .job import NodeBarrier
    barrier = NodeBarrier.of(dist, rank, world)           # (the bracket's barrier of the GPU run: the ranks meet in shared memory)
    run(args.warmup)
    walls = []
    for _ in range(args.reps):
        barrier()
        t0 = time.perf_counter()
        state["reductions"] = 0
        run(args.steps)
        barrier()
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
            "counters": state["red"], "barrier": barrier.kind}
    if fallback:
        line["rccl"]["fallback"] = fallback
    # the side legs under their deadlines, as in the GPU run (benchlib/sidelegs.py).  TH_BENCH_TEST_HANG=<leg>:<rank> (tests): that
    # rank never reaches the leg's first collective
    from .sidelegs import SideLegs
    legs = SideLegs(line, rank)

    def leg(name, fn):
        def body():
            if hang == "%s:%d" % (name, rank):
                time.sleep(1e6)
            return fn()
        return body
    legs.run("c3_strong", leg("c3_strong", lambda: dry_strong(args, rank, world, dist, O, packed=False)), ["c5", "frame_loop_sharded"])
    legs.run("c5", leg("c5", lambda: dry_strong(args, rank, world, dist, O, packed=True)), ["frame_loop_sharded"])
    legs.run("frame_loop_sharded", leg("frame_loop_sharded", lambda: dry_frame_loop_sharded(rank, world, dist, O, state["band"], fl, row0, gheight)))

    def finish():
        dist.barrier()
        dist.destroy_process_group()
    legs.run("shutdown", finish, record=False)
    if rank == 0:
        print(json.dumps(line), flush=True)


def dry_strong(args, rank, world, dist, O, packed):
    """(dry run) the strong-scaling legs' plumbing: ONE texture row-sharded over the ranks, counters reduced after every 4-step
    group.  packed (config 5): the state kept in the packed 8-byte form between steps - SNORM16 position over [-2, 2), fp16
    velocity, as th_logic.hpp packs it (inert and NaN codes aside: none occur here); else (c3_strong) plain f32 texels."""
    import torch
    from tendrils_amd.sharding import reduce_counters, shard_rows
    n, gheight = 32, 64
    row0, rows = shard_rows(gheight, world, rank)
    band = synth_rows(n, gheight, 4242)[row0:row0 + rows]            # (every rank generates the texture, keeps its band)
    fl = np.zeros((27, 48, 4), np.float32)
    fl[..., :2] = np.random.default_rng(6).uniform(-.01, .01, (27, 48, 2))
    fl[..., 2] = 990.0

    def pack(b):
        if not packed:
            return b
        return np.rint(np.clip(b[..., :2] * np.float32(16384.0), -32767, 32767)).astype(np.int16), b[..., 2:].astype(np.float16)

    def unpack(q):
        if not packed:
            return q
        out = np.empty(q[0].shape[:2] + (4,), np.float32)
        out[..., :2] = q[0].astype(np.float32) * np.float32(6.103515625e-05)
        out[..., 2:] = q[1].astype(np.float32)
        return out
    ring, tm, red, reductions = pack(band), 1000.0, None, 0
    dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        tm += 1000.0 / 60.0
        u = O.logic_uniforms(n, gheight, tm, 1000.0 / 60.0, view_size=(1, 48 / 27))
        ring = pack(O.logic_step(u, unpack(ring), fl, y0=row0))
        if (k + 1) % 4 == 0 or k + 1 == args.steps:
            b = unpack(ring)
            sp = np.hypot(b[..., 2].astype(np.float64), b[..., 3].astype(np.float64))
            red = reduce_counters(dist, dict(particles=rows * n, live=rows * n, nan=0, capped=0, respawned=0,
                                             sum_speed=float(sp.sum()), max_speed=float(sp.max())))
            reductions += 1
    dist.barrier()
    v = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return {"dry_run": True, "value": n * gheight * args.steps / float(v[0]), "unit": "particle-steps/s", "scaling": "strong",
            "n_gpus": world, "steps": args.steps, "particles": n * gheight, "particles_per_gpu": n * rows, "rows_per_gpu": rows,
            "rccl": {"world": world, "nranks_seen": red["particles"] / float(rows * n), "backend": "gloo", "reductions": reductions},
            "roofline": {"bound": "hbm", "frac": None, "note": "dry run: nothing measured (%d B per particle-step on the GPU run)" % (16 if packed else 32)}}


def dry_frame_loop_sharded(rank, world, dist, O, band, fl, row0, gheight, frames=2):
    """(dry run) the exchange of a row-band job's draw() over gloo: every rank steps its band, keys a fragment per particle in
    view by the flow texel it lies in (owner << 56 | texel << 32 | global particle id - th::kOwnerShift), the counts and
    then the keys travel to the texels' owners (all-to-all), the owner tallies them per texel, and the owned ranges are
    gathered back to every rank - the collectives of th_draw_sharded's stream-ordered pass, with a count in place of the blend."""
    import torch
    from tendrils_amd.sharding import OWNER_SHIFT, TEXEL_MASK, owner_chunk, split_by_owner
    n = band.shape[1]
    fh, fw = fl.shape[:2]
    texels = fw * fh
    chunk = owner_chunk(texels, world)
    tm, sent, recv, frags, walls = 2000.0, [], [], [], []
    for _ in range(frames):
        dist.barrier()
        t0 = time.perf_counter()
        tm += 1000.0 / 60.0
        u = O.logic_uniforms(n, gheight, tm, 1000.0 / 60.0, view_size=(1, fw / fh))
        band = O.logic_step(u, band, fl, y0=row0)
        x, y = band[..., 0].astype(np.float64), band[..., 1].astype(np.float64) * (fw / fh)
        inside = (np.abs(x) < 1) & (np.abs(y) < 1)
        tx = np.clip(((x + 1) * 0.5 * fw).astype(np.int64), 0, fw - 1)[inside]
        ty = np.clip(((y + 1) * 0.5 * fh).astype(np.int64), 0, fh - 1)[inside]
        texel = ty * fw + tx
        ids = (np.arange(band.shape[0] * n, dtype=np.int64).reshape(band.shape[:2]) + row0 * n)[inside]
        owner = np.minimum(texel // chunk, world - 1)
        order = np.lexsort((ids, owner))
        keys = torch.from_numpy(((owner << OWNER_SHIFT) | (texel << 32) | ids)[order])
        send = split_by_owner(keys, texels, world)
        send_t, recv_t = torch.tensor(send, dtype=torch.int64), torch.empty(world, dtype=torch.int64)
        dist.all_to_all_single(recv_t, send_t)
        got = [int(c) for c in recv_t.tolist()]
        mine = torch.empty(sum(got), dtype=torch.int64)
        dist.all_to_all_single(mine, keys, got, send)
        tally = np.zeros(chunk, np.int64)
        np.add.at(tally, ((mine.numpy() >> 32) & TEXEL_MASK) - rank * chunk, 1)
        parts = [torch.empty(chunk, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(tally))
        whole = torch.cat(parts).numpy()
        total = torch.tensor([int(inside.sum())], dtype=torch.int64)
        dist.all_reduce(total)
        assert int(whole.sum()) == int(total[0]), "fragments were lost or duplicated in the exchange"
        dist.barrier()
        walls.append(time.perf_counter() - t0)
        sent.append(8 * (sum(send) - send[rank]) + 8 * chunk)
        recv.append(8 * (sum(got) - got[rank]) + 8 * chunk * (world - 1))
        frags.append(int(total[0]))
    v = torch.tensor([float(np.median(walls)) * 1e3, float(np.median(sent)), float(np.median(recv))], dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return {"dry_run": True, "frames": frames, "n_gpus": world, "wall_ms_per_frame": float(v[0]), "sent_bytes_per_draw": float(v[1]),
            "received_bytes_per_draw": float(v[2]), "fragments_per_draw_all_ranks": float(np.mean(frags)), "pipeline": "stream (tallied, not blended)"}


