"""Frame loop of the reference across row-band shards: step() + draw() per frame, one process per GPU.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/frame_bench_dist.py [frames]
(also runs as a plain script = world size 1).  Prints ms per frame part on rank 0 (max over ranks) and checks
that every rank ends with the same flow field."""
import json
import os
import sys
import time

import numpy as np
import torch                     # before the library (two HIP runtimes in one process, see tendrils_amd/sharding.py)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import workload as bench  # noqa: E402  (the synthetic C3 inputs)
import tendrils_amd as ta  # noqa: E402
from tendrils_amd.sharding import flow_view, shard_rows  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(os.environ.get("TH_N", "4096"))               # particles per GPU: N x N
rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world)

bench.N = N
opts = ta.defaults()
opts.update(device=local, row0=shard_rows(N * world, world, rank)[0], rows=N, globalHeight=N * world, dist=dist)
t = ta.Tendrils(View(1920, 1080), opts)
t.resize()
t.setup(N)
t.particles.upload_texels(bench.synth_state(rank))
t.timer.time = 1000.0


def timed(fn):
    torch.cuda.synchronize(); t.particles.sync()
    t0 = time.perf_counter()
    fn()
    t.particles.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(3):
    t.timer.tick(); t.step(); t.draw()
step_ms, draw_ms, frags = [], [], []
dist.barrier()
for _ in range(frames):
    t.timer.tick()
    step_ms.append(timed(t.step))
    draw_ms.append(timed(t.draw))
    frags.append(t.fragments)
part = torch.tensor([np.mean(step_ms), np.mean(draw_ms)], dtype=torch.float64, device="cuda")
dist.all_reduce(part, op=dist.ReduceOp.MAX)
fsum = torch.tensor([float(np.mean(frags))], dtype=torch.float64, device="cuda")
dist.all_reduce(fsum, op=dist.ReduceOp.SUM)
# the replicated flow field must be identical everywhere
flow = flow_view(t)
sig = torch.stack([flow.double().sum(), flow.double().abs().max()])
lo, hi = sig.clone(), sig.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN)
dist.all_reduce(hi, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"n_gpus": world, "particles": N * N * world, "frames": frames, "step_ms": float(part[0]),
                      "draw_ms": float(part[1]), "fragments_per_frame": float(fsum[0]),
                      "frames_per_s": 1e3 / float(part[0] + part[1]), "flow_identical_on_all_ranks": bool((lo == hi).all())}))
t.dispose()
dist.barrier()
dist.destroy_process_group()
