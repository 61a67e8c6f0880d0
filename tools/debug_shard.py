"""Which texels differ between the sharded and the unsharded deposit (crowded case)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from tendrils_amd import sharding
from test_gpu_deposit_sharded import make_shard
n, view, world, spread = 128, (48, 27), int(sys.argv[1]) if len(sys.argv) > 1 else 4, float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rng = np.random.default_rng(n + world)
prev = np.zeros((n, n, 4), np.float32)
prev[..., :2] = rng.uniform(-spread, spread, (n, n, 2)) * [1.0, view[1] / view[0]]
prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
cur = prev.copy()
cur[..., :2] += rng.uniform(-.08, .08, (n, n, 2)).astype(np.float32)
cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
k = rng.random((n, n)) < 0.1
cur[k] = [-1e6, -1e6, 0, 0]
fw, fh = view
base = np.zeros((fh, fw, 4), np.float32)
want, frags, cov = O.flow_deposit(cur, prev, base, 2500.0, view_size=(1.0, fw / fh), coverage=True)
shards = [make_shard(n, view, *sharding.shard_rows(n, world, r), cur, prev, base, 2500.0) for r in range(world)]
texels = fw * fh
chunk = sharding.owner_chunk(texels, world)
for t in shards: sharding.set_owners(t, world)
emitted = [sharding.emit_fragments(t) for t in shards]
print("frags", frags, sum(int(k.numel()) for k, _ in emitted), "max cov", cov.max())
sends = [sharding.split_by_owner(k, texels, world) for k, _ in emitted]
print(sends)
for d, t in enumerate(shards):
    pk, pc = [], []
    for s, (keys, colors) in enumerate(emitted):
        lo = sum(sends[s][:d]); pk.append(keys[lo:lo + sends[s][d]].clone()); pc.append(colors[lo:lo + sends[s][d]].clone())
    rk, rc = torch.cat(pk), torch.cat(pc)
    sharding.merge_fragments(t, rk.contiguous(), rc.contiguous())
    got = t.flow.read().reshape(-1, 4)[d * chunk:min((d + 1) * chunk, texels)]
    w = want.reshape(-1, 4)[d * chunk:min((d + 1) * chunk, texels)]
    bad = np.where((got.view(np.uint32) != w.view(np.uint32)).any(axis=1))[0]
    c = cov.reshape(-1)[d * chunk:min((d + 1) * chunk, texels)]
    print("owner", d, "received", rk.numel(), "bad texels", len(bad), "their coverage", sorted(set(c[bad].tolist()))[:20], "max", c.max())
    txa = ((rk >> 32) & sharding.TEXEL_MASK).cpu().numpy(); ida = (rk & 0xffffffff).cpu().numpy()
    exp = 0; lens = []
    for tt in np.unique(txa):
        q = ida[txa == tt]
        if (np.diff(q.astype(np.int64)) < 0).any(): exp += 1; lens.append(len(q))
    print(" texels with a fall of the stream index:", exp, "run lengths", sorted(lens))
    if len(bad):
        tx = ((rk >> 32) & sharding.TEXEL_MASK).cpu().numpy(); ids = (rk & 0xffffffff).cpu().numpy()
        for b in bad[:2]:
            sel = np.where(tx == d * chunk + b)[0]
            print(" texel", d * chunk + b, "arrival positions", sel.tolist()[:40], "ids", ids[sel].tolist())
            print(" got", got[b], "want", w[b])
