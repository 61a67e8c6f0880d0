"""Multi-GPU host logic on CPU: world size 2 over gloo.  Each rank integrates its row band with
GLOBAL coordinates (checked here with the oracle standing in for the device), the bands
reassemble to the unsharded result bit-for-bit, and the statistics counters reduce correctly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from tendrils_amd.sharding import reduce_counters, shard_rows

    rng = np.random.default_rng(77)          # same global state on every rank
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    st[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    fl = np.zeros((27, 48, 4), np.float32)
    fl[..., :2] = rng.uniform(-.01, .01, (27, 48, 2))
    fl[..., 2] = 990.0

    row0, rows = shard_rows(n, world, rank)
    u = O.logic_uniforms(n, n, 1016.67, 16.67, view_size=(1, 48 / 27))
    band = O.logic_step(u, st[row0:row0 + rows], fl, y0=row0)
    np.save(os.path.join(out_dir, "band_%d.npy" % rank), band)

    live = (band[..., 0] != -1e6) | (band[..., 1] != -1e6)
    sp = np.hypot(band[..., 2].astype(np.float64), band[..., 3].astype(np.float64))[live]
    local = dict(particles=band.shape[0] * n, live=int(live.sum()), nan=int(np.isnan(band).any(-1).sum()),
                 capped=int((sp >= 0.01 * (1 - 2 ** -20)).sum()), respawned=100 + rank, sum_speed=float(sp.sum()),
                 max_speed=float(sp.max()))
    red = reduce_counters(dist, local)
    if rank == 0:
        full = O.logic_step(u, st, fl)
        np.save(os.path.join(out_dir, "full.npy"), full)
        np.save(os.path.join(out_dir, "red.npy"), np.array([red[k] for k in
                ("particles", "live", "nan", "capped", "sum_speed", "max_speed", "respawned")], np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows_partition():
    from tendrils_amd.sharding import shard_rows
    for h, w in ((4096, 8), (10, 3), (7, 7), (5, 8)):
        spans = [shard_rows(h, w, r) for r in range(w)]
        assert spans[0][0] == 0
        for (a0, an), (b0, _) in zip(spans, spans[1:]):
            assert a0 + an == b0
        assert spans[-1][0] + spans[-1][1] == h
        assert max(s[1] for s in spans) - min(s[1] for s in spans) <= 1
    with pytest.raises(ValueError):
        shard_rows(8, 2, 2)


def test_two_rank_bands_equal_unsharded_run(tmp_path, oracle):
    n, world = 96, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    got = np.concatenate([np.load(tmp_path / ("band_%d.npy" % r)) for r in range(world)])
    assert (got.view(np.uint32) == full.view(np.uint32)).all()
    red = np.load(tmp_path / "red.npy")
    live = (full[..., 0] != -1e6) | (full[..., 1] != -1e6)
    sp = np.hypot(full[..., 2].astype(np.float64), full[..., 3].astype(np.float64))[live]
    assert red[0] == n * n and red[1] == live.sum() and red[2] == 0
    assert abs(red[4] - sp.sum()) < 1e-9 * max(1.0, sp.sum()) and red[5] == sp.max()
    assert red[6] == sum(100 + r for r in range(world))          # respawn counts add up over the ranks


def _exchange_worker(rank, world, port, texels, out_dir):
    """The fragment exchange of draw_sharded (counts, then keys, by flow-texel owner) over gloo, on synthetic
    fragments: every rank ends up with exactly the fragments of its texel range."""
    import torch
    import torch.distributed as dist
    from tendrils_amd.sharding import OWNER_SHIFT, TEXEL_MASK, owner_chunk, split_by_owner
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = 5000 + 700 * rank
    texel = rng.integers(0, texels, n).astype(np.int64)
    ids = (rng.permutation(n).astype(np.int64) * world + rank)          # distinct stream indices across ranks
    # what th_deposit_emit hands over: owner << 56 | texel << 32 | stream index, parted by owner (one stable pass), every
    # part in the band's stream order
    chunk = owner_chunk(texels, world)
    owner = np.minimum(texel // chunk, world - 1)
    order = np.lexsort((ids, owner))
    keys = torch.from_numpy(((owner << OWNER_SHIFT) | (texel << 32) | ids)[order])
    send = split_by_owner(keys, texels, world)
    assert sum(send) == n
    send_t = torch.tensor(send, dtype=torch.int64)
    recv_t = torch.empty_like(send_t)
    dist.all_to_all_single(recv_t, send_t)
    recv = [int(v) for v in recv_t.tolist()]
    rkeys = torch.empty(sum(recv), dtype=torch.int64)
    dist.all_to_all_single(rkeys, keys, recv, send)
    got = rkeys.numpy()
    tx = (got >> 32) & TEXEL_MASK
    assert (tx >= rank * chunk).all() and (tx < (rank + 1) * chunk).all() and ((got >> OWNER_SHIFT) == rank).all()
    np.save(os.path.join(out_dir, "sent_%d.npy" % rank), keys.numpy())
    np.save(os.path.join(out_dir, "recv_%d.npy" % rank), got)
    dist.barrier()
    dist.destroy_process_group()


def test_fragment_exchange_by_texel_owner(tmp_path):
    world, texels = 2, 96 * 54
    mp.spawn(_exchange_worker, args=(world, _free_port(), texels, str(tmp_path)), nprocs=world, join=True)
    sent = np.concatenate([np.load(tmp_path / ("sent_%d.npy" % r)) for r in range(world)])
    recv = np.concatenate([np.load(tmp_path / ("recv_%d.npy" % r)) for r in range(world)])
    assert np.array_equal(np.sort(sent), np.sort(recv))                  # nothing lost, nothing duplicated


def _id_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tendrils_amd.sharding import comm_id, share_comm_id
    made = []

    def make():
        made.append(1)
        return comm_id()              # th_comm_unique_id (ncclGetUniqueId needs no GPU)
    ident = share_comm_id(dist, make)
    assert len(made) == (1 if rank == 0 else 0)          # only rank 0 makes an id
    with open(os.path.join(out_dir, "id_%d.bin" % rank), "wb") as f:
        f.write(ident)
    dist.barrier()
    dist.destroy_process_group()


def test_communicator_id_travels_from_rank_0(tmp_path):
    """The only thing the library asks the host to carry between the ranks: rank 0's 128-byte id (th_comm_unique_id),
    handed to every rank before the collective th_comm_init."""
    world = 2
    mp.spawn(_id_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ids = [open(tmp_path / ("id_%d.bin" % r), "rb").read() for r in range(world)]
    assert len(ids[0]) == 128 and ids[0] == ids[1] and any(ids[0])
