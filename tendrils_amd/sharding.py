"""Multi-GPU layout of the particle path: one process per GPU, each holding a contiguous
row band of the N x H state texture (every ring buffer and the targets texture), with the
flow texture and the optical-flow frames replicated.  Within a step particles are
independent given the read-only flow/targets (src/logic.frag:48,75,85 read only the
particle's own texel), so the data path needs no collective; the only exchange is the
small reduction of the statistics counters.
"""
import ctypes as C


def shard_rows(global_height, world, rank):
    """Contiguous, balanced row bands: returns (row0, rows) for `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(int(global_height), int(world))
    rows = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, rows


COUNT_FIELDS = ("particles", "live", "nan", "capped", "respawned")


def reduce_counters(dist, counters, device=None):
    """All-reduce a th_counters dict over the process group: counts and sum_speed by SUM,
    max_speed by MAX.  Works on any backend (RCCL on the GPUs, gloo in the CPU tests)."""
    import torch
    kw = {"device": device} if device is not None else {}
    counts = torch.tensor([int(counters[k]) for k in COUNT_FIELDS], dtype=torch.int64, **kw)
    ssum = torch.tensor([float(counters["sum_speed"])], dtype=torch.float64, **kw)
    smax = torch.tensor([float(counters["max_speed"])], dtype=torch.float64, **kw)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
    dist.all_reduce(smax, op=dist.ReduceOp.MAX)
    out = {k: int(v) for k, v in zip(COUNT_FIELDS, counts.tolist())}
    out["sum_speed"] = float(ssum[0])
    out["max_speed"] = float(smax[0])
    return out


class DeviceCounters:
    """Zero-copy torch views of the context's device-side th_counters block (5 x u64, 2 x f64)
    so that RCCL can reduce it in place on the context's own stream."""

    def __init__(self, dptr):
        import torch

        class _Span:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                                 "version": 3, "strides": None}
        self.counts = torch.as_tensor(_Span(dptr, 5, "<i8"), device="cuda")
        self.sum_speed = torch.as_tensor(_Span(dptr + 40, 1, "<f8"), device="cuda")
        self.max_speed = torch.as_tensor(_Span(dptr + 48, 1, "<f8"), device="cuda")

    def all_reduce_async(self, dist):
        return [dist.all_reduce(self.counts, op=dist.ReduceOp.SUM, async_op=True),
                dist.all_reduce(self.sum_speed, op=dist.ReduceOp.SUM, async_op=True),
                dist.all_reduce(self.max_speed, op=dist.ReduceOp.MAX, async_op=True)]
