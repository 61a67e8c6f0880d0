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


def share_comm_id(dist, make_id):
    """The 128-byte communicator id of the job: rank 0 makes it (`make_id()` -> bytes: th_comm_unique_id), every other
    rank receives it through `dist` - any torch.distributed backend will do (gloo in the CPU tests, nccl on the GPUs):
    the id is the only thing the library asks the host to carry between the ranks."""
    box = [bytes(make_id()) if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def comm_id():
    """th_comm_unique_id -> bytes (a fresh id; meant for rank 0)"""
    from . import _capi
    buf = (C.c_ubyte * _capi.COMM_ID_BYTES)()
    _capi.call("th_comm_unique_id", buf)
    return bytes(buf)


def loopback_id():
    """th_comm_loopback_id -> bytes: the id of an in-process world (contexts of ONE process, a host thread per rank) - the
    exchange logic with more ranks than the box has GPUs; th_comm_init takes it like an RCCL id"""
    from . import _capi
    buf = (C.c_ubyte * _capi.COMM_ID_BYTES)()
    _capi.call("th_comm_loopback_id", buf)
    return bytes(buf)


def comm_join(ctx, ident, rank, world):
    """th_comm_init with an id the caller carries itself"""
    from . import _capi
    buf = (C.c_ubyte * _capi.COMM_ID_BYTES).from_buffer_copy(ident)
    _capi.call("th_comm_init", ctx, buf, int(rank), int(world))


def comm_init(ctx, dist):
    """Every rank, collectively: the context joins the job's RCCL communicator (th_comm_init), the id travelling from
    rank 0 through `dist`.  From then on the path's collective - the counter all-reduce - is the library's own
    (th_stats_allreduce on the context's stream), whatever the host is."""
    from . import _capi
    ident = share_comm_id(dist, comm_id)
    buf = (C.c_ubyte * _capi.COMM_ID_BYTES).from_buffer_copy(ident)
    _capi.call("th_comm_init", ctx, buf, int(dist.get_rank()), int(dist.get_world_size()))


def comm_query(ctx):
    from . import _capi
    q = _capi.CommInfo()
    _capi.call("th_comm_query", ctx, C.byref(q))
    return {"active": bool(q.active), "rank": q.rank, "world": q.world, "rccl_version": q.rccl_version}


class DeviceCounters:
    """The context's device-side th_counters block reduced over the ranks in place, on the context's own stream: a thin
    caller of th_stats_allreduce (the library's RCCL all-reduce; round 2 reduced torch views of the block through
    torch.distributed)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def all_reduce_async(self):
        """enqueue the reduction behind the statistics pass; nothing to wait for on the host (stream order)"""
        from . import _capi
        _capi.call("th_stats_allreduce", self.ctx)


# ---- flow deposit across row-band shards ------------------------------------------------------------------
# (torch bundles its own ROCm runtime: tendrils_amd/_capi.py:load() imports torch before it loads the library and
# refuses to run with two HSA runtimes mapped, so the order of the caller's imports does not matter.)
# Tendrils.draw() blends every particle line into the flow texture in the order of ONE vertex stream
# (src/index.js:295-303); with the particles split into row bands that stream interleaves the bands column by
# column.  Exact multi-GPU form: every rank rasterises its own lines into fragments keyed
# (owner rank, flow texel, global stream index) and parts them by owner - contiguous texel ranges of
# ceil(texels/world) - with one radix pass [th_deposit_emit]; an all-to-all sends every part to its owner; the
# owner sorts what it received by texel (stably: inside a texel the parts of the source bands follow each other,
# each in stream order) and merges the bands by stream index as it blends [th_deposit_merge]; an all-gather of the
# owned ranges restores the replicated flow texture.

OWNER_SHIFT = 56            # th::kOwnerShift: key = owner << 56 | texel << 32 | stream index
TEXEL_MASK = 0xffffff


def device_view(ptr, shape, typestr):
    """Zero-copy torch view of library-owned device memory."""
    import torch

    class _Span:
        def __init__(self):
            self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                             "version": 3, "strides": None}
    return torch.as_tensor(_Span(), device="cuda")


def owner_chunk(texels, world):
    return (int(texels) + int(world) - 1) // int(world)


def split_by_owner(keys, texels, world):
    """keys: int64 tensor parted by owner (th_deposit_emit after set_owners(world)).  Returns the per-destination
    counts (list)."""
    import torch
    bounds = torch.tensor([r << OWNER_SHIFT for r in range(1, world)], dtype=torch.int64, device=keys.device)
    cuts = torch.searchsorted(keys, bounds).tolist() if world > 1 else []      # (parted is enough: owner >= r is monotonic)
    edges = [0] + [int(v) for v in cuts] + [int(keys.numel())]
    return [edges[r + 1] - edges[r] for r in range(world)]


def set_owners(tendrils, world):
    """The number of ranks owning flow texels: th_deposit_emit parts its fragments for them."""
    from . import _capi
    _capi.call("th_deposit_set_owners", tendrils.particles._ctx, int(world))


def emit_fragments(tendrils):
    """th_deposit_emit on this rank's context -> (keys int64[n], colors float32[n, 4]) views (or empty tensors)."""
    import torch
    from . import _capi
    p = tendrils.particles
    u = _capi.DepositUniforms(time=float(tendrils.timer.time), speedLimit=float(tendrils.state["speedLimit"]))
    u.viewSize[0], u.viewSize[1] = float(tendrils.viewSize[0]), float(tendrils.viewSize[1])
    n, kp, cp = C.c_uint64(0), C.c_void_p(), C.c_void_p()
    _capi.call("th_deposit_emit", p._ctx, C.byref(u), C.byref(n), C.byref(kp), C.byref(cp))
    if n.value == 0:
        return torch.empty(0, dtype=torch.int64, device="cuda"), torch.empty((0, 4), dtype=torch.float32, device="cuda")
    return device_view(kp.value, (n.value,), "<i8"), device_view(cp.value, (n.value, 4), "<f4")


def edge_rows(tendrils):
    """First and last row of buffers[0] and buffers[1] of this band: tensor [2, 2, W, 4] (a copy)."""
    import torch
    from . import _capi
    p = tendrils.particles
    p.sync()                                                     # the context's stream has produced the state
    w, h = p.shape
    out = torch.empty((2, 2, w, 4), dtype=torch.float32, device="cuda")
    for b in range(2):
        ptr = C.c_void_p()
        _capi.call("th_state_device_ptr", p._ctx, b, C.byref(ptr))
        buf = device_view(ptr.value, (h, w, 4), "<f4")
        out[0, b] = buf[0]
        out[1, b] = buf[h - 1]
    torch.cuda.synchronize()
    return out


def set_halo(tendrils, lo, hi):
    """lo / hi: [2 (cur, prev), W, 4] device tensors (or None) - the neighbouring bands' edge rows."""
    from . import _capi
    tendrils._halo = (lo, hi)                                    # keep the tensors alive while the library reads them
    _capi.call("th_deposit_set_halo", tendrils.particles._ctx,
               C.c_void_p(lo.data_ptr()) if lo is not None else None,
               C.c_void_p(hi.data_ptr()) if hi is not None else None)


def merge_fragments(tendrils, keys, colors):
    from . import _capi
    if keys.numel():
        assert keys.is_contiguous() and colors.is_contiguous()
        _capi.call("th_deposit_merge", tendrils.particles._ctx, C.c_void_p(keys.data_ptr()), C.c_void_p(colors.data_ptr()),
                   C.c_uint64(keys.numel()))


def flow_view(tendrils):
    from . import _capi
    fw, fh = tendrils.flow.shape
    ptr = C.c_void_p()
    _capi.call("th_flow_device_ptr", tendrils.particles._ctx, C.byref(ptr))
    return device_view(ptr.value, (fw * fh, 4), "<f4")


def emit_view_fragments(tendrils):
    """th_view_emit on this rank's context -> (keys int64[n], colors float32[n, 4]) views (or empty tensors): the view
    pass's fragments of this band's lines, parted by owner like the flow pass's."""
    import torch
    from . import _capi
    u = tendrils.render_uniforms()
    n, kp, cp = C.c_uint64(0), C.c_void_p(), C.c_void_p()
    _capi.call("th_view_emit", tendrils.particles._ctx, C.byref(u), C.byref(n), C.byref(kp), C.byref(cp))
    if n.value == 0:
        return torch.empty(0, dtype=torch.int64, device="cuda"), torch.empty((0, 4), dtype=torch.float32, device="cuda")
    return device_view(kp.value, (n.value,), "<i8"), device_view(cp.value, (n.value, 4), "<f4")


def merge_view_fragments(tendrils, keys, colors):
    from . import _capi
    if keys.numel():
        assert keys.is_contiguous() and colors.is_contiguous()
        _capi.call("th_view_merge", tendrils.particles._ctx, C.c_void_p(keys.data_ptr()), C.c_void_p(colors.data_ptr()),
                   C.c_uint64(keys.numel()))


def emit_draw_fragments(tendrils):
    """th_draw_emit -> (keys int64[n], colors float32[n, 8]): both passes' fragments of this band's lines in one - the flow
    pass's varying and the view pass's colour side by side -, parted by owner.  Needs both passes to draw with one line width."""
    import torch
    from . import _capi
    d = _capi.DepositUniforms(time=float(tendrils.timer.time), speedLimit=float(tendrils.state["speedLimit"]))
    d.viewSize[0], d.viewSize[1] = float(tendrils.viewSize[0]), float(tendrils.viewSize[1])
    u = tendrils.render_uniforms()
    n, kp, cp = C.c_uint64(0), C.c_void_p(), C.c_void_p()
    _capi.call("th_draw_emit", tendrils.particles._ctx, C.byref(d), C.byref(u), C.byref(n), C.byref(kp), C.byref(cp))
    if n.value == 0:
        return torch.empty(0, dtype=torch.int64, device="cuda"), torch.empty((0, 8), dtype=torch.float32, device="cuda")
    return device_view(kp.value, (n.value,), "<i8"), device_view(cp.value, (n.value, 8), "<f4")


def merge_draw_fragments(tendrils, keys, colors):
    from . import _capi
    if keys.numel():
        assert keys.is_contiguous() and colors.is_contiguous() and colors.shape[-1] == 8
        _capi.call("th_draw_merge", tendrils.particles._ctx, C.c_void_p(keys.data_ptr()), C.c_void_p(colors.data_ptr()),
                   C.c_uint64(keys.numel()))


def same_line_widths(tendrils):
    """both passes of draw() draw their lines equally wide (after the clamp to the context's range)"""
    from . import _capi
    d0, d1 = C.c_float(), C.c_float()
    _capi.call("th_line_width_query", tendrils.particles._ctx, _capi.TH_PASS_FLOW, None, C.byref(d0), None)
    _capi.call("th_line_width_query", tendrils.particles._ctx, _capi.TH_PASS_VIEW, None, C.byref(d1), None)
    return d0.value == d1.value


def view_view(tendrils):
    """the context's RGBA8 view buffer as a [texels, 4] uint8 tensor"""
    from . import _capi
    fw, fh = tendrils.flow.shape
    ptr = C.c_void_p()
    _capi.call("th_view_device_ptr", tendrils.particles._ctx, C.byref(ptr))
    return device_view(ptr.value, (fw * fh, 4), "|u1")


def gather_state(tendrils, buffer=0):
    """Row-band shard: the whole particle texture of ring buffer `buffer` on every rank (th_state_gather: the library's RCCL
    all-gather; needs sharding.comm_init), for the spawners that sample arbitrary particles (src/demo.main.js:433-441)."""
    from . import _capi
    index = buffer.index if hasattr(buffer, "index") else int(buffer)
    _capi.call("th_state_gather", tendrils.particles._ctx, index)


def _exchange(dist, keys, colors, texels):
    """the owners' all-to-all: every part of (keys, colors) to its owner; returns what this rank received"""
    import torch
    world = dist.get_world_size()
    send = split_by_owner(keys, texels, world)
    send_t = torch.tensor(send, dtype=torch.int64, device="cuda")
    recv_t = torch.empty_like(send_t)
    dist.all_to_all_single(recv_t, send_t)
    recv = [int(v) for v in recv_t.tolist()]
    rkeys = torch.empty(sum(recv), dtype=torch.int64, device="cuda")
    rcolors = torch.empty((sum(recv), colors.shape[1]), dtype=torch.float32, device="cuda")     # (4, or 8: both passes' varyings)
    dist.all_to_all_single(rkeys, keys.contiguous(), recv, send)
    dist.all_to_all_single(rcolors, colors.contiguous(), recv, send)
    torch.cuda.synchronize()
    return rkeys, rcolors


def _gather_owned(dist, plane, texels):
    """every owner's texel range of `plane` ([texels, k]) to every rank (equal chunks: the tail padded)"""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    chunk = owner_chunk(texels, world)
    mine = torch.zeros((chunk, plane.shape[1]), dtype=plane.dtype, device="cuda")
    lo, hi = min(rank * chunk, texels), min((rank + 1) * chunk, texels)
    mine[:hi - lo] = plane[lo:hi]
    gathered = torch.empty((world * chunk, plane.shape[1]), dtype=plane.dtype, device="cuda")
    dist.all_gather_into_tensor(gathered, mine)
    plane.copy_(gathered[:texels])
    torch.cuda.synchronize()


def draw_sharded(dist, tendrils, view=False):
    """Tendrils.draw() for a row-band shard of a torch.distributed job (backend nccl = RCCL): the flow pass and - `view` -
    the view pass (after the clear / fade the caller applied to its copy of the view buffer, the same on every rank).
    One exchange step per pass: fragment all-to-all by texel owner, then an all-gather of the owned ranges."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    fw, fh = tendrils.flow.shape
    texels = fw * fh
    # edge rows of both state buffers to the neighbouring bands (the fp32 row lookup of the vertex stream can land
    # one row beside a line's own row for some texture heights)
    lo = hi = None
    if tendrils._state_format == 0 and world > 1:                # f32 ring (a packed band keeps to its own rows)
        edges = edge_rows(tendrils)                              # [2 (first, last), 2 (cur, prev), W, 4]
        every = torch.empty((world,) + tuple(edges.shape), dtype=torch.float32, device="cuda")
        dist.all_gather_into_tensor(every, edges)
        lo = every[rank - 1, 1].contiguous() if rank > 0 else None
        hi = every[rank + 1, 0].contiguous() if rank + 1 < world else None
        torch.cuda.synchronize()
    set_halo(tendrils, lo, hi)
    set_owners(tendrils, world)
    if view and same_line_widths(tendrils):
        # both passes draw the same lines: one rasterisation, one exchange of fragments carrying both varyings, two all-gathers
        keys, colors = emit_draw_fragments(tendrils)
        fragments = int(keys.numel())
        rkeys, rcolors = _exchange(dist, keys, colors, texels)
        merge_draw_fragments(tendrils, rkeys, rcolors)
        _gather_owned(dist, flow_view(tendrils), texels)
        _gather_owned(dist, view_view(tendrils), texels)
        return fragments
    keys, colors = emit_fragments(tendrils)
    fragments = int(keys.numel())
    rkeys, rcolors = _exchange(dist, keys, colors, texels)
    merge_fragments(tendrils, rkeys, rcolors)
    _gather_owned(dist, flow_view(tendrils), texels)
    if view:
        keys, colors = emit_view_fragments(tendrils)
        rkeys, rcolors = _exchange(dist, keys, colors, texels)
        merge_view_fragments(tendrils, rkeys, rcolors)
        _gather_owned(dist, view_view(tendrils), texels)
    return fragments


def draw_sharded_native(tendrils, view=False):
    """The same draw() with the exchange issued by the LIBRARY over its own communicator (th_draw_sharded; needs
    sharding.comm_init): what a Node process per GPU calls as drawSharded - no torch.distributed in the data path."""
    from . import _capi
    d = _capi.DepositUniforms(time=float(tendrils.timer.time), speedLimit=float(tendrils.state["speedLimit"]))
    d.viewSize[0], d.viewSize[1] = float(tendrils.viewSize[0]), float(tendrils.viewSize[1])
    n = C.c_uint64(0)
    u = tendrils.render_uniforms() if view else None
    _capi.call("th_draw_sharded", tendrils.particles._ctx, C.byref(d), C.byref(u) if view else None, C.byref(n))
    return int(n.value)
