"""Flow deposit across row-band shards: th_deposit_emit / th_deposit_merge and the ownership logic of
tendrils_amd/sharding.py, with two shards as two contexts on this GPU and the exchange done by hand (the
torch.distributed calls of draw_sharded are the same slices sent through all_to_all_single / all_gather).
The result on every shard must equal the unsharded deposit bit for bit."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import bits_equal  # noqa: E402

pytestmark = pytest.mark.gpu


def make_shard(n, view, row0, rows, st_cur, st_prev, base, time):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    opts.update(row0=row0, rows=rows, globalHeight=n)
    t = ta.Tendrils(View(*view), opts)
    t.resize()
    t.setup(n)
    t.particles.upload_texels(st_cur[row0:row0 + rows], 0)
    t.particles.upload_texels(st_prev[row0:row0 + rows], 1)
    t.flow.set_pixels(base)
    t.timer.time = time
    return t


@pytest.mark.parametrize("n,view,world,spread", [(64, (96, 54), 2, 0.95), (128, (48, 27), 4, 0.95), (64, (80, 60), 3, 0.95),
                                                 (128, (48, 27), 4, 0.3), (128, (48, 27), 2, 0.2)])
def test_sharded_deposit_equals_unsharded(oracle, n, view, world, spread):
    """spread < 1 crowds the particles into the view's centre: runs of 100+ fragments per texel made of every band's
    fragments (the owner's merge by stream index: selection for short runs, band cursors for long ones, and the
    wave-wide walk noticing that a run is not in one piece)."""
    torch = pytest.importorskip("torch")
    from tendrils_amd import sharding
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
    base[..., :2] = rng.uniform(-.01, .01, (fh, fw, 2))
    base[..., 2] = 2400.0
    base[..., 3] = rng.uniform(0, 1, (fh, fw))
    time = 2500.0
    want, frags, cov = oracle.flow_deposit(cur, prev, base, time, view_size=(1.0, fw / fh), coverage=True)
    assert cov.max() >= (3 if spread > 0.5 else 100)

    shards = []
    for r in range(world):
        row0, rows = sharding.shard_rows(n, world, r)
        shards.append(make_shard(n, view, row0, rows, cur, prev, base, time))
    texels = fw * fh
    chunk = sharding.owner_chunk(texels, world)
    for t in shards:
        sharding.set_owners(t, world)
    emitted = [sharding.emit_fragments(t) for t in shards]
    assert sum(int(k.numel()) for k, _ in emitted) == frags
    sends = [sharding.split_by_owner(k, texels, world) for k, _ in emitted]
    # "all-to-all": destination d receives, from every source in rank order, the slice addressed to it
    for d, t in enumerate(shards):
        parts_k, parts_c = [], []
        for s, (keys, colors) in enumerate(emitted):
            lo = sum(sends[s][:d])
            parts_k.append(keys[lo:lo + sends[s][d]].clone())
            parts_c.append(colors[lo:lo + sends[s][d]].clone())
        rk, rc = torch.cat(parts_k), torch.cat(parts_c)
        if rk.numel():
            tx = (rk >> 32) & sharding.TEXEL_MASK
            assert int(tx.min()) >= d * chunk and int(tx.max()) < (d + 1) * chunk and bool(((rk >> sharding.OWNER_SHIFT) == d).all())
        sharding.merge_fragments(t, rk.contiguous(), rc.contiguous())
    # "all-gather": every shard takes the owners' ranges
    views = [sharding.flow_view(t) for t in shards]
    owned = [views[d][min(d * chunk, texels):min((d + 1) * chunk, texels)].clone() for d in range(world)]
    for v in views:
        v.copy_(torch.cat(owned))
    torch.cuda.synchronize()
    for t in shards:
        assert bits_equal(t.flow.read(), want).all()
        t.dispose()


def test_shard_draw_points_to_the_exchange():
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    opts.update(row0=16, rows=16, globalHeight=64)
    t = ta.Tendrils(View(32, 32), opts)
    t.resize()
    t.setup(64)
    with pytest.raises(ta.TendrilsHipError) as e:
        t.draw()                                   # a shard's draw() is collective: it needs the job's communicator ...
    assert "th_comm_init" in str(e.value)
    from tendrils_amd import _capi
    import ctypes as C
    n = C.c_uint64(0)
    u = _capi.DepositUniforms(time=1.0, speedLimit=0.01)
    u.viewSize[0] = u.viewSize[1] = 1.0
    with pytest.raises(ta.TendrilsHipError) as e:  # ... and the single-context entry points say where the exchange lives
        _capi.call("th_flow_deposit", t.particles._ctx, C.byref(u), C.byref(n))
    assert e.value.status == 4 and "th_deposit_emit" in str(e.value)
    t.dispose()


def test_draw_sharded_through_rccl_world_size_1(oracle):
    """draw_sharded() itself - the torch.distributed (RCCL) calls - at world size 1 (the only size one GPU allows)."""
    torch = pytest.importorskip("torch")
    import os
    import torch.distributed as dist
    from tendrils_amd import sharding
    n, view = 64, (96, 54)
    rng = np.random.default_rng(12)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.9, 0.9, (n, n, 2)) * [1.0, 0.5]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.05, .05, (n, n, 2)).astype(np.float32)
    base = np.zeros((54, 96, 4), np.float32)
    want, frags = oracle.flow_deposit(cur, prev, base, 700.0, view_size=(1.0, 96 / 54))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        t = make_shard(n, view, 0, n, cur, prev, base, 700.0)
        assert sharding.draw_sharded(dist, t) == frags
        assert bits_equal(t.flow.read(), want).all()
        t.dispose()
        # ... and Tendrils.draw() of a sharded job with renderView: flow pass + view pass, against the local th_draw
        local = make_shard(n, view, 0, n, cur, prev, base, 700.0)
        local.draw()
        t = make_shard(n, view, 0, n, cur, prev, base, 700.0)
        t.dist = dist
        t.draw()
        assert t.fragments == local.fragments == frags and t.view_fragments == frags
        assert bits_equal(t.flow.read(), local.flow.read()).all()
        assert (t.read_view() == local.read_view()).all() and t.read_view().any()
        t.dispose()
        local.dispose()
    finally:
        dist.destroy_process_group()


def test_band_edge_lookup_needs_and_uses_halo_rows(oracle):
    """Texture height 100: the fp32 row lookup of the vertex stream lands on row 52 for line 53 (and 58 for 59).
    A band starting at row 53 fails its emit without the neighbour's edge row and is exact with it."""
    torch = pytest.importorskip("torch")
    import tendrils_amd as ta
    from tendrils_amd import sharding
    n, view = 100, (96, 54)
    rng = np.random.default_rng(100)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.95, 0.95, (n, n, 2)) * [1.0, 54 / 96]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.06, .06, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    base = np.zeros((54, 96, 4), np.float32)
    want, frags = oracle.flow_deposit(cur, prev, base, 900.0, view_size=(1.0, 96 / 54))
    bands = [(0, 53), (53, 47)]
    shards = [make_shard(n, view, r0, rows, cur, prev, base, 900.0) for r0, rows in bands]
    with pytest.raises(ta.TendrilsHipError) as e:
        sharding.emit_fragments(shards[1])
    assert e.value.status == 4 and "halo" in str(e.value)
    edges = [sharding.edge_rows(t) for t in shards]
    sharding.set_halo(shards[0], None, edges[1][0].contiguous())
    sharding.set_halo(shards[1], edges[0][1].contiguous(), None)
    emitted = [sharding.emit_fragments(t) for t in shards]
    assert sum(int(k.numel()) for k, _ in emitted) == frags
    keys = torch.cat([k for k, _ in emitted]).contiguous()
    colors = torch.cat([c for _, c in emitted]).contiguous()
    sharding.merge_fragments(shards[0], keys, colors)            # one owner for everything: order comes from the keys
    assert bits_equal(shards[0].flow.read(), want).all()
    for t in shards:
        t.dispose()


def test_edge_rows_after_a_sorted_step_are_current():
    """sharding.edge_rows reads the band's first and last state rows through torch views of library memory: when the
    step left the slots tile-sorted (option bucket = 1 forces it), th_state_device_ptr first restores texel order on the
    context's own stream - the rows handed to the neighbours must be the restored ones, not a buffer still being written
    (stream race)."""
    import ctypes as C
    pytest.importorskip("torch")
    import tendrils_amd as ta
    from tendrils_amd import sharding
    from tendrils_amd.tendrils import View
    n, rows = 256, 64
    opts = ta.defaults()
    opts.update(row0=64, rows=rows, globalHeight=n)
    t = ta.Tendrils(View(64, 36), opts)
    t.resize()
    t.setup(n)
    t.particles.option("bucket", 1)
    t.particles.option("resort_steps", 2)
    rng = np.random.default_rng(3)
    st = np.empty((rows, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (rows, n, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (rows, n, 2))
    t.particles.upload_texels(st)
    fl = np.zeros((36, 64, 4), np.float32)
    fl[..., :2] = rng.uniform(-.01, .01, (36, 64, 2))
    fl[..., 2] = 990.0
    t.flow.set_pixels(fl)
    t.timer.time = 1000.0
    for k in range(5):
        t.timer.tick()
        t.step()
        info = ta._capi.SlotOrderInfo()
        ta._capi.call("th_slot_order", t.particles._ctx, C.byref(info))
        assert info.sorted_buffers > 0, "the step was expected to leave sorted slots"
        e = sharding.edge_rows(t).cpu().numpy()
        cur, prev = t.particles.read(0), t.particles.read(1)
        assert (e[0, 0] == cur[0]).all() and (e[1, 0] == cur[-1]).all() and (e[0, 1] == prev[0]).all() and (e[1, 1] == prev[-1]).all(), k
    t.dispose()


@pytest.mark.parametrize("n,view,world", [(64, (96, 54), 2), (128, (48, 27), 3)])
def test_sharded_view_pass_equals_unsharded(n, view, world):
    """The view pass of draw() on row-band shards (th_view_emit / th_view_merge, the owners' exchange by hand as above):
    every shard's view buffer ends up byte-identical to the unsharded th_view_draw - crowded texels with fragments of every
    band included - also on top of an earlier frame (a translucent fade in between)."""
    torch = pytest.importorskip("torch")
    from tendrils_amd import sharding
    rng = np.random.default_rng(7 * n + world)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.5, 0.5, (n, n, 2)) * [1.0, view[1] / view[0]]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.08, .08, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    fw, fh = view
    base = np.zeros((fh, fw, 4), np.float32)
    time = 2500.0

    def frame(t):
        """what Tendrils.draw() does to the view before its lines: the fade"""
        t.state["autoClearView"] = False
        t.state["autoFade"] = True
        t.drawFade()

    whole = make_shard(n, view, 0, n, cur, prev, base, time)
    shards = []
    for r in range(world):
        row0, rows = sharding.shard_rows(n, world, r)
        shards.append(make_shard(n, view, row0, rows, cur, prev, base, time))
    texels = fw * fh
    chunk = sharding.owner_chunk(texels, world)
    for t in shards:
        sharding.set_owners(t, world)
    import ctypes as C
    from tendrils_amd import _capi
    for rep in range(2):                       # the second frame blends over the first one's pixels
        frame(whole)
        u, nf = whole.render_uniforms(), C.c_uint64(0)
        _capi.call("th_view_draw", whole.particles._ctx, C.byref(u), C.byref(nf))
        want = whole.read_view()
        for t in shards:
            frame(t)
        emitted = [sharding.emit_view_fragments(t) for t in shards]
        assert sum(int(k.numel()) for k, _ in emitted) == nf.value > 1000
        sends = [sharding.split_by_owner(k, texels, world) for k, _ in emitted]
        for d, t in enumerate(shards):
            parts_k, parts_c = [], []
            for s, (keys, colors) in enumerate(emitted):
                lo = sum(sends[s][:d])
                parts_k.append(keys[lo:lo + sends[s][d]].clone())
                parts_c.append(colors[lo:lo + sends[s][d]].clone())
            sharding.merge_view_fragments(t, torch.cat(parts_k).contiguous(), torch.cat(parts_c).contiguous())
        views = [sharding.view_view(t) for t in shards]
        owned = [views[d][min(d * chunk, texels):min((d + 1) * chunk, texels)].clone() for d in range(world)]
        for v in views:
            v.copy_(torch.cat(owned))
        torch.cuda.synchronize()
        for t in shards:
            got = t.read_view()
            assert (got == want).all() and got.any()
    for t in shards + [whole]:
        t.dispose()


@pytest.mark.parametrize("n,view,world", [(64, (96, 54), 2), (128, (48, 27), 3), (96, (80, 60), 4)])
def test_sharded_draw_in_one_pass_equals_unsharded(n, view, world):
    """Both passes of draw() on row-band shards over ONE rasterisation and ONE exchange (th_draw_emit / th_draw_merge:
    fragments carrying the flow pass's varying and the view pass's colour side by side; the owners' exchange by hand as
    above): every shard's flow texture and view buffer end up identical to the unsharded th_draw, frame after frame."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from tendrils_amd import _capi, sharding
    rng = np.random.default_rng(11 * n + world)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.6, 0.6, (n, n, 2)) * [1.0, view[1] / view[0]]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.08, .08, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    fw, fh = view
    base = np.zeros((fh, fw, 4), np.float32)
    base[..., 2] = 2400.0
    base[..., 3] = rng.uniform(0, 1, (fh, fw))
    time = 2500.0
    whole = make_shard(n, view, 0, n, cur, prev, base, time)
    shards = []
    for r in range(world):
        row0, rows = sharding.shard_rows(n, world, r)
        shards.append(make_shard(n, view, row0, rows, cur, prev, base, time))
    texels = fw * fh
    chunk = sharding.owner_chunk(texels, world)
    for t in shards + [whole]:
        t.state["autoClearView"] = False
        t.state["autoFade"] = True
    for t in shards:
        sharding.set_owners(t, world)
    for rep in range(2):                       # the second frame blends over the first one's texels and pixels
        whole.renderView = True
        whole.draw()
        want_flow, want_view = whole.flow.read(), whole.read_view()
        for t in shards:
            t.drawFade()
            t.line_widths()
        emitted = [sharding.emit_draw_fragments(t) for t in shards]
        assert all(c.shape[1] == 8 for _, c in emitted)
        assert sum(int(k.numel()) for k, _ in emitted) == whole.fragments > 1000
        sends = [sharding.split_by_owner(k, texels, world) for k, _ in emitted]
        for d, t in enumerate(shards):
            parts_k, parts_c = [], []
            for s, (keys, colors) in enumerate(emitted):
                lo = sum(sends[s][:d])
                parts_k.append(keys[lo:lo + sends[s][d]].clone())
                parts_c.append(colors[lo:lo + sends[s][d]].clone())
            sharding.merge_draw_fragments(t, torch.cat(parts_k).contiguous(), torch.cat(parts_c).contiguous())
        for plane_of in (sharding.flow_view, sharding.view_view):
            planes = [plane_of(t) for t in shards]
            owned = [planes[d][min(d * chunk, texels):min((d + 1) * chunk, texels)].clone() for d in range(world)]
            for v in planes:
                v.copy_(torch.cat(owned))
        torch.cuda.synchronize()
        for t in shards:
            assert bits_equal(t.flow.read(), want_flow).all()
            got = t.read_view()
            assert (got == want_view).all() and got.any()
    # two widths: the passes cannot share a rasterisation, and the entry point says so
    t = shards[0]
    _capi.call("th_line_width_range", t.particles._ctx, 1.0, 8.0)
    _capi.call("th_line_width", t.particles._ctx, _capi.TH_PASS_FLOW, 3.0)
    with pytest.raises(_capi.TendrilsHipError, match="th_deposit_emit and th_view_emit"):
        sharding.emit_draw_fragments(t)
    for t in shards + [whole]:
        t.dispose()


def test_particle_texture_sampling_on_shards_equals_unsharded():
    """Best-sample spawning from the PARTICLE texture (src/demo.main.js:433-441) reads arbitrary particles: a row-band shard
    reads them from a copy of the whole texture (th_state_gather_ptr, filled here by hand - th_state_gather is the same
    copy over RCCL); every band comes out as the band of the unsharded pass, bit for bit.  Without the copy: an error that
    says what to do."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    import tendrils_amd as ta
    from tendrils_amd import _capi, sharding
    from tendrils_amd.spawn import PixelSpawner, data_sample_frag
    n, view, world = 96, (64, 36), 3
    rng = np.random.default_rng(19)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.02, .02, (n, n, 2)) * (rng.random((n, n, 1)) < 0.5)
    base = np.zeros((36, 64, 4), np.float32)

    def spawn(t):
        sp = PixelSpawner(None, dict(shader=data_sample_frag(), buffer=t.particles.buffers[0], spawnSize=[0.8, 0.8], speed=0.01, bias=0.3))
        sp.jitter = [0.003, 0.002]
        t.timer.time = 480.0
        sp.spawn(t)

    whole = make_shard(n, view, 0, n, st, st, base, 500.0)
    spawn(whole)
    want = whole.particles.read(0)
    assert not bits_equal(want, st).all()
    whole.dispose()
    shards = []
    for r in range(world):
        row0, rows = sharding.shard_rows(n, world, r)
        shards.append((row0, rows, make_shard(n, view, row0, rows, st, st, base, 500.0)))
    with pytest.raises(ta.TendrilsHipError) as e:
        spawn(shards[1][2])
    assert "th_state_gather" in str(e.value)
    full = torch.from_numpy(st).cuda().contiguous()
    for row0, rows, t in shards:
        t2 = make_shard(n, view, row0, rows, st, st, base, 500.0)        # (the failed attempt above ticked the timer)
        ptr = C.c_void_p()
        _capi.call("th_state_gather_ptr", t2.particles._ctx, 0, C.byref(ptr))
        sharding.device_view(ptr.value, (n, n, 4), "<f4").copy_(full)
        torch.cuda.synchronize()
        spawn(t2)
        got = t2.particles.read(0)
        assert bits_equal(got, want[row0:row0 + rows]).all()
        t2.dispose()
        t.dispose()
