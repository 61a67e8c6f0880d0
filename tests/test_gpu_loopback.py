"""The sharded draw()'s exchange logic with MORE THAN ONE RANK on one GPU.  RCCL refuses two ranks on one device, so the
ranks here are contexts of this process joined by the in-process transport (th_comm_loopback_id; th_loopback.hip), each
driven by a thread of its own - everything above the byte transport is the code an RCCL job runs (th_shard.hip): the
owners' bounds, the count exchange, the fragment all-to-all, the merge, the all-gather of owned texel ranges that are NOT
all the same size (the path RCCL serves with grouped broadcasts), the edge-row exchange, the agreement on rank-local
failures.  Every rank's flow texture and view buffer must equal the unsharded draw() bit for bit."""
import ctypes as C
import threading

import numpy as np
import pytest

from helpers import bits_equal

pytestmark = pytest.mark.gpu


def inputs(n, view, seed, spread=0.9):
    rng = np.random.default_rng(seed)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-spread, spread, (n, n, 2)) * [1.0, view[1] / view[0]]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.08, .08, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    fw, fh = view
    base = np.zeros((fh, fw, 4), np.float32)
    base[..., :2] = rng.uniform(-.01, .01, (fh, fw, 2))
    base[..., 2] = 2400.0
    base[..., 3] = rng.uniform(0, 1, (fh, fw))
    return cur, prev, base


def make(n, view, cur, prev, base, band=None, fmt="f32", widths=None):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    if widths:                                      # a GL that honours gl.lineWidth up to 2: (flowWidth, lineWidth)
        opts["lineWidthRange"] = (1, 2)
        opts["state"]["flowWidth"], opts["state"]["lineWidth"] = widths
    row0, rows = band if band else (0, n)
    opts.update(row0=row0, rows=rows, globalHeight=n, stateFormat=ta._capi.TH_STATE_F16 if fmt == "f16" else ta._capi.TH_STATE_F32)
    t = ta.Tendrils(View(*view), opts)
    t.resize()
    t.setup(n)
    t.particles.upload_texels(cur[row0:row0 + rows], 0)
    t.particles.upload_texels(prev[row0:row0 + rows], 1)
    t.flow.set_pixels(base)
    t.timer.time = 2500.0
    t.state["baseColor"] = [1, 0.7, 0.3, 0.2]
    t.state["flowColor"] = [0.2, 1, 0.9, 0.1]
    return t


def in_threads(world, body):
    """body(rank) on a thread per rank (a ctypes call releases the GIL: the ranks really meet inside the library)"""
    out, err = [None] * world, [None] * world

    def run(r):
        try:
            out[r] = body(r)
        except BaseException as e:          # noqa: BLE001 - handed to the main thread
            err[r] = e
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not any(th.is_alive() for th in threads), "a rank is still waiting inside a collective"
    return out, err


def world_of(n, view, world, cur, prev, base, fmt="f32", pipeline=None):
    from tendrils_amd import sharding
    ident = sharding.loopback_id()
    shards = [make(n, view, cur, prev, base, sharding.shard_rows(n, world, r), fmt) for r in range(world)]
    if pipeline:
        for t in shards:
            t.particles.draw_pipeline(pipeline)
    _, err = in_threads(world, lambda r: sharding.comm_join(shards[r].particles._ctx, ident, r, world))
    assert err == [None] * world, err
    assert all(sharding.comm_query(t.particles._ctx)["world"] == world for t in shards)
    return shards


@pytest.mark.parametrize("n,view,world,fmt", [(64, (96, 54), 2, "f32"), (100, (50, 27), 3, "f32"), (128, (50, 27), 4, "f32"),
                                              (100, (50, 27), 4, "f16"), (64, (96, 54), 2, "f16")])
def test_draw_sharded_over_loopback_equals_unsharded(n, view, world, fmt):
    """n = 100: bands of 34 / 33 / 33 rows, and a height whose vertex lookup lands one row beside a line's own - the edge
    rows of the neighbours are needed (f16: through the f32 views of the packed ring).  50 x 27 texels over 3 or 4 owners:
    the last owner's range is shorter than the others' (the unequal all-gather)."""
    from tendrils_amd import sharding
    cur, prev, base = inputs(n, view, 7 * n + world)
    one = make(n, view, cur, prev, base, None, fmt)
    if fmt == "f16":                               # what the packed texels decode to is what everybody draws
        cur, prev = one.particles.read(0), one.particles.read(1)
    one.draw()
    want_flow, want_view, want_frags = one.flow.read(), one.read_view(), one.fragments
    one.dispose()
    assert want_frags > 1000 and want_view.any()
    shards = world_of(n, view, world, cur, prev, base, fmt)
    for frame in range(2):                         # (the second draw: grown buffers are reused, the agreement of the first call is not repeated)
        frags, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
        assert err == [None] * world, err
        if frame == 0:
            assert sum(frags) == want_frags
            for t in shards:
                assert bits_equal(t.flow.read(), want_flow).all()
                assert (t.read_view() == want_view).all()
    for t in shards:
        t.dispose()


def last_pipeline(t):
    from tendrils_amd import _capi
    info = _capi.DrawInfo()
    _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
    return info.pipeline


@pytest.mark.parametrize("n,view,world,spread,sorted_slots,pool,pages", [(64, (96, 54), 2, 0.9, False, 0, 0), (128, (50, 27), 3, 0.9, True, 0, 0),
                                                                         (256, (96, 54), 4, 0.15, True, 0, 0), (256, (40, 200), 3, 0.9, False, 0, 0),
                                                                         (256, (96, 54), 2, 0.15, False, 8, 0), (256, (96, 54), 3, 0.3, True, 40, 0), (256, (96, 54), 4, 0.12, True, 8, 0),
                                                                         (256, (96, 54), 3, 0.15, True, 8, 2), (256, (96, 54), 2, 0.3, False, 0, 4),
                                                                         (1024, (160, 90), 2, 0.3, True, 0, 0)])
def test_draw_sharded_through_the_bins_equals_unsharded(n, view, world, spread, sorted_slots, pool, pages):
    """The sharded draw() through the binned pipeline (th_bins.hip: the bins travel to the ranks that own them): every rank
    rasterises into its own page store, whole bin rows change hands with their counts, the owner lays them out as if it had
    emitted them and blends with the unchanged kernels.  spread 0.15 at 256^2: bins of tens of thousands of fragments (pool
    pages in the owner's layout, the crowded bins' kernels); 40 x 200 texels over 3 owners: 13 bin rows, 4 / 4 / 5 each;
    sorted_slots: after a step over tile-sorted slots (the bands keep their order: no return to texel order); pool: a page
    pool of that many pages to start with - it runs dry in the emitting pass (repeated with a larger one) and again when the
    other ranks' fragments arrive (grown with the owner's own bins kept in it); pages: lists that can grow to that many pages
    at first - a bin outgrows them in the emitting pass (wider table, pass repeated) and again at its owner (widened with the
    entries of the owner's own fragments kept); a million particles over 160 x 90 texels at spread 0.3: texels of thousands of
    fragments - the giants' kernels (parted, ordered window by window, walked by a wave per target) over bins that were put
    together from two ranks' fragments."""
    from tendrils_amd import sharding
    cur, prev, base = inputs(n, view, 31 * n + world, spread)
    one = make(n, view, cur, prev, base)
    shards = world_of(n, view, world, cur, prev, base, pipeline="bins")
    everybody = [one] + shards
    for t in shards:
        if pool:
            t.particles.option("bins_pool", pool)
        if pages:
            t.particles.option("bins_pages", pages)
    if sorted_slots:
        for t in everybody:
            t.particles.option("bucket", 1)
            t.state["noiseWeight"] = 0.0005
    for frame in range(3):
        for t in everybody:
            t.timer.tick()
            if sorted_slots:
                t.step()
        one.draw()
        frags, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
        assert err == [None] * world, err
        assert sum(frags) == one.fragments > 1000
        want_flow, want_view = one.flow.read(), one.read_view()
        for t in shards:
            assert last_pipeline(t) == 1                       # TH_DRAW_BINS
            assert bits_equal(t.flow.read(), want_flow).all()
            assert (t.read_view() == want_view).all()
    if sorted_slots:                                           # the bands stepped over sorted slots all along
        info = __import__("tendrils_amd")._capi.SlotOrderInfo()
        __import__("tendrils_amd")._capi.call("th_slot_order", shards[0].particles._ctx, C.byref(info))
        assert info.sorted_buffers == 2
    for t in everybody:
        t.dispose()


@pytest.mark.parametrize("n,world,fmt,sorted_slots", [(100, 3, "f32", True), (100, 4, "f32", False), (100, 3, "f16", True), (128, 3, "f16", True),
                                                      (128, 2, "f16", False)])
def test_sharded_bins_over_drifting_rows_and_packed_rings(n, world, fmt, sorted_slots):
    """Round 6: the sharded draw() goes through the bins for the shapes whose vertex lookup lands beside a line's own texel -
    n = 100: rows 53 and 59 read the row above, and with bands of 34 / 33 / 33 (or 25) rows some of those lookups cross into
    the neighbouring band: every band picks its edge rows out of its slot order (th::LineSources) and sends them on - and for
    packed rings, read in place.  Bands stepping over tile-sorted slots, three frames, every rank bit for bit the unsharded
    draw (which takes the same pipeline over its own slots)."""
    from tendrils_amd import sharding
    view = (48, 27)                                # (a band of 33 x 100 particles steps over sorted slots: twice the target's texels)
    cur, prev, base = inputs(n, view, 13 * n + world)
    one = make(n, view, cur, prev, base, None, fmt)
    if fmt == "f16":                               # what the packed texels decode to is what everybody starts from
        cur, prev = one.particles.read(0), one.particles.read(1)
    shards = world_of(n, view, world, cur, prev, base, fmt, pipeline="bins")
    one.particles.draw_pipeline("bins")
    everybody = [one] + shards
    for t in everybody:
        t.particles.option("bucket", 1 if sorted_slots else 0)
        t.particles.option("resort_steps", 2)
        t.state["noiseWeight"] = 0.0005
    for frame in range(3):
        for t in everybody:
            t.timer.tick()
            t.step()
        one.draw()
        frags, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
        assert err == [None] * world, err
        assert sum(frags) == one.fragments > 1000
        want_flow, want_view = one.flow.read(), one.read_view()
        for t in shards:
            assert last_pipeline(t) == 1                       # TH_DRAW_BINS
            assert bits_equal(t.flow.read(), want_flow).all()
            assert (t.read_view() == want_view).all()
    if sorted_slots:
        info = __import__("tendrils_amd")._capi.SlotOrderInfo()
        __import__("tendrils_amd")._capi.call("th_slot_order", shards[0].particles._ctx, C.byref(info))
        assert info.sorted_buffers == 2
    whole = np.concatenate([t.particles.read(0) for t in shards])
    assert bits_equal(whole, one.particles.read(0)).all()
    for t in everybody:
        t.dispose()


@pytest.mark.parametrize("bands,pipeline,fmt", [([(0, 53), (53, 47)], "bins", "f32"), ([(0, 53), (53, 6), (59, 41)], "bins", "f32"),
                                                ([(0, 53), (53, 6), (59, 41)], "stream", "f32"), ([(0, 53), (53, 47)], "bins", "f16")])
def test_a_drifting_row_at_the_edge_of_its_band_reads_the_neighbours_last_row(bands, pipeline, fmt):
    """100 x 100 particles: the vertices of rows 53 and 59 read the row ABOVE.  With bands that begin exactly there the row a line
    reads is the LAST row of the rank below: it arrives as a halo row - through the bins picked out of that rank's slot order
    (bins_edge_rows_kernel over th::LineSources), through the stream-ordered pass straight from texel order - and the lines of a
    band's first row are made of another rank's particles.  (Balanced bands of 34 / 33 / 33 rows keep both lookups inside band 1:
    the other tests exchange edge rows nobody reads.)"""
    from tendrils_amd import sharding
    n, view, world = 100, (16, 9), len(bands)
    cur, prev, base = inputs(n, view, 1234 + world)
    one = make(n, view, cur, prev, base, None, fmt)
    if fmt == "f16":
        cur, prev = one.particles.read(0), one.particles.read(1)
    ident = sharding.loopback_id()
    shards = [make(n, view, cur, prev, base, band, fmt) for band in bands]
    _, err = in_threads(world, lambda r: sharding.comm_join(shards[r].particles._ctx, ident, r, world))
    assert err == [None] * world, err
    everybody = [one] + shards
    for t in everybody:
        t.particles.draw_pipeline(pipeline)
        t.particles.option("bucket", 1 if pipeline == "bins" else 0)
        t.particles.option("resort_steps", 2)
        t.state["noiseWeight"] = 0.0005
    for frame in range(3):
        for t in everybody:
            t.timer.tick()
            t.step()
        one.draw()
        frags, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
        assert err == [None] * world, err
        assert sum(frags) == one.fragments > 500
        want_flow, want_view = one.flow.read(), one.read_view()
        for t in shards:
            assert last_pipeline(t) == (1 if pipeline == "bins" else 0)
            assert bits_equal(t.flow.read(), want_flow).all()
            assert (t.read_view() == want_view).all()
    # (the lines in question do draw: without the halo rows the emit refuses the pass - "looks up a particle row outside the band")
    for t in everybody:
        t.dispose()


@pytest.mark.parametrize("how", ["flow pass only", "two widths"])
def test_sharded_bins_pass_by_pass(how):
    """Tendrils.draw() of band contexts (the library's exchange, through the bins) when the passes do not share one
    rasterisation: the flow pass alone (renderView off: mode 0), and lines 2 wide in the flow pass but 1 wide in the view -
    two sharded passes, the second with the view's colours only (mode 1)."""
    from tendrils_amd import sharding
    n, view, world = 128, (96, 54), 3
    widths = (2, 1) if how == "two widths" else None
    cur, prev, base = inputs(n, view, 23)
    one = make(n, view, cur, prev, base, widths=widths)
    ident = sharding.loopback_id()
    shards = [make(n, view, cur, prev, base, sharding.shard_rows(n, world, r), widths=widths) for r in range(world)]
    _, err = in_threads(world, lambda r: sharding.comm_join(shards[r].particles._ctx, ident, r, world))
    assert err == [None] * world, err
    for t in [one] + shards:
        t.particles.draw_pipeline("bins")
        t.renderView = how != "flow pass only"
    one.draw()
    _, err = in_threads(world, lambda r: shards[r].draw())
    assert err == [None] * world, err
    want_flow, want_view = one.flow.read(), one.read_view()
    assert one.fragments > 1000 and want_view.any() == (how != "flow pass only")
    for t in shards:
        assert last_pipeline(t) == 1
        assert bits_equal(t.flow.read(), want_flow).all() and (t.read_view() == want_view).all()
    for t in [one] + shards:
        t.dispose()


@pytest.mark.parametrize("n,spread", [(128, 0.9), (256, 0.15)])
def test_one_rank_giving_up_on_the_bins_sends_everybody_to_the_stream_ordered_pass(n, spread):
    """A rank whose binned pass cannot go on (no store, a bin beyond its lists' reach: here TH_OPT_INJECT_FAILURE = 4) says so
    with its counts; every rank then runs the stream-ordered pass for that draw - same result, nobody waits.  The other
    ranks had emitted into their stores by then and nothing blended them: the next binned draw - of ANOTHER state - must not
    meet what they left (spread 0.15 at 256^2: lists of many pages, whose stale page ids a later pass would follow)."""
    from tendrils_amd import sharding
    view, world = (96, 54), 3
    cur, prev, base = inputs(n, view, 17, spread)
    cur2, prev2, _ = inputs(n, view, 18, spread * 1.5)
    one = make(n, view, cur, prev, base)
    one.draw()
    want_flow, want_view = one.flow.read(), one.read_view()
    one.flow.set_pixels(base)
    one.clearView()
    one.particles.upload_texels(cur2, 0)
    one.particles.upload_texels(prev2, 1)
    one.draw()
    want_flow2, want_view2 = one.flow.read(), one.read_view()
    one.dispose()
    assert not bits_equal(want_flow, want_flow2).all()
    shards = world_of(n, view, world, cur, prev, base, pipeline="bins")
    shards[2].particles.option("inject_failure", 4)
    frags, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
    assert err == [None] * world, err
    for t in shards:
        assert last_pipeline(t) == 0                           # TH_DRAW_STREAM: the fallback drew
        assert bits_equal(t.flow.read(), want_flow).all() and (t.read_view() == want_view).all()
    for r, t in enumerate(shards):                             # ... and the next draw goes through the bins again
        row0, rows = sharding.shard_rows(n, world, r)
        t.flow.set_pixels(base)
        t.clearView()
        t.particles.upload_texels(cur2[row0:row0 + rows], 0)
        t.particles.upload_texels(prev2[row0:row0 + rows], 1)
    _, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
    assert err == [None] * world, err
    for t in shards:
        assert last_pipeline(t) == 1
        assert bits_equal(t.flow.read(), want_flow2).all() and (t.read_view() == want_view2).all()
        t.dispose()


def test_gather_and_counters_over_loopback_with_unequal_bands():
    """th_state_gather (bands of 34 / 33 / 33 rows: parts of different sizes) and the counter all-reduce"""
    from tendrils_amd import _capi, sharding
    n, view, world = 100, (50, 27), 3
    cur, prev, base = inputs(n, view, 11)
    one = make(n, view, cur, prev, base)
    want = one.particles.stats(0.01)
    one.dispose()
    shards = world_of(n, view, world, cur, prev, base)

    def body(r):
        ctx = shards[r].particles._ctx
        _capi.call("th_state_gather", ctx, 0)
        p = C.c_void_p()
        _capi.call("th_state_gather_ptr", ctx, 0, C.byref(p))            # (the same copy: its address)
        g = _capi.Counters()
        _capi.call("th_stats_global", ctx, C.c_float(0.01), C.byref(g))
        return {k: getattr(g, k) for k, _ in _capi.Counters._fields_}
    got, err = in_threads(world, body)
    assert err == [None] * world, err
    for g in got:
        assert {k: v for k, v in g.items() if k != "sum_speed"} == {k: v for k, v in want.items() if k != "sum_speed"}
        assert abs(g["sum_speed"] - want["sum_speed"]) <= 1e-9 * want["sum_speed"]        # (a sum of three partial sums)
    assert got[0] == got[1] == got[2]
    for t in shards:                                # every rank holds the whole texture
        whole = sharding.device_view(_ptr(t), (n, n, 4), "<f4")
        assert bits_equal(whole.cpu().numpy(), cur).all()
        t.dispose()


def _ptr(t):
    from tendrils_amd import _capi
    p = C.c_void_p()
    _capi.call("th_state_gather_ptr", t.particles._ctx, 0, C.byref(p))
    return p.value


@pytest.mark.parametrize("stage,fmt", [(2, "f32"), (3, "f32"), (1, "f16")])
def test_a_rank_local_failure_ends_the_draw_on_every_rank(stage, fmt):
    """One rank fails on its own (TH_OPT_INJECT_FAILURE: while preparing its edge rows / rasterising / making room): it reports
    its error, every other rank reports that a peer failed - nobody is left waiting in a collective, nothing is blended -
    and the next draw of the same world works."""
    import tendrils_amd as ta
    from tendrils_amd import sharding
    n, view, world, bad = 64, (96, 54), 3, 1
    cur, prev, base = inputs(n, view, 5)
    one = make(n, view, cur, prev, base, None, fmt)
    if fmt == "f16":
        cur, prev = one.particles.read(0), one.particles.read(1)
    one.draw()
    want_flow, want_view = one.flow.read(), one.read_view()
    one.dispose()
    shards = world_of(n, view, world, cur, prev, base, fmt)
    _, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))      # (first call: fixed buffers agreed)
    assert err == [None] * world, err
    for t in shards:                                # back to the start
        t.flow.set_pixels(base)
        t.clearView()
    shards[bad].particles.option("inject_failure", stage)
    _, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
    assert all(isinstance(e, ta.TendrilsHipError) for e in err), err
    assert "injected failure" in str(err[bad])
    for r in range(world):
        if r != bad:
            assert "rank %d failed" % bad in str(err[r]), str(err[r])
    for t in shards:
        assert bits_equal(t.flow.read(), base).all() and not t.read_view().any()
    assert shards[bad].particles.option("inject_failure") == 0
    _, err = in_threads(world, lambda r: sharding.draw_sharded_native(shards[r], view=True))
    assert err == [None] * world, err
    for t in shards:
        assert bits_equal(t.flow.read(), want_flow).all() and (t.read_view() == want_view).all()
        t.dispose()


def test_a_collective_nobody_else_joins_times_out(monkeypatch):
    import tendrils_amd as ta
    from tendrils_amd import _capi
    monkeypatch.setenv("TH_LOOPBACK_TIMEOUT_MS", "300")
    cur, prev, base = inputs(64, (96, 54), 2)
    shards = world_of(64, (96, 54), 2, cur, prev, base)
    with pytest.raises(ta.TendrilsHipError) as e:
        _capi.call("th_stats_global", shards[0].particles._ctx, C.c_float(0.01), C.byref(_capi.Counters()))
    assert "did not arrive" in str(e.value)
    for t in shards:
        t.dispose()
