"""The binned draw() where a line's vertices are NOT the line's own particle, and over packed rings (round 6).

Particles.generateLUT writes vertex coordinates as i/(W-1), j/(2H-1) (src/particles.js:171-190) and state-at-frame.glsl:12-22
turns them back into a texel with fp32 arithmetic: for some shapes - heights such as 100, 1080, 3000; every texture of 8192
and more - the lookup of a few rows / columns lands one texel BESIDE the line's own (th_order.hip: line_rows).  The binned
pipeline walks slots; it finds those other particles through a table of where their rows / columns lie in the slot order
(th::LineSources).  And a packed ring (TH_STATE_F16) is read in place, texel by texel, as what its texels decode to.
Everything here against the restatement (small sizes) and against the stream-ordered pipeline in texel order, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from helpers import bits_equal

pytestmark = pytest.mark.gpu


def state(n, view, seed, spread=0.95):
    rng = np.random.default_rng(seed)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-spread, spread, (n, n, 2)) * [1.0, view[1] / view[0]]
    st[..., 2:] = rng.uniform(-.008, .008, (n, n, 2))
    st[rng.random((n, n)) < 0.03] = [-1e6, -1e6, 0, 0]
    return st


def make(n, view, pipeline, bucket=None, fmt="f32", resort=3):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    opts.update(stateFormat=ta._capi.TH_STATE_F16 if fmt == "f16" else ta._capi.TH_STATE_F32)
    t = ta.Tendrils(View(*view), opts)
    t.resize()
    t.setup(n)
    if bucket is not None:
        t.particles.option("bucket", bucket)
        t.particles.option("resort_steps", resort)
    t.particles.draw_pipeline(pipeline)
    t.timer.time = 3000.0
    t.renderView = True
    return t


def pipeline_of(t):
    from tendrils_amd import _capi
    info = _capi.DrawInfo()
    _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
    return int(info.pipeline)


def lookups_drift(n):
    """the rows whose vertex lookup lands on another row, as the reference's arithmetic has it (fp32)"""
    f = np.float32
    rows = []
    for m in range(n):
        for v in range(2):
            uvy = f((2 * m + v) / (2 * n - 1))
            fl = np.floor(uvy * f(n))
            r = int(np.clip(np.floor((fl / f(n)) * f(n)), 0, n - 1))
            if r != m:
                rows.append(m)
    cols = [i for i in range(n) if int(np.clip(np.floor(f(i / (n - 1)) * f(n)), 0, n - 1)) != i]
    return sorted(set(rows)), cols


@pytest.mark.parametrize("n,slots", [(100, "sorted"), (100, "texel"), (1080, "sorted"), (3000, "sorted")])
def test_frame_loop_over_drifting_rows_bins_equal_restatement_and_stream(oracle, n, slots):
    """heights 100, 1080 and 3000 (2, 22, 177 rows whose vertices read the row beside them).  A loop of tick(); step(); draw() through the bins -
    over tile-sorted slots re-sorted every 3 steps, and in texel order - against the restatement (n = 100) and the
    stream-ordered pipeline, every frame's fragments and the last frame's targets and particles bit for bit."""
    rows, cols = lookups_drift(n)
    assert rows and not cols                                   # (what this size is here for)
    view, frames = ((96, 54) if n > 128 else (64, 36)), (7 if n < 2000 else 4)     # (sorted slots need twice as many particles as target texels)
    if n >= 2000:
        view = (480, 270)
    st = state(n, view, 100 + n)
    a = make(n, view, "bins", bucket=1 if slots == "sorted" else 0)
    b = make(n, view, "stream", bucket=0)
    for t in (a, b):
        t.particles.upload_texels(st)
    size = (1.0, view[0] / view[1])
    cur, prev, flow, time, dt = st.copy(), st.copy(), np.zeros((view[1], view[0], 4), np.float32), 3000.0, 1000.0 / 60.0
    for k in range(frames):
        for t in (a, b):
            t.timer.tick()
            t.step().draw()
        assert a.fragments == b.fragments > 100
        assert pipeline_of(a) == 1 and pipeline_of(b) == 0
        if n <= 128:
            time += dt
            u = oracle.logic_uniforms(n, n, time, dt, view_size=size, **oracle.DEFAULT_STATE)
            prev, cur = cur, oracle.logic_step(u, cur, flow)
            flow, count = oracle.flow_deposit(cur, prev, flow, time, view_size=size, speedLimit=oracle.DEFAULT_STATE["speedLimit"])
            assert a.fragments == count
    if slots == "sorted":
        from tendrils_amd import _capi
        info = _capi.SlotOrderInfo()
        _capi.call("th_slot_order", a.particles._ctx, C.byref(info))
        assert info.sorted_buffers == 2 and info.sorts >= 2    # (the loop never left the sorted order)
    assert bits_equal(a.flow.read(), b.flow.read()).all()
    assert (a.read_view() == b.read_view()).all() and a.read_view().any()
    assert bits_equal(a.particles.read(0), b.particles.read(0)).all() and bits_equal(a.particles.read(1), b.particles.read(1)).all()
    if n <= 128:
        assert bits_equal(a.flow.read(), flow).all() and bits_equal(a.particles.read(0), cur).all()
    a.dispose(); b.dispose()


def test_c4_texture_draws_through_the_bins_by_itself():
    """8192 x 8192 particles (BASELINE config 4 on one GPU): column 8190 and rows 8188-8190 look another texel up.  The default
    policy keeps the frame loop on tile-sorted slots and draws through the bins (round 5: the stream-ordered pipeline, 3.6 ms);
    against the stream-ordered pipeline in texel order: fragments, flow field and view buffer of three frames, bit for bit."""
    n, view, frames = 8192, (1920, 1080), 3
    rows, cols = lookups_drift(n)
    assert rows == [8188, 8189, 8190] and cols == [8190]
    outs = []
    for pipeline in ("auto", "stream"):
        t = make(n, view, pipeline)
        for r0 in range(0, n, 1024):
            rng = np.random.default_rng(9000 + r0)
            band = np.empty((1024, n, 4), np.float32)
            band[..., :2] = rng.uniform(-1, 1, (1024, n, 2))
            band[..., 2:] = rng.uniform(-.01, .01, (1024, n, 2))
            # (the drifting rows and column in the middle of the view, so that their lines are certainly drawn)
            if r0 + 1024 == n:
                band[-4:, :, :2] *= np.float32(0.3)
            band[:, 8185:, :2] *= np.float32(0.3)
            from tendrils_amd import _capi
            for b in (0, 1):
                _capi.call("th_upload_state", t.particles._ctx, b, np.ascontiguousarray(band).ctypes.data_as(_capi._fp), 0, r0, n, 1024)
        band = None
        frags, used = [], []
        for _ in range(frames):
            t.timer.tick()
            t.step().draw()
            frags.append(t.fragments)
            used.append(pipeline_of(t))
        outs.append((frags, used, t.flow.read(), t.read_view()))
        t.dispose()
    (fa, ua, flow_a, view_a), (fb, ub, flow_b, view_b) = outs
    assert ua == [1] * frames and ub == [0] * frames
    assert fa == fb and min(fa) > 10_000_000
    assert bits_equal(flow_a, flow_b).all() and (view_a == view_b).all() and view_a.any()


@pytest.mark.parametrize("n,slots", [(256, "sorted"), (100, "sorted"), (256, "texel")])
def test_packed_ring_draws_through_the_bins_in_place(oracle, n, slots):
    """TH_STATE_F16: the lines are made of what the stored texels decode to.  Through the bins - over sorted slots, the ring
    read in place - against the stream-ordered pipeline, and against the restatement's draw of the decoded texels."""
    view, frames = ((96, 54) if n > 128 else (64, 36)), 5
    st = state(n, view, 7 + n, spread=0.9)
    a = make(n, view, "bins", bucket=1 if slots == "sorted" else 0, fmt="f16")
    b = make(n, view, "stream", bucket=0, fmt="f16")
    for t in (a, b):
        t.particles.upload_texels(st)
    size = (1.0, view[0] / view[1])
    for k in range(frames):
        for t in (a, b):
            t.timer.tick()
            t.step()
        before = b.flow.read()
        for t in (a, b):
            t.draw()
        assert a.fragments == b.fragments > 100 and pipeline_of(a) == 1 and pipeline_of(b) == 0
        if k in (0, frames - 1):                               # the restatement over what the ring decodes to
            cur, prev = b.particles.read(0), b.particles.read(1)
            want, count = oracle.flow_deposit(cur, prev, before, a.timer.time, view_size=size, speedLimit=oracle.DEFAULT_STATE["speedLimit"])
            assert count == a.fragments and bits_equal(a.flow.read(), want).all()
    assert bits_equal(a.flow.read(), b.flow.read()).all() and (a.read_view() == b.read_view()).all() and a.read_view().any()
    assert bits_equal(a.particles.read(0), b.particles.read(0)).all()
    a.dispose(); b.dispose()


def test_spanning_lines_through_a_pool_that_runs_dry_and_lists_that_outgrow_their_pages(oracle, monkeypatch):
    """100 x 100 particles: rows 53 and 59 join unrelated particles - lines across the whole 64 x 36 target, a wave each
    (bins_span_lines).  With a page pool of 4 pages and lists of ONE page at first (two bins of 8000 fragments) the emitting pass runs out of both while those
    lines reserve their places a bin's worth at a time: repeated with a larger pool / a wider table before anything is blended -
    the restatement's result, and the same again on the next draw."""
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    monkeypatch.setenv("TH_BINS_POOL", "4")
    monkeypatch.setenv("TH_BINS_PAGES", "1")
    n, view = 100, (64, 36)
    rng = np.random.default_rng(5)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.1, 0.1, (n, n, 2)) * [1.0, view[1] / view[0]]            # a crowded middle: lists of many pages
    prev[53:61, :, :2] = rng.uniform(-0.95, 0.95, (8, n, 2)) * [1.0, view[1] / view[0]]      # the drifting rows' particles all over the view
    prev[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.15, .15, (n, n, 2)).astype(np.float32)                     # (lines of up to ten texels)
    base = np.zeros((view[1], view[0], 4), np.float32)
    size = (1.0, view[0] / view[1])
    want, frags = oracle.flow_deposit(cur, prev, base, 2500.0, view_size=size)
    again, frags2 = oracle.flow_deposit(cur, prev, want, 2500.0, view_size=size)
    t = ta.Tendrils(View(*view))
    t.resize()
    t.setup(n)
    assert t.particles.option("bins_pool") == 4 and t.particles.option("bins_pages") == 1
    t.particles.option("bucket", 1)
    t.particles.draw_pipeline("bins")
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.flow.set_pixels(base)
    t.timer.time = 2500.0
    t.renderView = False
    t.draw()
    assert t.fragments == frags > 15_000 and pipeline_of(t) == 1
    assert bits_equal(t.flow.read(), want).all()
    t.draw()
    assert t.fragments == frags2 and bits_equal(t.flow.read(), again).all()
    t.dispose()
