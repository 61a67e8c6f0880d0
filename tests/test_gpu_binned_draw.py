"""The binned draw() pipeline (th_bins.hip: particles walked in slot order, fragments bucketed by 16 x 16-texel bin of the
target, ordered by (texel, stream index) inside each bin) must reproduce GL's primitive order exactly like the
stream-ordered one: the draw suites run with it forced (in texel order, and over tile-sorted slots re-sorted every few
steps: the "bins" / "bins-on-sorted" variants of tests/conftest.py), both pipelines side by side at BASELINE's C3 size, and crowded targets (bins and single texels of more fragments
than LDS holds at once) against the CPU restatement."""
import numpy as np
import pytest

from helpers import bits_equal

pytestmark = pytest.mark.gpu

def make(n, view_res, pipeline):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(*view_res))
    t.resize()
    t.setup(n)
    t.particles.draw_pipeline(pipeline)
    return t


def test_frame_loop_at_c3_size_bins_equal_stream():
    """4096^2 particles over a 1920x1080 target: tick(); step(); draw() (both passes) with the default policy - tile-sorted
    slots, binned pipeline - against the stream-ordered pipeline in texel order: same fragments, same flow field, same view
    buffer, same particles, bit for bit, frame after frame (the deposited wake steers the next step)."""
    n, frames = 4096, 4
    rng = np.random.default_rng(41)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2)).astype(np.float32)
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2)).astype(np.float32)
    st[rng.random((n, n)) < 0.01] = [-1e6, -1e6, 0, 0]
    outs = []
    for pipeline in ("stream", "bins"):
        t = make(n, (1920, 1080), pipeline)
        t.particles.upload_texels(st)
        t.timer.time = 1000.0
        t.renderView = True
        frags = []
        for _ in range(frames):
            t.timer.tick()
            t.step()
            t.draw()
            frags.append(t.fragments)
        order = t.particles.slot_order() if hasattr(t.particles, "slot_order") else None
        outs.append((frags, t.flow.read(), t.read_view(), t.particles.read(0), t.particles.read(1), order))
        t.dispose()
    a, b = outs
    assert a[0] == b[0] and min(a[0]) > 4_000_000
    assert bits_equal(a[1], b[1]).all()
    assert (a[2] == b[2]).all() and a[2].any()
    assert bits_equal(a[3], b[3]).all() and bits_equal(a[4], b[4]).all()


def test_seven_hundred_frames_at_c3_bins_equal_stream():
    """The loop bench.py and tools/deposit_bench.py time, run until the wake has drawn the particles together: after ~650 frames
    a thousand texels of the 1920 x 1080 target receive runs of 1000-17 000 fragments per draw (the giants: parted by stream
    index, ordered window by window, walked by a wave per target) and single bins more than half a million (their page tables
    are widened on the way) - profiles/r4_g_giants.txt.  The default policy (tile-sorted slots, binned pipeline all the way)
    against the stream-ordered pipeline in texel order: the fragments of every frame, and flow field, view buffer and
    particles after the last, bit for bit - a single texel blended in another order anywhere on the way steers the particles
    apart."""
    import ctypes as C
    from benchlib import workload
    from tendrils_amd import _capi
    n, frames = 4096, 700
    workload.N = n
    st = workload.synth_state(0)
    outs = []
    for pipeline in ("stream", "auto"):
        t = make(n, (1920, 1080), pipeline)
        t.particles.upload_texels(st)
        t.timer.time = 1000.0
        t.renderView = True
        frags, used, crowd = [], set(), 0
        info = _capi.DrawInfo()
        for _ in range(frames):
            t.timer.tick()
            t.step()
            t.draw()
            frags.append(t.fragments)
            _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
            used.add(int(info.pipeline))
            crowd = max(crowd, int(info.crowded_fragments))
        outs.append((frags, t.flow.read(), t.read_view(), t.particles.read(0), used, crowd))
        t.dispose()
    a, b = outs
    assert a[4] == {0} and b[4] == {1}                   # stream-ordered all the way / binned all the way
    assert b[5] > 8_000_000                              # (most fragments in crowded bins)
    assert a[0] == b[0] and min(a[0]) > 4_000_000
    assert bits_equal(a[1], b[1]).all()
    assert (a[2] == b[2]).all() and a[2].any()
    assert bits_equal(a[3], b[3]).all()


def test_loop_on_a_crowded_target_bins_equal_stream():
    """A million particles over a 160 x 90 target: every texel holds a run of dozens of fragments from the first frame on,
    hundreds and thousands (the long list, the giants, their windows) once the wake has pulled the particles together.  30
    frames of tick(); step(); draw() with both passes: the binned pipeline over tile-sorted slots against the stream-ordered
    one in texel order - fragments, flow field, view buffer and particles bit for bit, frame after frame."""
    import ctypes as C
    from tendrils_amd import _capi
    n, frames = 1024, 30
    rng = np.random.default_rng(77)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2)).astype(np.float32) * np.float32(0.7)
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2)).astype(np.float32)
    outs = []
    for pipeline in ("stream", "bins"):
        t = make(n, (160, 90), pipeline)
        t.particles.upload_texels(st)
        t.timer.time = 1000.0
        t.renderView = True
        frags, crowd = [], []
        for _ in range(frames):
            t.timer.tick()
            t.step()
            t.draw()
            frags.append(t.fragments)
        info = _capi.DrawInfo()
        _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
        outs.append((frags, t.flow.read(), t.read_view(), t.particles.read(0), int(info.crowded_fragments)))
        t.dispose()
    a, b = outs
    assert a[0] == b[0] and min(a[0]) > 100_000
    assert b[4] > 0.3 * b[0][-1]                     # (the crowded bins' kernels did run: a third of the last draw's fragments and more)
    assert bits_equal(a[1], b[1]).all()
    assert (a[2] == b[2]).all() and a[2].any()
    assert bits_equal(a[3], b[3]).all()


def crowded(n, seed, spread):
    rng = np.random.default_rng(seed)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-spread, spread, (n, n, 2))
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.03, .03, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    return cur, prev


@pytest.mark.parametrize("n,spread,view", [(256, 0.25, (64, 36)),      # bins of > 4096 fragments: groups of texels
                                             (512, 0.02, (64, 36))],     # single texels of > 4096 fragments: id windows
                         ids=["crowded_bins", "giant_texels"])
def test_crowded_targets_are_order_exact(oracle, n, spread, view):
    cur, prev = crowded(n, 91, spread)
    base = np.zeros((view[1], view[0], 4), np.float32)
    want, frags, cov = oracle.flow_deposit(cur, prev, base, 2500.0, view_size=(1.0, view[0] / view[1]), coverage=True)
    assert cov.max() > (4096 if spread < 0.1 else 150)
    for pipeline in ("bins", "stream"):
        t = make(n, view, pipeline)
        t.particles.upload_texels(cur, 0)
        t.particles.upload_texels(prev, 1)
        t.flow.set_pixels(base)
        t.timer.time = 2500.0
        t.draw()
        got, nf = t.flow.read(), t.fragments
        t.renderView = True                          # and both targets in one call on top
        t.draw()
        view_px = t.read_view()
        t.dispose()
        assert nf == frags
        assert bits_equal(got, want).all(), pipeline
        if pipeline == "bins":
            first_view = view_px
        else:
            assert (view_px == first_view).all() and view_px.any()


def test_pool_growth_and_overfull_bins_exactly(oracle, monkeypatch):
    """The binned pass hands pages out from a pool sized from experience: a pool that runs dry is grown and the pass repeated
    before anything is blended (TH_BINS_POOL=8 forces it); a bin that outgrows its lists (more than half a million
    fragments in 16 x 16 texels - where the wake of a long-running loop ends up) gets a page table four times as wide and the
    pass is repeated; and a table that may not be widened (option bins_pages < 0) leaves the draw to the stream-ordered
    pipeline - all with the exact result."""
    import ctypes as C
    import tendrils_amd as ta
    from tendrils_amd import _capi
    from tendrils_amd.tendrils import View
    monkeypatch.setenv("TH_BINS_POOL", "8")              # (a context's switches start from the environment it is created in)

    def run(n, spread, view, seed, pages=0, pipeline=1):        # (pipeline: TH_DRAW_BINS = 1, TH_DRAW_STREAM = 0)
        rng = np.random.default_rng(seed)
        prev = np.zeros((n, n, 4), np.float32)
        prev[..., :2] = rng.uniform(-spread, spread, (n, n, 2))
        prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
        cur = prev.copy()
        cur[..., :2] += rng.uniform(-.03, .03, (n, n, 2)).astype(np.float32)
        base = np.zeros((view[1], view[0], 4), np.float32)
        want, frags = oracle.flow_deposit(cur, prev, base, 2500.0, view_size=(1.0, view[0] / view[1]))
        t = ta.Tendrils(View(*view))
        t.resize()
        t.setup(n)
        assert t.particles.option("bins_pool") == 8
        if pages:
            t.particles.option("bins_pages", pages)
        t.particles.draw_pipeline("bins")
        t.particles.upload_texels(cur, 0)
        t.particles.upload_texels(prev, 1)
        t.flow.set_pixels(base)
        t.timer.time = 2500.0
        t.renderView = False
        t.draw()
        assert t.fragments == frags, (t.fragments, frags)
        assert bits_equal(t.flow.read(), want).all()
        info = _capi.DrawInfo()
        _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
        assert info.pipeline == pipeline, (info.pipeline, pipeline)
        t.draw()                                  # (and again: the store is clean after a repeated / abandoned pass)
        t.dispose()
        return frags
    a = run(256, 0.2, (96, 54), 5)               # a few bins, hundreds of fragments per list, a pool of 8 pages: grown, pass repeated
    b = run(1536, 0.004, (64, 36), 6)            # 1.2 M drawable lines inside one bin: the bin outgrows its lists, the table is widened
    assert a > 20_000 and b > 600_000, (a, b)
    assert run(256, 0.2, (96, 54), 5, pages=2) == a                 # lists of 2 pages at first: widened
    assert run(256, 0.02, (96, 54), 7, pages=-2, pipeline=0) > 20_000      # ... and never: the stream-ordered pipeline draws


def test_giant_runs_whose_stream_indices_lie_side_by_side(oracle):
    """A texel that receives thousands of fragments has its run parted by the leading bits of the stream indices it holds,
    ordered window by window and walked by a wave per target (th_bins.hip: giant_*_kernel).  Three such texels of a
    4096 x 4096 state's lines: 9000 particles from all over the state (indices spread over all sixteen million); the
    NEIGHBOURS of columns 2000-2002 (12 288 indices in a row, half of them rows that draw: the buckets cover just that range);
    and columns 0-2 together with columns 4000-4002 - two dense clusters sixteen million apart: a thousand buckets over that
    range put each cluster into one, more than a window holds, and the run goes to the workgroup that narrows its windows as
    it goes (crowd_blend_kernel)."""
    import ctypes as C
    import tendrils_amd as ta
    from tendrils_amd import _capi
    from tendrils_amd.tendrils import View
    n, view = 4096, (64, 36)
    rng = np.random.default_rng(11)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = 5.0                                           # everybody else: outside the view

    def gather(rows, cols, x, y):                                 # (the middle of a texel; lines a tenth of a texel long)
        prev[rows, cols, 0] = x + rng.uniform(-5e-4, 5e-4, np.broadcast(rows, cols).shape)
        prev[rows, cols, 1] = y + rng.uniform(-5e-4, 5e-4, np.broadcast(rows, cols).shape)
    every = np.arange(n)[:, None]
    gather(every, np.array([0, 1, 2, 4000, 4001, 4002])[None, :], 0.296875, 0.17)
    gather(every, np.array([2000, 2001, 2002])[None, :], 0.609375, -0.390625)
    taken = np.zeros((n, n), bool)
    taken[:, [0, 1, 2, 2000, 2001, 2002, 4000, 4001, 4002]] = True
    free = np.flatnonzero(~taken.ravel())
    far = rng.choice(free, 9000, replace=False)
    gather(far // n, far % n, -0.421875, -0.23)
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-1e-3, 1e-3, (n, n, 2)).astype(np.float32)
    base = np.zeros((view[1], view[0], 4), np.float32)
    want, frags = oracle.flow_deposit(cur, prev, base, 2500.0, view_size=(1.0, view[0] / view[1]))
    assert frags > 15000 and np.count_nonzero(np.abs(want).sum(-1)) == 3         # three texels, thousands of fragments each
    t = ta.Tendrils(View(*view))
    t.resize()
    t.setup(n)
    t.particles.draw_pipeline("bins")
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.flow.set_pixels(base)
    t.timer.time = 2500.0
    t.renderView = False
    t.draw()
    assert t.fragments == frags, (t.fragments, frags)
    assert bits_equal(t.flow.read(), want).all()
    info = _capi.DrawInfo()
    _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
    assert info.pipeline == 1 and info.crowded_fragments == frags, (info.pipeline, info.crowded_fragments)     # (all three bins are crowded ones)
    t.dispose()


def test_auto_keeps_a_crowded_target_on_the_binned_pipeline():
    """TH_DRAW_AUTO stays with the bins however crowded the target is (th_draw_query: nearly every fragment in bins of more than
    4096) - with the same results as a context that used the stream-ordered pipeline all along, bit for bit (option bucket =
    1: sorted slots, hence the binned pipeline, at this small size)."""
    import ctypes as C
    import tendrils_amd as ta
    from tendrils_amd import _capi
    from tendrils_amd.tendrils import View
    n, view = 512, (96, 54)
    rng = np.random.default_rng(3)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-0.12, 0.12, (n, n, 2))            # everybody inside a few bins of the target (share ~1)
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    outs = []
    for pipeline in ("auto", "stream"):
        t = ta.Tendrils(View(*view))
        t.resize()
        t.setup(n)
        t.particles.option("bucket", 1)
        t.particles.upload_texels(st)
        t.particles.draw_pipeline(pipeline)
        t.timer.time = 1000.0
        t.state["noiseWeight"] = 0.0005
        info, used, share = _capi.DrawInfo(), [], []
        for _ in range(8):
            t.timer.tick()
            t.step()
            t.draw()
            _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
            used.append(info.pipeline)
            share.append(info.crowded_fragments / max(info.fragments, 1))
        outs.append((used, share, t.flow.read(), t.read_view(), t.particles.read(0)))
        t.dispose()
    (a_used, a_share, *a), (s_used, s_share, *s) = outs
    assert a_used == [1] * 8 and min(a_share) > 0.8 and s_used == [0] * 8, (a_used, a_share)
    assert bits_equal(a[0], s[0]).all() and (a[1] == s[1]).all() and a[1].any() and bits_equal(a[2], s[2]).all()
