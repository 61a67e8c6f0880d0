"""Lines wider than 1 on the GPU (th_line_width / th_line_width_range: gl.lineWidth and ALIASED_LINE_WIDTH_RANGE of
src/index.js:302,336) against the CPU restatement, bit for bit, through both draw() pipelines, both passes, one call and
two, over the host mirrors' flowWidth / lineWidth state.  The restatement's wide line is UNPINNED against the reference (the
captured GL draws every width as 1: tests/test_wide_lines_oracle.py); the default range [1, 1] keeps every capture the pin."""
import ctypes as C

import numpy as np
import pytest

from helpers import bits_equal, deposit_hashed_inputs

pytestmark = pytest.mark.gpu


def make(n, view_res, pipeline, widths=(1, 64), **state):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    opts["lineWidthRange"] = widths
    t = ta.Tendrils(View(*view_res), opts)
    t.resize()
    t.setup(n)
    t.particles.draw_pipeline(pipeline)
    t.state.update(state)
    return t


def load(t, cur, prev, base, time):
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.flow.set_pixels(base)
    t.timer.time = time


@pytest.mark.parametrize("pipeline", ["stream", "bins"])
@pytest.mark.parametrize("width", [2, 3.5, 5, 16])
def test_flow_pass_with_wide_lines_equals_oracle(oracle, pipeline, width):
    n, view = 128, (160, 90)
    cur, prev = deposit_hashed_inputs(n, 31, 1.15, 0.06, 23)             # some lines cross the view's edges
    base = np.zeros((view[1], view[0], 4), np.float32)
    want, frags, cov = oracle.flow_deposit(cur, prev, base, 900.0, view_size=(1.0, view[0] / view[1]), coverage=True, line_width=width)
    thin = oracle.flow_deposit(cur, prev, base, 900.0, view_size=(1.0, view[0] / view[1]))[1]
    assert frags > 1.6 * thin and cov.max() > 3
    t = make(n, view, pipeline, flowWidth=width)
    load(t, cur, prev, base, 900.0)
    t.renderView = False
    t.draw()
    assert t.fragments == frags
    assert bits_equal(t.flow.read(), want).all()
    t.dispose()


@pytest.mark.parametrize("pipeline", ["stream", "bins"])
@pytest.mark.parametrize("flow_w,view_w", [(5, 5), (5, 1), (3, 2), (1, 4)])
def test_both_passes_with_their_own_widths(oracle, pipeline, flow_w, view_w):
    """Tendrils.draw() with renderView: gl.lineWidth(flowWidth) ... gl.lineWidth(lineWidth) - one rasterisation when the two
    agree (th_draw), two when they do not."""
    n, view = 96, (128, 72)
    cur, prev = deposit_hashed_inputs(n, 77, 1.1, 0.07, 19)
    base = np.zeros((view[1], view[0], 4), np.float32)
    vs = (1.0, view[0] / view[1])
    want_flow, frags = oracle.flow_deposit(cur, prev, base, 1200.0, view_size=vs, line_width=flow_w)
    t = make(n, view, pipeline, flowWidth=flow_w, lineWidth=view_w, autoClearView=False, autoFade=False)
    load(t, cur, prev, base, 1200.0)
    t.renderView = True
    t.clearView()
    t.draw()
    u = {k: v for k, v in t.state.items() if k in ("speedLimit", "flowDecay", "speedAlpha", "colorMapAlpha", "baseColor", "flowColor")}
    want_view, vfrags = oracle.view_render(cur, prev, np.zeros((view[1], view[0], 4), np.uint8), 1200.0, view_size=vs, line_width=view_w, **u)
    assert t.fragments == frags
    assert bits_equal(t.flow.read(), want_flow).all()
    assert (t.read_view() == want_view).all() and want_view.any()
    t.dispose()


def test_default_range_clamps_every_width_to_one(oracle):
    """The captured GL's range: flowWidth 5 (the reference's default) draws width-1 lines, as it does there."""
    n, view = 64, (96, 54)
    cur, prev = deposit_hashed_inputs(n, 5, 1.0, 0.05, 17)
    base = np.zeros((view[1], view[0], 4), np.float32)
    want, frags = oracle.flow_deposit(cur, prev, base, 300.0, view_size=(1.0, view[0] / view[1]))
    t = make(n, view, "auto", widths=(1, 1), flowWidth=5, lineWidth=9)
    load(t, cur, prev, base, 300.0)
    t.renderView = False
    t.draw()
    assert t.fragments == frags and bits_equal(t.flow.read(), want).all()
    from tendrils_amd import _capi
    w, d, r = C.c_float(), C.c_float(), (C.c_float * 2)()
    _capi.call("th_line_width_query", t.particles._ctx, _capi.TH_PASS_FLOW, C.byref(w), C.byref(d), r)
    assert (w.value, d.value, list(r)) == (5.0, 1.0, [1.0, 1.0])
    _capi.call("th_line_width_query", t.particles._ctx, _capi.TH_PASS_VIEW, C.byref(w), C.byref(d), r)
    assert (w.value, d.value) == (9.0, 1.0)
    # gl.lineWidth(0) / a negative width: INVALID_VALUE, the state stays
    with pytest.raises(_capi.TendrilsHipError):
        _capi.call("th_line_width", t.particles._ctx, _capi.TH_PASS_FLOW, 0.0)
    with pytest.raises(_capi.TendrilsHipError):
        _capi.call("th_line_width_range", t.particles._ctx, 2.0, 4.0)
    t.state["flowWidth"] = 0
    t.draw()                                                              # the mirror skips the call like the reference's Math.max(0, ...)
    _capi.call("th_line_width_query", t.particles._ctx, _capi.TH_PASS_FLOW, C.byref(w), None, None)
    assert w.value == 5.0
    t.dispose()


def test_wide_lines_in_a_frame_loop_both_pipelines_agree():
    """tick(); step(); draw() with flowWidth 5 honoured: the wake is wider, the loop stays bit-identical between the
    pipelines (the binned one over tile-sorted slots) and differs from the width-1 loop."""
    n, view, frames = 512, (320, 180), 6
    rng = np.random.default_rng(3)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    outs = {}
    for name, pipeline, rng_w in (("stream5", "stream", (1, 64)), ("bins5", "bins", (1, 64)), ("thin", "auto", (1, 1))):
        t = make(n, view, pipeline, widths=rng_w, flowWidth=5, lineWidth=2)
        t.particles.upload_texels(st)
        t.timer.time = 1000.0
        t.renderView = True
        frags = []
        for _ in range(frames):
            t.timer.tick(); t.step(); t.draw()
            frags.append(t.fragments)
        outs[name] = (frags, t.flow.read(), t.read_view(), t.particles.read(0))
        t.dispose()
    a, b, c = outs["stream5"], outs["bins5"], outs["thin"]
    assert a[0] == b[0] and bits_equal(a[1], b[1]).all() and (a[2] == b[2]).all() and bits_equal(a[3], b[3]).all()
    assert min(a[0]) > 2.5 * max(c[0]) and not bits_equal(a[3], c[3]).all()


def test_node_host_draws_wide_lines(oracle):
    """The Node host (tendrils_amd/js): lineWidthRange option, state.flowWidth / lineWidth through N-API."""
    import base64, json, shutil, subprocess
    from helpers import ROOT
    if shutil.which("node") is None:
        pytest.skip("node is not installed")
    n, view = 40, (96, 54)                      # (the state travels in argv: < 128 KB)
    cur, prev = deposit_hashed_inputs(n, 13, 1.1, 0.06, 17)
    vs = (1.0, view[0] / view[1])
    script = """
    const T = require('./tendrils_amd/js');
    const cfg = JSON.parse(process.argv[1]);
    const f32 = (b) => new Float32Array(new Uint8Array(Buffer.from(b, 'base64')).buffer);
    const t = new T.Tendrils({drawingBufferWidth: cfg.view[0], drawingBufferHeight: cfg.view[1]}, {lineWidthRange: [1, 64]});
    t.resize(); t.setup(cfg.n);
    t.viewSize[0] = cfg.viewSize[0]; t.viewSize[1] = cfg.viewSize[1];
    Object.assign(t.state, {flowWidth: 5, lineWidth: 3, autoClearView: false, autoFade: false});
    t.particles.uploadTexels(f32(cfg.cur), 0); t.particles.uploadTexels(f32(cfg.prev), 1);
    t.timer.time = cfg.time;
    t.clearView();
    t.draw();
    const native = require('./tendrils_amd/js/native');
    const out = {flow: Buffer.from(t.flow.read().buffer).toString('base64'), view: Buffer.from(t.readView().buffer).toString('base64'),
                 frags: t.fragments, q0: native.lineWidthQuery(t.particles.handle, 0), q1: native.lineWidthQuery(t.particles.handle, 1)};
    let threw = false;
    try { native.lineWidth(t.particles.handle, 0, -1); } catch (e) { threw = /INVALID_VALUE/.test(e.message); }
    out.threw = threw;
    t.dispose();
    console.log(JSON.stringify(out));
    """
    cfg = dict(n=n, view=view, viewSize=vs, time=700.0,
               cur=base64.b64encode(np.ascontiguousarray(cur, np.float32).tobytes()).decode(),
               prev=base64.b64encode(np.ascontiguousarray(prev, np.float32).tobytes()).decode())
    r = subprocess.run([shutil.which("node"), "-e", script, json.dumps(cfg)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    base = np.zeros((view[1], view[0], 4), np.float32)
    want_flow, frags = oracle.flow_deposit(cur, prev, base, 700.0, view_size=vs, line_width=5)
    want_view, _ = oracle.view_render(cur, prev, np.zeros((view[1], view[0], 4), np.uint8), 700.0, view_size=vs, line_width=3)
    assert res["frags"] == frags and res["threw"]
    assert res["q0"] == {"width": 5, "drawn": 5, "range": [1, 64]} and res["q1"]["drawn"] == 3
    got_flow = np.frombuffer(base64.b64decode(res["flow"]), np.float32).reshape(view[1], view[0], 4)
    got_view = np.frombuffer(base64.b64decode(res["view"]), np.uint8).reshape(view[1], view[0], 4)
    assert bits_equal(got_flow, want_flow).all() and (got_view == want_view).all() and want_view.any()


def test_wide_lines_across_row_band_shards(oracle):
    """Row-band shards (two contexts on this GPU, the exchange by hand as in tests/test_gpu_deposit_sharded.py): every band
    emits its wide lines' fragments, the owners merge them in stream order - the unsharded flow field bit for bit."""
    torch = pytest.importorskip("torch")
    import tendrils_amd as ta
    from tendrils_amd import sharding
    from tendrils_amd.tendrils import View
    n, view, world, width = 64, (96, 54), 2, 4
    cur, prev = deposit_hashed_inputs(n, 21, 1.05, 0.07, 17)
    base = np.zeros((view[1], view[0], 4), np.float32)
    want, frags = oracle.flow_deposit(cur, prev, base, 800.0, view_size=(1.0, view[0] / view[1]), line_width=width)
    shards = []
    for r in range(world):
        row0, rows = sharding.shard_rows(n, world, r)
        opts = ta.defaults()
        opts.update(row0=row0, rows=rows, globalHeight=n, lineWidthRange=(1, 64))
        t = ta.Tendrils(View(*view), opts)
        t.resize(); t.setup(n)
        t.state["flowWidth"] = width
        t.particles.upload_texels(cur[row0:row0 + rows], 0)
        t.particles.upload_texels(prev[row0:row0 + rows], 1)
        t.flow.set_pixels(base)
        t.timer.time = 800.0
        t.line_widths()                                   # (what Tendrils.draw() does first)
        sharding.set_owners(t, world)
        shards.append(t)
    texels = view[0] * view[1]
    chunk = sharding.owner_chunk(texels, world)
    emitted = [sharding.emit_fragments(t) for t in shards]
    assert sum(int(k.numel()) for k, _ in emitted) == frags
    sends = [sharding.split_by_owner(k, texels, world) for k, _ in emitted]
    for d, t in enumerate(shards):
        parts_k, parts_c = [], []
        for s, (keys, colors) in enumerate(emitted):
            lo = sum(sends[s][:d])
            parts_k.append(keys[lo:lo + sends[s][d]].clone())
            parts_c.append(colors[lo:lo + sends[s][d]].clone())
        sharding.merge_fragments(t, torch.cat(parts_k).contiguous(), torch.cat(parts_c).contiguous())
    views = [sharding.flow_view(t) for t in shards]
    owned = [views[d][min(d * chunk, texels):min((d + 1) * chunk, texels)].clone() for d in range(world)]
    for v in views:
        v.copy_(torch.cat(owned))
    torch.cuda.synchronize()
    for t in shards:
        assert bits_equal(t.flow.read(), want).all()
        t.dispose()
