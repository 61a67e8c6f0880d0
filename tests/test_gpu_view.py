"""The view pass on the GPU (th_view_draw behind Tendrils.draw()): the RGBA8 image must equal the restatement's byte for
byte (same arithmetic, same stream-order blending), which in turn is pinned to the reference's captured view render
(tests/test_view_oracle.py); export_lines(view=True) carries the same vertex colours."""
import base64
import json
import shutil
import subprocess

import numpy as np
import pytest

from helpers import ROOT, bits_equal, golden
from test_view_oracle import check_against_reference, view_fixture

pytestmark = pytest.mark.gpu


def make(m, n):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(*m["viewRes"]))
    t.resize()
    t.setup(n)
    t.viewSize[:] = m["viewSize"]
    for k, v in m["render"].items():
        if k != "lineWidth":
            t.state[k] = v
    return t


@pytest.mark.parametrize("path", golden("view"), ids=lambda p: p.split("/")[-1][:-4])
def test_view_draw_byte_exact_to_oracle_reference_coverage_exact_colours_toleranced(oracle, path):
    m, cur, prev, ref = view_fixture(path)
    t = make(m, cur.shape[0])
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.timer.time = m["time"]
    t.draw()
    got, lines = t.read_view(), t.export_lines(view=True)
    frags = t.view_fragments
    t.dispose()
    fh, fw = ref.shape[:2]
    want, n = oracle.view_render(cur, prev, np.zeros((fh, fw, 4), np.uint8), m["time"], view_size=m["viewSize"], **m["render"])
    assert frags == n
    assert (got == want).all()
    check_against_reference(got, ref)
    want_lines = oracle.export_view_lines(cur, prev, m["time"], view_size=m["viewSize"], **m["render"])
    assert lines.shape == want_lines.shape and bits_equal(lines, want_lines).all()


def test_view_accumulates_over_frames_with_fade_and_colormap(oracle):
    """Two draw() calls into the same view with a translucent fade between them and a position-dependent colour map."""
    n, view = 32, (64, 36)
    rng = np.random.default_rng(5)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.8, 0.8, (n, n, 2)) * [1.0, 36 / 64]
    prev[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.08, .08, (n, n, 2)).astype(np.float32)
    cmap = rng.uniform(0, 1, (4, 8, 4)).astype(np.float32)
    m = dict(viewRes=list(view), viewSize=[1.0, 64 / 36], render=dict(speedLimit=0.01, flowDecay=0.005, speedAlpha=0.5,
             colorMapAlpha=0.4, baseColor=[1, 1, 1, 0.5], flowColor=[1, 1, 1, 0.04]))
    t = make(m, n)
    t.colorMap.set_pixels(cmap)
    t.state["fadeColor"] = [0.1333, 0.1333, 0.1333, 0.3]
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    want = np.zeros((view[1], view[0], 4), np.uint8)
    for time in (1000.0, 1016.5):
        t.timer.time = time
        t.draw()
        want = oracle.view_fill(want, t.state["fadeColor"])
        want, _ = oracle.view_render(cur, prev, want, time, view_size=m["viewSize"], colormap=cmap, **m["render"])
    got = t.read_view()
    t.clearView()
    cleared = t.read_view()
    t.dispose()
    assert (got == want).all()
    assert not cleared.any()


def test_view_crowded_texels_are_order_exact(oracle):
    """Hundreds of fragments per view pixel: the per-fragment quantised blend must follow the stream order (runs longer
    than the lane's share are blended by the whole wave, 64 fragments at a time)."""
    n, view = 128, (48, 27)
    rng = np.random.default_rng(78)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.3, 0.3, (n, n, 2)) * [1.0, 27 / 48]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.08, .08, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur[rng.random((n, n)) < 0.2] = [-1e6, -1e6, 0, 0]
    m = dict(viewRes=list(view), viewSize=[1.0, 48 / 27], render=dict(speedLimit=0.01, flowDecay=0.005, speedAlpha=0.5,
             colorMapAlpha=0.0, baseColor=[1, 0.6, 0.2, 0.07], flowColor=[0.3, 1, 0.8, 0.05]))
    t = make(m, n)
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.timer.time = 1200.0
    t.draw()
    got, frags = t.read_view(), t.view_fragments
    t.dispose()
    want, count = oracle.view_render(cur, prev, np.zeros((view[1], view[0], 4), np.uint8), 1200.0, view_size=m["viewSize"], **m["render"])
    assert frags == count and count > 10000          # ~ 100 fragments per pixel of the crowded centre
    assert (got == want).all()


def test_draw_in_one_call_equals_the_two_passes():
    """Tendrils.draw() with renderView runs both passes over one rasterisation, one sort and one gather (th_draw); the
    flow texture and the view buffer must be what th_flow_deposit followed by th_view_draw leave - which reuses the flow
    pass's geometry - and what the two passes leave when each rasterises for itself (option draw_reuse = 0)."""
    import ctypes as C
    from tendrils_amd import _capi
    n, view = 128, (96, 54)
    rng = np.random.default_rng(21)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.6, 0.6, (n, n, 2)) * [1.0, 54 / 96]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.05, .05, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    m = dict(viewRes=list(view), viewSize=[1.0, 96 / 54], render=dict(speedLimit=0.01, flowDecay=0.005, speedAlpha=0.5,
             colorMapAlpha=0.0, baseColor=[1, 0.7, 0.3, 0.2], flowColor=[0.2, 1, 0.9, 0.1]))
    out = []
    for how in ("one call", "two passes", "two passes, no reuse"):
        t = make(m, n)
        t.particles.upload_texels(cur, 0)
        t.particles.upload_texels(prev, 1)
        t.timer.time = 900.0
        if how == "one call":
            t.draw()
        else:
            if how.endswith("no reuse"):
                t.particles.option("draw_reuse", 0)
            t.renderView = False
            t.draw()                                            # th_flow_deposit
            u, k = t.render_uniforms(), C.c_uint64(0)
            _capi.call("th_view_draw", t.particles._ctx, C.byref(u), C.byref(k))
            assert k.value == t.fragments
        out.append((t.flow.read(), t.read_view(), t.fragments))
        t.dispose()
    assert out[0][2] == out[1][2] == out[2][2] > 10000
    for other in out[1:]:
        assert bits_equal(out[0][0], other[0]).all() and (out[0][1] == other[1]).all() and out[0][1].any()


def test_draw_paths_agree_at_scale():
    """4.2 M particles over a 1080p field (7 M fragments, long runs in the wake of a few frames): draw() in one call, the two
    passes one after the other (the view pass reusing the flow pass's geometry) and the sharded form's emit + merge at one
    owner leave the same flow texture, bit for bit; the first two also the same view buffer."""
    import ctypes as C
    import tendrils_amd as ta
    from tendrils_amd import _capi, sharding
    from tendrils_amd.tendrils import View
    n, view = 2048, (1920, 1080)
    rng = np.random.default_rng(5)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2)) * [1.0, 1080 / 1920]
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    results = []
    for how in ("one call", "two passes", "emit + merge"):
        t = ta.Tendrils(View(*view))
        t.resize()
        t.setup(n)
        t.particles.upload_texels(st)
        t.timer.time = 1000.0
        t.renderView = how == "one call"
        for _ in range(3):                                      # a few frames: the wake forms, runs grow
            t.timer.tick(); t.step()
            if how == "emit + merge":
                keys, colors = sharding.emit_fragments(t)
                sharding.merge_fragments(t, keys, colors)
                continue
            t.draw()
            if how == "two passes":
                u, k = t.render_uniforms(), C.c_uint64(0)
                _capi.call("th_view_draw", t.particles._ctx, C.byref(u), C.byref(k))
                assert k.value == t.fragments
        results.append((t.flow.read(), t.read_view() if how != "emit + merge" else None, t.particles.read(0)))
        t.dispose()
    a, b, c = results
    assert bits_equal(a[2], b[2]).all() and bits_equal(a[2], c[2]).all()          # the same particles after three closed-loop frames
    assert bits_equal(a[0], b[0]).all() and bits_equal(a[0], c[0]).all()
    assert (a[1] == b[1]).all() and a[1].any()


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_node_host_view_draw(oracle):
    path = [p for p in golden("view") if "colours" in p][0]
    m, cur, prev, ref = view_fixture(path)
    script = """
    const T = require('./tendrils_amd/js');
    const cfg = JSON.parse(process.argv[1]);
    const f32 = (b) => new Float32Array(new Uint8Array(Buffer.from(b, 'base64')).buffer);
    const t = new T.Tendrils({drawingBufferWidth: cfg.view[0], drawingBufferHeight: cfg.view[1]}, {});
    t.resize(); t.setup(cfg.n);
    t.viewSize[0] = cfg.viewSize[0]; t.viewSize[1] = cfg.viewSize[1];
    Object.assign(t.state, cfg.render);
    t.particles.uploadTexels(f32(cfg.cur), 0); t.particles.uploadTexels(f32(cfg.prev), 1);
    t.timer.time = cfg.time;
    t.draw();
    const out = {view: Buffer.from(t.readView().buffer).toString('base64'), frags: t.viewFragments,
                 lines: t.exportLines(true).length / 12};
    t.dispose();
    console.log(JSON.stringify(out));
    """
    cfg = dict(n=cur.shape[0], view=m["viewRes"], viewSize=m["viewSize"], time=m["time"],
               render={k: v for k, v in m["render"].items() if k != "lineWidth"},
               cur=base64.b64encode(np.ascontiguousarray(cur, np.float32).tobytes()).decode(),
               prev=base64.b64encode(np.ascontiguousarray(prev, np.float32).tobytes()).decode())
    r = subprocess.run([shutil.which("node"), "-e", script, json.dumps(cfg)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout)
    fh, fw = ref.shape[:2]
    got = np.frombuffer(base64.b64decode(res["view"]), np.uint8).reshape(fh, fw, 4)
    want, n = oracle.view_render(cur, prev, np.zeros((fh, fw, 4), np.uint8), m["time"], view_size=m["viewSize"], **m["render"])
    assert res["frags"] == n
    # Math.sin on fp32 operands = the Python host's math.sin: byte-identical images
    assert (got == want).all()
    check_against_reference(got, ref)
