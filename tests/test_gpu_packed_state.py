"""Packed state (TH_STATE_F16, config C5): 8 B per particle.  The arithmetic is the same exact fp32
integrator applied to the DECODED texel; only the storage is quantised.  So for a state that is
already representable, K steps on the GPU must equal K x [decode -> oracle step -> encode], bit for bit
(the encoding is defined by this build; tests/helpers.py mirrors it in numpy)."""
import numpy as np
import pytest

from helpers import bits_equal, pack_state, unpack_state
from test_gpu_logic_parity import seeded_case

pytestmark = pytest.mark.gpu


def packed_tendrils(n, view, overrides=None):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    opts["stateFormat"] = ta.TH_STATE_F16
    t = ta.Tendrils(View(*view), opts)
    t.resize()
    t.setup(n)
    t.state.update(overrides or {})
    return t


def test_numpy_mirror_roundtrip():
    rng = np.random.default_rng(0)
    st = np.empty((64, 64, 4), np.float32)
    st[..., :2] = rng.uniform(-2.5, 2.5, (64, 64, 2))
    st[..., 2:] = rng.uniform(-.02, .02, (64, 64, 2))
    st[0, :8] = [-1e6, -1e6, 0, 0]
    st[1, :4, 0] = np.nan
    q = unpack_state(pack_state(st))
    assert bits_equal(unpack_state(pack_state(q)), q).all()                  # idempotent on representable states
    live = ~((st[..., 0] == -1e6) & (st[..., 1] == -1e6)) & ~np.isnan(st[..., 0])
    inside = live & (np.abs(st[..., :2]) < 1.99).all(-1)
    assert np.abs(q[inside][:, :2] - st[inside][:, :2]).max() <= 2.0 ** -15  # half a position quantum
    assert np.allclose(q[live][:, 2:], st[live][:, 2:], rtol=2.0 ** -11, atol=1e-7)
    assert (q[0, :8, 0] == -1e6).all() and np.isnan(q[1, :4, 0]).all()


@pytest.mark.parametrize("n,overrides", [(256, {}), (256, {"noiseWeight": 0}), (200, {"target": 0.0005})])
def test_packed_steps_equal_decode_oracle_encode(oracle, n, overrides):
    st, fl = seeded_case(n, 4242 + n)
    st[0, :3, 0] = np.nan                                   # NaN marker survives
    st = unpack_state(pack_state(st))                       # a representable start state
    rng = np.random.default_rng(n)
    tg = np.zeros((n, n, 4), np.float32)
    tg[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    t = packed_tendrils(n, (96, 54), overrides)
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.targets.set_pixels(tg)
    assert bits_equal(t.particles.read(0), st).all()        # upload -> pack -> unpack -> download is lossless here
    t.timer.time = 4000.0
    cur = st
    for _ in range(4):
        t.timer.tick()
        t.step()
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        cur = unpack_state(pack_state(oracle.logic_step(u, cur, fl, tg)))
        assert bits_equal(t.particles.read(0), cur).all()
    t.dispose()


def test_packed_statistics_taken_by_the_fused_launch():
    """As tests/test_gpu_logic_parity.py:test_statistics_taken_by_the_fused_launch..., on a packed ring: the statistics are
    those of what the stored texels decode to."""
    from tendrils_amd import _capi
    import tendrils_amd as ta
    n = 256
    st, fl = seeded_case(n, 99)
    t = packed_tendrils(n, (96, 54), {"forceWeight": 0.2, "noiseWeight": 0.02})
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.timer.time = 4000.0
    limit = t.state["speedLimit"]
    for steps in (6, 7):
        t.step_n(steps)
        fused = t.particles.stats(limit)
        _capi.call("th_set_mode", t.particles._ctx, ta.TH_MODE_EXACT)      # drops the launch's partials: the pass over the state
        again = t.particles.stats(limit)
        got = t.particles.read(0)
        speed = np.sqrt(got[..., 2] ** 2 + got[..., 3] ** 2)
        live = (got[..., 0] != np.float32(-1e6)) | (got[..., 1] != np.float32(-1e6))
        assert {k: v for k, v in fused.items() if k != "sum_speed"} == {k: v for k, v in again.items() if k != "sum_speed"}
        assert abs(fused["sum_speed"] - again["sum_speed"]) <= 1e-12 * again["sum_speed"]
        assert fused["live"] == int(live.sum()) and fused["max_speed"] == float(speed[live].max()) and 0 < fused["capped"] < fused["live"]
    t.dispose()


def test_packed_graph_replay_and_spawners(oracle):
    from tendrils_amd.spawn import spawnBall
    n = 128
    st, fl = seeded_case(n, 99)
    st = unpack_state(pack_state(st))
    outs = []
    for graph in (False, True):
        t = packed_tendrils(n, (96, 54))
        t.particles.upload_texels(st)
        t.flow.set_pixels(fl)
        t.timer.time = 4000.0
        if graph:
            t.step_n(6)
        else:
            for _ in range(6):
                t.timer.tick()
                t.step()
        outs.append(t.particles.read(0))
        stats = t.particles.stats(t.state["speedLimit"])
        assert stats["particles"] == n * n
        # a spawn pass renders f32 and is re-packed
        spawnBall(None, dict(uniforms=dict(radius=0.5, speed=0.004))).spawn(t)
        ball = t.particles.read(0)
        assert bits_equal(ball, unpack_state(pack_state(oracle.spawn_ball(n, n, radius=0.5, speed=0.004)))).all()
        t.dispose()
    assert bits_equal(outs[0], outs[1]).all()


@pytest.mark.parametrize("row0", [0, 8188, 16376])
def test_c5_16384_wide_band_equals_oracle(oracle, row0):
    """Config C5 geometry: a row band of the 16384^2 packed-state texture (index i = (fx + fy*W)/(W*H) with
    W*H = 2^28, well past fp32's exact integers).  The reference's GL stops at 8192^2, so the pin here is the
    restatement (itself pinned to the reference up to 8192^2) evaluated with the global coordinates."""
    import tendrils_amd as ta
    from helpers import hashed_state
    from tendrils_amd.tendrils import View
    n, rows = 16384, 8
    st = unpack_state(pack_state(hashed_state(n, 555, 31, rows=(row0, row0 + rows))))
    fl = seeded_case(64, 5)[1]
    fl[..., 2] += 86000.0                                   # deposit times around the run's clock: live flow
    opts = ta.defaults()
    opts.update(stateFormat=ta.TH_STATE_F16, row0=row0, rows=rows, globalHeight=n)
    t = ta.Tendrils(View(96, 54), opts)
    t.resize()
    t.setup(n)
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.timer.time = 90000.0
    cur = st
    for _ in range(3):
        t.timer.tick()
        t.step()
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        cur = unpack_state(pack_state(oracle.logic_step(u, cur, fl, y0=row0)))
        assert bits_equal(t.particles.read(0), cur).all()
    t.dispose()


def test_packed_frame_loop_deposit(oracle):
    """step() + draw() on the packed ring: the deposit reads what the stored texels decode to."""
    n, view = 64, (96, 54)
    rng = np.random.default_rng(9)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-0.9, 0.9, (n, n, 2)) * [1.0, 0.5]
    st[..., 2:] = rng.uniform(-.008, .008, (n, n, 2))
    st = unpack_state(pack_state(st))
    t = packed_tendrils(n, view)
    t.particles.upload_texels(st)
    t.timer.time = 1000.0
    cur, prev, flow = st.copy(), st.copy(), np.zeros((54, 96, 4), np.float32)
    for _ in range(3):
        t.timer.tick()
        t.step()
        t.draw()
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        prev, cur = cur, unpack_state(pack_state(oracle.logic_step(u, cur, flow)))
        flow, _ = oracle.flow_deposit(cur, prev, flow, t.timer.time, view_size=t.viewSize, speedLimit=t.state["speedLimit"])
        assert bits_equal(t.particles.read(0), cur).all() and bits_equal(t.flow.read(), flow).all()
    t.dispose()


def test_c5_per_gpu_band_full_size(oracle):
    """Config C5's per-GPU share at 8 GPUs: a 2048-row band of the 16384^2 packed-state texture (33.5 M particles), 16
    steps as one fused launch (the configuration's step group).  Size-independent properties over the whole band, the
    oracle on rows from its first, a middle and its last row, and the second-newest state the ring keeps."""
    import tendrils_amd as ta
    from helpers import hashed_state
    from tendrils_amd.tendrils import View
    n, rows, row0, steps = 16384, 2048, 6 * 2048, 16
    fl = seeded_case(64, 5)[1]
    fl[..., 2] += 86000.0
    opts = ta.defaults()
    opts.update(stateFormat=ta.TH_STATE_F16, row0=row0, rows=rows, globalHeight=n)
    t = ta.Tendrils(View(96, 54), opts)
    t.resize()
    t.setup(n)
    probes = [0, 1000, rows - 2]                       # two rows each
    kept = {}
    for r0 in range(0, rows, 256):                     # generated and uploaded in slabs (host memory)
        slab = unpack_state(pack_state(hashed_state(n, 777, 29, rows=(row0 + r0, row0 + r0 + 256))))
        for pr in probes:
            if r0 <= pr < r0 + 256:
                kept[pr] = slab[pr - r0:pr - r0 + 2].copy()
        inert_rows = (slab[..., 0] == -1e6) & (slab[..., 1] == -1e6)
        kept.setdefault("inert", []).append(inert_rows)
        _capi = __import__("tendrils_amd")._capi
        _capi.call("th_upload_state", t.particles._ctx, -1, np.ascontiguousarray(slab).ctypes.data_as(_capi._fp), 0, r0, n, 256)
    t.flow.set_pixels(fl)
    t.timer.time = 90000.0
    tm = ta.Timer(0, 0)
    tm.step, tm.time = t.timer.step, t.timer.time
    t.step_n(steps)
    got0, got1 = t.particles.read(0), t.particles.read(1)
    stats = t.particles.stats(t.state["speedLimit"])
    t.dispose()
    inert = np.concatenate(kept["inert"])
    assert stats["particles"] == n * rows and stats["live"] == int((~inert).sum())
    # inert particles pass through every step untouched; live ones keep |vel| <= speedLimit (+ one fp16 rounding)
    assert (got0[inert] == np.array([-1e6, -1e6, 0, 0], np.float32)).all()
    speed = np.hypot(got0[..., 2], got0[..., 3])[~inert]
    speed = speed[np.isfinite(speed)]
    assert speed.max() <= t.state["speedLimit"] * (1 + 2.0 ** -10)
    cur = {pr: kept[pr] for pr in probes}
    prev = dict(cur)
    for _ in range(steps):
        tm.tick()
        u = oracle.logic_uniforms(n, n, tm.time, tm.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        for pr in probes:
            prev[pr] = cur[pr]
            cur[pr] = unpack_state(pack_state(oracle.logic_step(u, cur[pr], fl, y0=row0 + pr)))
    for pr in probes:
        assert bits_equal(got0[pr:pr + 2], cur[pr]).all(), "rows %d.." % pr
        assert bits_equal(got1[pr:pr + 2], prev[pr]).all(), "previous state, rows %d.." % pr
