"""GPU parity of the HIP integrator (through the C ABI) against (a) the golden vectors captured
from the reference's own shader and (b) the CPU oracle on larger seeded inputs.
EXACT mode: bit-for-bit.  FAST mode: absolute tolerance for one step and a drift bound over fused launches, both stated below."""
import numpy as np
import pytest

from helpers import bits_equal, golden, load, state_overrides

pytestmark = pytest.mark.gpu

# FAST mode (FMA contraction, rcp/rsq): |delta| per component after ONE step.
# Velocities are O(1e-2) (speedLimit 0.01); the bound is ~1e-4 relative to that.
FAST_ATOL = 2e-6


def make_tendrils(n, view_res, flow_shape, state, mode):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    opts["mode"] = mode
    t = ta.Tendrils(View(*view_res), opts)
    t.resize()
    t.setup(n)
    t.flow.shape = flow_shape
    for k, v in state.items():
        t.state[k] = v
    return t


def run_fixture(fx, mode):
    m = fx["meta"]
    n = m["N"]
    t = make_tendrils(n, m["viewRes"], m["flowShape"], state_overrides(m), mode)
    t.viewSize[:] = m["viewSize"]
    t.particles.upload_texels(fx["state"])
    t.flow.set_pixels(fx["flow"])
    if "targets" in fx:
        t.targets.set_pixels(fx["targets"])
    outs = []
    cur = fx["state"]
    for k in range(m["steps"]):
        # follow the reference trajectory exactly as the oracle test does
        if k:
            t.particles.upload_texels(cur)
        t.timer.time = m["times"][k] - m["dts"][k]
        t.timer.tick()
        assert t.timer.time == m["times"][k] and t.timer.dt == m["dts"][k]
        t.step()
        outs.append(t.particles.read(0))
        cur = fx["out"][k]
    t.dispose()
    return outs


@pytest.mark.parametrize("path", golden("logic"), ids=lambda p: p.split("/")[-1][:-4])
def test_exact_mode_matches_reference_bits(path):
    import tendrils_amd as ta
    fx = load(path)
    outs = run_fixture(fx, ta.TH_MODE_EXACT)
    for k, got in enumerate(outs):
        ok = bits_equal(got, fx["out"][k]).all(-1)
        bad = np.argwhere(~ok & fx["valid"][k])
        assert len(bad) == 0, "%s step %d: %d texels differ, first %s" % (fx["name"], k, len(bad), bad[:4])


@pytest.mark.parametrize("path", golden("logic"), ids=lambda p: p.split("/")[-1][:-4])
def test_fast_mode_within_tolerance(path):
    import tendrils_amd as ta
    fx = load(path)
    outs = run_fixture(fx, ta.TH_MODE_FAST)
    for k, got in enumerate(outs):
        ref = fx["out"][k]
        v = fx["valid"][k]
        assert (np.isnan(got) == np.isnan(ref))[v].all()
        d = np.abs(np.nan_to_num(got) - np.nan_to_num(ref))[v]
        assert d.max() <= FAST_ATOL, "%s step %d: max |delta| %.3g" % (fx["name"], k, d.max())


# FAST mode over many steps: no bit claim, a drift bound.  Rounding differences of ~1e-7 per step add up along a
# trajectory (and a particle sitting on the edge of a flow texel may take its neighbour's tap once: the few outliers);
# measured on an MI355X at 512^2 (tools/fast_drift_probe.py): p99.99 of |d pos| 8e-6 after 20 steps, 6e-5 after 100,
# the largest single particle 2.5e-3 / 4.3e-3.  Positions are O(1), one step moves a particle by <= speedLimit = 0.01.
FAST_DRIFT = {20: (5e-5, 2e-2), 100: (5e-4, 5e-2)}      # steps: (p99.99 of |d pos|, max |d pos|)


def test_fast_mode_drift_over_fused_launches():
    import tendrils_amd as ta
    n = 512
    st, fl = seeded_case(n, 99, inert=0.02, pos_range=1.0)
    outs = {}
    for mode in (ta.TH_MODE_EXACT, ta.TH_MODE_FAST):
        t = make_tendrils(n, (96, 54), (96, 54), {}, mode)
        t.particles.upload_texels(st)
        t.flow.set_pixels(fl)
        t.timer.time = 4000.0
        seq = []
        for k in (20, 80):
            t.step_n(k)
            seq.append(t.particles.read(0).copy())
        outs[mode] = seq
        t.dispose()
    for steps, a, b in zip((20, 100), outs[ta.TH_MODE_EXACT], outs[ta.TH_MODE_FAST]):
        assert (np.isnan(a) == np.isnan(b)).all(), "%d steps: NaN lanes differ" % steps
        parked = np.abs(a[..., 0]) > 1e5
        assert (parked == (np.abs(b[..., 0]) > 1e5)).all(), "%d steps: idle particles differ" % steps
        d = np.abs(np.nan_to_num(a) - np.nan_to_num(b))[~parked][:, :2]
        q, worst = FAST_DRIFT[steps]
        assert np.quantile(d, 0.9999) <= q and d.max() <= worst, "%d steps: p99.99 %.3g max %.3g" % (steps, np.quantile(d, 0.9999), d.max())


def seeded_case(n, seed, flow_shape=(96, 54), inert=0.05, pos_range=1.2):
    rng = np.random.default_rng(seed)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-pos_range, pos_range, (n, n, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    st[rng.random((n, n)) < inert] = [-1e6, -1e6, 0, 0]
    fw, fh = flow_shape
    fl = np.zeros((fh, fw, 4), np.float32)
    fl[..., :2] = rng.uniform(-.01, .01, (fh, fw, 2))
    fl[..., 2] = 4000 + rng.uniform(-150, 16, (fh, fw))
    fl[..., 3] = 1
    return st, fl


@pytest.mark.parametrize("n,overrides", [
    (512, {}),
    (512, {"noiseWeight": 0}),
    (1024, {}),
    (1024, {"target": 0.0005}),
    (200, {}),                      # non power-of-two: true divisions by dataRes
    (200, {"noiseWeight": 0, "target": 0.001}),
])
def test_exact_mode_matches_oracle_on_seeded_inputs(oracle, n, overrides):
    import tendrils_amd as ta
    st, fl = seeded_case(n, 7 + n)
    rng = np.random.default_rng(n)
    tg = np.zeros((n, n, 4), np.float32)
    tg[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    t = make_tendrils(n, (96, 54), (96, 54), overrides, ta.TH_MODE_EXACT)
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.targets.set_pixels(tg)
    t.timer.time = 4000.0
    cur = st
    for _ in range(3):
        t.timer.tick()
        t.step()
        got = t.particles.read(0)
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{**{k: v for k, v in t.state.items() if isinstance(v, (int, float))}})
        want = oracle.logic_step(u, cur, fl, tg)
        ok = bits_equal(got, want)
        assert ok.all(), "n=%d %s: %d components differ" % (n, overrides, (~ok).sum())
        cur = want
    t.dispose()


def test_out_of_domain_lanes_take_reference_path(oracle):
    """Huge / infinite / NaN positions leave the fast path's proven domain: results must still be
    the reference's (here: the oracle's), bit for bit."""
    import tendrils_amd as ta
    n = 64
    st, fl = seeded_case(n, 99)
    st[0, 0, :2] = [3e6, 0.1]
    st[0, 1, :2] = [1e30, -1e30]
    st[0, 2, :2] = [np.inf, 0.0]
    st[0, 3, :2] = [np.nan, 0.5]
    st[0, 4, 2:] = [np.nan, 0.0]
    st[0, 5, :2] = [-1e6, 0.25]          # only one component inert -> still live
    st[0, 6, :2] = [5e5, 5e5]
    for overrides in ({}, {"noiseWeight": 0}):
        t = make_tendrils(n, (96, 54), (96, 54), overrides, ta.TH_MODE_EXACT)
        t.particles.upload_texels(st)
        t.flow.set_pixels(fl)
        t.timer.time = 4000.0
        t.timer.tick()
        t.step()
        got = t.particles.read(0)
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        want = oracle.logic_step(u, st, fl)
        assert bits_equal(got, want).all()
        t.dispose()


def test_ring_semantics_and_errors():
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(32, 32))
    t.resize()
    t.setup(32)
    b0, b1 = t.particles.buffers
    st, _ = seeded_case(32, 3, inert=0)
    t.particles.upload_texels(st)
    t.timer.tick()
    t.step()
    # utils.step rotated the ring: the written buffer is the old LAST one, now in front
    assert t.particles.buffers == [b1, b0]
    assert np.array_equal(t.particles.read(b0), st)                 # previous state untouched
    assert not np.array_equal(t.particles.read(b1), st)
    # Particles.step needs buffers[1]
    t.particles.setup(1)
    with pytest.raises(ta.TendrilsHipError):
        t.step()
    t.dispose()


def test_full_size_properties():
    """C3-sized (4096^2) run: size-independent properties instead of a full CPU check.
    (1) inert texels pass through bit-identically, (2) speed never exceeds speedLimit,
    (3) newPos == pos + newVel exactly, (4) two identical runs are bit-identical,
    (5) a row band computed by a sharded context equals the same rows of the full run."""
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    n = 4096
    rng = np.random.default_rng(2024)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2)).astype(np.float32)
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2)).astype(np.float32)
    inert = rng.random((n, n)) < 0.03
    st[inert] = [-1e6, -1e6, 0, 0]
    fl = np.zeros((270, 480, 4), np.float32)
    fl[..., :2] = rng.uniform(-.02, .02, (270, 480, 2))
    fl[..., 2] = 990.0
    outs = []
    for rep in range(2):
        t = ta.Tendrils(View(480, 270))
        t.resize()
        t.setup(n)
        t.particles.upload_texels(st)
        t.flow.set_pixels(fl)
        t.timer.time = 1000.0
        t.timer.tick()
        t.step()
        outs.append(t.particles.read(0))
        vs, tm, dt = list(t.viewSize), t.timer.time, t.timer.dt
        t.dispose()
    a = outs[0]
    assert bits_equal(a, outs[1]).all()
    assert bits_equal(a[inert], st[inert]).all()
    live = ~inert
    sp = np.hypot(a[..., 2].astype(np.float64), a[..., 3].astype(np.float64))[live]
    assert sp.max() <= 0.01 * (1 + 1e-6)
    assert np.array_equal(a[live][:, :2], (st[live][:, :2] + a[live][:, 2:]).astype(np.float32))
    # sharded band (rows 1024..1536 of the global texture) on its own context
    opts = ta.defaults()
    opts.update(row0=1024, rows=512, globalHeight=n)
    tb = ta.Tendrils(View(480, 270), opts)
    tb.resize()
    tb.setup(n)
    tb.particles.upload_texels(st[1024:1536])
    tb.flow.set_pixels(fl)
    tb.timer.time = 1000.0
    tb.timer.tick()
    tb.step()
    band = tb.particles.read(0)
    tb.dispose()
    assert bits_equal(band, a[1024:1536]).all()


@pytest.mark.parametrize("n,steps", [(256, 7), (64, 4), (200, 5)])
def test_step_n_graph_equals_single_steps(n, steps):
    """th_step_n (captured hipGraph, device-resident per-step time) == the same number of th_step calls,
    bit for bit, including a second replay of the cached graph and the odd/even ring parity."""
    import tendrils_amd as ta
    st, fl = seeded_case(n, 1000 + n)
    outs = []
    for graph in (False, True):
        t = make_tendrils(n, (96, 54), (96, 54), {}, ta.TH_MODE_EXACT)
        t.particles.upload_texels(st)
        t.flow.set_pixels(fl)
        t.timer.time = 4000.0
        for _ in range(3):               # 3 batches: capture, replay (other ring parity when steps is odd), replay
            if graph:
                t.step_n(steps)
            else:
                for _ in range(steps):
                    t.timer.tick()
                    t.step()
        outs.append((t.particles.read(0), t.particles.read(1), t.timer.time))
        t.dispose()
    assert outs[0][2] == outs[1][2]
    assert bits_equal(outs[0][0], outs[1][0]).all()
    assert bits_equal(outs[0][1], outs[1][1]).all()


@pytest.mark.parametrize("n,steps", [(256, 7), (200, 40), (1024, 6)])
def test_statistics_taken_by_the_fused_launch_equal_the_statistics_pass(n, steps):
    """th_step_n's last fused launch takes the statistics of the state it leaves in buffers[0] while that state is in
    registers; th_stats then only folds the per-workgroup partials.  They must be the statistics pass's: counts and the
    maximum exactly, the sum of the speeds up to the order of its additions - with the launch's own speedLimit (the cached
    partials), with another limit (the pass over the state), after one more host call (the cache is dropped: the pass
    again), and again after the next fused launch."""
    import ctypes as C
    import tendrils_amd as ta
    from tendrils_amd import _capi
    st, fl = seeded_case(n, 77 + n)
    t = make_tendrils(n, (96, 54), (96, 54), {"forceWeight": 0.2, "noiseWeight": 0.02}, ta.TH_MODE_EXACT)    # forces that drive many particles into the speed limit: `capped`
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.timer.time = 4000.0
    limit = t.state["speedLimit"]
    for _ in range(2):
        t.step_n(steps)
        fused = t.particles.stats(limit)                   # (the fused launch's partials, when the launch was a fused one)
        other = t.particles.stats(limit * 0.5)             # another limit: the pass over the state
        _capi.call("th_set_mode", t.particles._ctx, ta.TH_MODE_EXACT)      # any call that may write state drops the partials
        again = t.particles.stats(limit)
        got = t.particles.read(0)
        speed = np.sqrt(got[..., 2].astype(np.float32) ** 2 + got[..., 3].astype(np.float32) ** 2)
        live = (got[..., 0] != np.float32(-1e6)) | (got[..., 1] != np.float32(-1e6))
        assert fused["live"] == again["live"] == int(live.sum()) and fused["nan"] == again["nan"] == other["nan"]
        assert 0 < fused["capped"] == again["capped"] < fused["live"] and other["capped"] > fused["capped"]
        assert fused["max_speed"] == again["max_speed"] == float(speed[live].max())
        assert abs(fused["sum_speed"] - again["sum_speed"]) <= 1e-12 * again["sum_speed"]
        assert fused["particles"] == n * n
    t.dispose()


def test_c3_size_matches_reference_row_bands():
    """4096^2 on the GPU against the row bands captured from the reference's own 4096^2 run."""
    import os
    import tendrils_amd as ta
    from helpers import GOLDEN, hashed_state
    from tendrils_amd.tendrils import View
    fx = load(os.path.join(GOLDEN, "logic_4096_bands.npz"))
    m = fx["meta"]
    n = m["N"]
    t = ta.Tendrils(View(*m["viewRes"]))
    t.resize()
    t.setup(n)
    for k, v in state_overrides(m).items():
        t.state[k] = v
    t.particles.upload_texels(hashed_state(n, m["seed"], m["inertMod"]))
    t.flow.set_pixels(fx["flow"])
    t.timer.time = m["times"][0] - m["dts"][0]
    t.timer.tick()
    assert t.timer.time == m["times"][0]
    t.step()
    got = t.particles.read(0)
    t.dispose()
    row = 0
    for (a, b) in m["bands"]:
        ok = bits_equal(got[a:b], fx["out"][row:row + (b - a)]).all(-1)
        assert (ok | ~fx["valid"][row:row + (b - a)]).all(), "rows %d..%d" % (a, b)
        row += b - a


def test_c2_curl_flow_trajectory_matches_reference_row_bands():
    """Config C2: 1024^2 particles, curl-noise flow 1024x1024, K = 4 steps - against the bands captured from the
    reference's own run, step by step (single th_step calls) and again through the fused th_step_n."""
    import os
    import tendrils_amd as ta
    from helpers import GOLDEN, band_fixture_flow, hashed_state
    from tendrils_amd.tendrils import View
    fx = load(os.path.join(GOLDEN, "logic_c2_1024_bands.npz"))
    m = fx["meta"]
    n, steps = m["N"], m["steps"]
    flow = band_fixture_flow(fx)
    st = hashed_state(n, m["seed"], m["inertMod"])

    def check(got, k):
        row = 0
        for (a, b) in m["bands"]:
            ok = bits_equal(got[a:b], fx["out"][k][row:row + (b - a)]).all(-1)
            assert (ok | ~fx["valid"][row:row + (b - a)]).all(), "step %d rows %d..%d" % (k, a, b)
            row += b - a

    for fused in (False, True):
        t = ta.Tendrils(View(*m["viewRes"]))
        t.resize()
        t.setup(n)
        for k, v in state_overrides(m).items():
            t.state[k] = v
        t.particles.upload_texels(st)
        t.flow.set_pixels(flow)
        t.timer.time = m["times"][0] - m["dts"][0]
        if fused:
            t.step_n(steps)
            assert t.timer.time == m["times"][-1]
            check(t.particles.read(0), steps - 1)
            check(t.particles.read(1), steps - 2)
        else:
            for k in range(steps):
                t.timer.tick()
                assert t.timer.time == m["times"][k]
                t.step()
                check(t.particles.read(0), k)
        t.dispose()


def test_c4_sharded_8192_matches_reference_row_bands():
    """Config C4: 8192^2 particles as 8 row-band shards (one context per shard, run one after the other on this
    GPU) - every shard reproduces the rows the reference's own unsharded 8192^2 run produced around its edges."""
    import os
    import tendrils_amd as ta
    from helpers import GOLDEN, hashed_state
    from tendrils_amd.sharding import shard_rows
    from tendrils_amd.tendrils import View
    path = os.path.join(GOLDEN, "logic_c4_8192_bands.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    fx = load(path)
    m = fx["meta"]
    n, world = m["N"], 8
    offs, row = {}, 0
    for (a, b) in m["bands"]:
        offs[(a, b)] = row
        row += b - a
    checked = 0
    for rank in range(world):
        row0, rows = shard_rows(n, world, rank)
        opts = ta.defaults()
        opts.update(row0=row0, rows=rows, globalHeight=n)
        t = ta.Tendrils(View(*m["viewRes"]), opts)
        t.resize()
        t.setup(n)
        for k, v in state_overrides(m).items():
            t.state[k] = v
        t.particles.upload_texels(hashed_state(n, m["seed"], m["inertMod"], rows=(row0, row0 + rows)))
        t.flow.set_pixels(fx["flow"])
        t.timer.time = m["times"][0] - m["dts"][0]
        t.timer.tick()
        t.step()
        got = t.particles.read(0)
        t.dispose()
        for (a, b), o in offs.items():
            lo, hi = max(a, row0), min(b, row0 + rows)
            if lo >= hi:
                continue
            ok = bits_equal(got[lo - row0:hi - row0], fx["out"][0][o + lo - a:o + hi - a]).all(-1)
            assert (ok | ~fx["valid"][o + lo - a:o + hi - a]).all(), "rank %d rows %d..%d" % (rank, lo, hi)
            checked += hi - lo
    assert checked == row
