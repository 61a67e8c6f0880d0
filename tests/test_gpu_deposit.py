"""Flow deposit on the GPU (th_flow_deposit behind Tendrils.draw()): bit for bit against the CPU restatement
(same arithmetic, same stream-order blending - the result must not depend on thread scheduling), and against the
captures of the reference's own draw() with the tolerance of tests/test_deposit_oracle.py."""
import numpy as np
import pytest

from helpers import bits_equal, golden, load
from test_deposit_oracle import deposit_close, deposit_inputs

pytestmark = pytest.mark.gpu


def make(n, view_res, view_size=None, speed_limit=None):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(*view_res))
    t.resize()
    t.setup(n)
    if view_size is not None:
        t.viewSize[:] = view_size
    if speed_limit is not None:
        t.state["speedLimit"] = speed_limit
    return t


def gpu_deposit(cur, prev, base, time, view_res, view_size, speed_limit):
    t = make(cur.shape[0], view_res, view_size, speed_limit)
    # buffers[0] = current, buffers[1] = previous (src/index.js:286, src/particles.js:153)
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.flow.set_pixels(base)
    t.timer.time = time
    t.draw()
    got, frags = t.flow.read(), t.fragments
    t.dispose()
    return got, frags


@pytest.mark.parametrize("path", golden("deposit"), ids=lambda p: p.split("/")[-1][:-4])
def test_deposit_bit_exact_to_oracle_reference_coverage_exact_values_toleranced(oracle, path):
    fx = load(path)
    m, base, ref = deposit_inputs(fx)
    got, frags = gpu_deposit(fx["current"], fx["previous"], base, m["time"], m["viewRes"], m["viewSize"], m["speedLimit"])
    want, n, cov = oracle.flow_deposit(fx["current"], fx["previous"], base, m["time"], view_size=m["viewSize"],
                                       speedLimit=m["speedLimit"], coverage=True)
    assert frags == n
    assert bits_equal(got, want).all()
    touched = np.zeros(cov.size, bool)
    touched[fx["idx"]] = True
    assert ((got != base).any(-1).ravel() == touched).all()          # the reference's coverage, texel for texel
    assert deposit_close(got, ref, m["time"]).all()


def test_deposit_crowded_texels_are_order_exact(oracle):
    """Many lines through few texels (up to hundreds of fragments per texel): the blend must follow the stream
    order whatever order the fragments were appended in."""
    n, view = 128, (48, 27)
    rng = np.random.default_rng(77)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.3, 0.3, (n, n, 2)) * [1.0, 27 / 48]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.08, .08, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    k = rng.random((n, n)) < 0.2
    cur[k] = [-1e6, -1e6, 0, 0]
    base = np.zeros((27, 48, 4), np.float32)
    outs = [gpu_deposit(cur, prev, base, 2500.0, view, None, None) for _ in range(2)]
    want, frags, cov = oracle.flow_deposit(cur, prev, base, 2500.0, view_size=(1.0, 48 / 27), coverage=True)
    assert cov.max() > 100 and outs[0][1] == frags
    assert bits_equal(outs[0][0], want).all() and bits_equal(outs[1][0], want).all()


def test_step_draw_loop_matches_oracle(oracle):
    """The closed loop of the reference's frame: step() then draw(), the deposited wake steering the next step."""
    import tendrils_amd as ta
    n, view = 64, (96, 54)
    rng = np.random.default_rng(5)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-0.9, 0.9, (n, n, 2)) * [1.0, 0.5]
    st[..., 2:] = rng.uniform(-.008, .008, (n, n, 2))
    t = make(n, view)
    t.particles.upload_texels(st)
    t.timer.time = 1000.0
    cur, prev = st.copy(), st.copy()
    flow = np.zeros((54, 96, 4), np.float32)
    for _ in range(5):
        t.timer.tick()
        t.step()
        t.draw()
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        prev, cur = cur, oracle.logic_step(u, cur, flow)
        flow, _ = oracle.flow_deposit(cur, prev, flow, t.timer.time, view_size=t.viewSize, speedLimit=t.state["speedLimit"])
        assert bits_equal(t.particles.read(0), cur).all()
        assert bits_equal(t.flow.read(), flow).all()
    assert (flow[..., 3] != 0).sum() > 500
    t.dispose()


def test_export_lines_matches_oracle(oracle):
    """Trail export: the line list of draw() (two live vertices, non-zero length), stream order, bit for bit."""
    import tendrils_amd as ta
    n, view = 100, (96, 54)
    rng = np.random.default_rng(6)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-1.2, 1.2, (n, n, 2))
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.05, .05, (n, n, 2)).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    cur[5, :7] = prev[5, :7]                                   # some particles did not move
    t = make(n, view)
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.timer.time = 321.0
    got = t.export_lines()
    t.dispose()
    want = oracle.export_lines(cur, prev, 321.0, view_size=(1.0, 96 / 54))
    assert got.shape == want.shape and 3000 < len(got) < n * n
    assert bits_equal(got, want).all()


def test_export_lines_of_a_packed_ring_are_those_of_what_it_decodes_to(oracle):
    """TH_STATE_F16: the trail export reads the ring in place (dep_state) - the lines of the decoded texels, bit for bit, with and
    without the view pass's colours; n = 100: rows whose vertices are other particles."""
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    n, view = 100, (96, 54)
    rng = np.random.default_rng(16)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-1.2, 1.2, (n, n, 2))
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.05, .05, (n, n, 2)).astype(np.float32)
    cur[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    opts = ta.defaults()
    opts.update(stateFormat=ta._capi.TH_STATE_F16)
    t = ta.Tendrils(View(*view), opts)
    t.resize()
    t.setup(n)
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.timer.time = 321.0
    got = t.export_lines()
    cur16, prev16 = t.particles.read(0), t.particles.read(1)        # what the ring decodes to
    t.dispose()
    assert not bits_equal(cur16, cur).all()
    want = oracle.export_lines(cur16, prev16, 321.0, view_size=(1.0, 96 / 54))
    assert got.shape == want.shape and 3000 < len(got) < n * n
    assert bits_equal(got, want).all()


def test_closed_loop_against_reference_frames():
    """The reference's own K = 6 frames of step() + draw() against the GPU frame loop (tolerances of
    tests/test_deposit_oracle.py:loop_close; the first frame is bit-exact)."""
    import os
    from helpers import GOLDEN
    from test_deposit_oracle import loop_close
    fx = load(os.path.join(GOLDEN, "loop_frames_64.npz"))
    m = fx["meta"]
    t = make(m["N"], m["viewRes"], m["viewSize"])
    t.particles.upload_texels(fx["state"])
    t.timer.time = m["times"][0] - m["dts"][0]
    states = []
    for k in range(m["frames"]):
        t.timer.tick()
        assert t.timer.time == m["times"][k]
        t.step()
        t.draw()
        states.append(t.particles.read(0))
    flow = t.flow.read()
    t.dispose()
    loop_close(states, flow, fx)
    assert bits_equal(states[0], fx["out"][0]).all()
