"""GPU parity of the optical-flow producer (th_optical_flow through the OpticalFlow mirror):
bit-for-bit against the vectors captured from the reference shader, and against the oracle."""
import numpy as np
import pytest

from helpers import bits_equal, golden, load, of_expected, of_inputs

pytestmark = pytest.mark.gpu


def run_of(meta, f0, f1, dst):
    import tendrils_amd as ta
    from tendrils_amd.optical_flow import OpticalFlow
    from tendrils_amd.tendrils import View
    ow, oh = meta["out"]
    t = ta.Tendrils(View(ow, oh))
    t.resize()                      # flow.shape = viewRes = out size
    t.setup(8)
    t.flow.set_pixels(dst)
    of = OpticalFlow(t)
    of.resize(meta["frame"])
    of.set_pixels(f0)               # older frame ...
    of.step()                       # ... becomes `last`
    of.set_pixels(f1)               # newest frame = `view`
    of.update(meta["uniforms"])
    of.render()
    out = t.flow.read()
    t.dispose()
    return out


@pytest.mark.parametrize("path", golden("of"), ids=lambda p: p.split("/")[-1][:-4])
def test_optical_flow_matches_reference_bits(oracle, path):
    fx = load(path)
    m = fx["meta"]
    f0, f1, dst = of_inputs(m)
    full = run_of(m, f0, f1, dst)
    ok = bits_equal(of_expected(fx, full), fx["out"])
    assert ok.all(), "%s: %d components differ from the reference capture" % (fx["name"], (~ok).sum())
    un = m["uniforms"]
    u = oracle.optical_flow_uniforms(un["time"], view_size=un["viewSize"], scaleUV=un["scaleUV"],
                                     offset=un["offset"], lambda_=un["lambda"], speed=un["speed"],
                                     speedLimit=un["speedLimit"])
    want = oracle.optical_flow(u, f1, f0, dst, blend=True)
    assert bits_equal(full, want).all()          # whole texture, incl. rows the fixture does not store


def test_optical_flow_then_integrate(oracle):
    """The per-frame sequence of the demo loop (src/demo.main.js:1082,1107-1159): step, then the
    optical-flow pass blended into flow, twice; state and flow must equal the oracle's bit for bit."""
    import tendrils_amd as ta
    from helpers import synth_frame
    from tendrils_amd.optical_flow import OpticalFlow
    from tendrils_amd.tendrils import View
    n, (w, h) = 256, (240, 135)
    rng = np.random.default_rng(5)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    frames = [synth_frame(w, h, 9, shift8=(6 * k, 3 * k)) for k in range(3)]
    t = ta.Tendrils(View(w, h))
    t.resize()
    t.setup(n)
    t.particles.upload_texels(st)
    of = OpticalFlow(t, uniforms=dict(speed=0.08, offset=0.1, scaleUV=[-1, -1]))
    of.resize([w, h])
    of.set_pixels(frames[0])
    t.timer.time = 2000.0
    cur, flow = st, np.zeros((h, w, 4), np.float32)
    for k in (1, 2):
        t.timer.tick()
        t.step()
        of.step()
        of.set_pixels(frames[k])
        of.update(dict(speedLimit=t.state["speedLimit"], time=t.timer.time, viewSize=t.viewSize))
        of.render()
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{a: b for a, b in t.state.items() if isinstance(b, (int, float))})
        cur = oracle.logic_step(u, cur, flow)
        ou = oracle.optical_flow_uniforms(t.timer.time, view_size=t.viewSize, scaleUV=[-1, -1], offset=0.1,
                                          lambda_=0.001, speed=0.08, speedLimit=t.state["speedLimit"])
        flow = oracle.optical_flow(ou, frames[k], frames[k - 1], flow, blend=True)
        assert bits_equal(t.particles.read(0), cur).all()
        assert bits_equal(t.flow.read(), flow).all()
    t.dispose()
