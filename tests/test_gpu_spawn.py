"""GPU parity of the respawn kernels through the spawner mirrors: bit-for-bit against the oracle
(same pinned sin/cos), statistically against the reference captures."""
import numpy as np
import pytest

from helpers import bits_equal, golden, load
from test_spawn_oracle import oracle_spawn

pytestmark = pytest.mark.gpu


def make(n, view=(96, 54)):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(*view))
    t.resize()
    t.setup(n)
    return t


@pytest.mark.parametrize("path", [p for p in golden("spawn") if "ball" in p], ids=lambda p: p.split("/")[-1][:-4])
def test_spawn_ball_bit_exact_to_oracle_statistical_to_reference(oracle, path):
    from tendrils_amd.spawn import spawnBall
    fx = load(path)
    m = fx["meta"]
    t = make(m["N"])
    time0 = t.timer.time
    spawnBall(None, dict(uniforms=m["uniforms"])).spawn(t)
    assert t.timer.time == time0 + t.timer.step            # spawnShader ticks the timer (src/index.js:433)
    got = t.particles.read(0)
    t.dispose()
    assert bits_equal(got, oracle_spawn(oracle, fx)).all()
    ref = fx["out"]
    assert abs(got.mean() - ref.mean()) < 0.02 * ref.std() + 1e-9


@pytest.mark.parametrize("path", [p for p in golden("spawn") if "sample" in p and "image" not in p and "geometry" not in p], ids=lambda p: p.split("/")[-1][:-4])
def test_spawn_sample_bit_exact_to_oracle_statistical_to_reference(oracle, path):
    from tendrils_amd.spawn import PixelSpawner, data_sample_frag, flow_sample_frag
    fx = load(path)
    m = fx["meta"]
    un = m["uniforms"]
    t = make(m["N"])
    t.particles.upload_texels(fx["state"])
    flow_src = m["apply"] == 0
    if flow_src:
        t.flow.set_pixels(fx["data"])
        t.state["flowDecay"] = un["flowDecay"]
    sp = PixelSpawner(None, dict(shader=flow_sample_frag() if flow_src else data_sample_frag(),
                                 buffer=t.flow if flow_src else t.particles.buffers[0],
                                 spawnSize=un["spawnSize"], speed=un["speed"], bias=un["bias"]))
    t.timer.time = un["time"] - t.timer.step               # spawnShader ticks first
    sp.spawn(t)
    assert abs(t.timer.time - un["time"]) < 1e-9
    assert np.allclose(sp.jitter, un["jitter"])
    got = t.particles.read(0)
    respawned = t.particles.stats(0.01)["respawned"]
    t.dispose()
    want = oracle_spawn(oracle, fx)
    assert bits_equal(got, want).all()
    # the respawn counter (th_counters.respawned) = particles that took a candidate
    assert respawned == int((~bits_equal(want, fx["state"]).all(-1)).sum()) > 0


def test_spawn_init_and_targets(oracle):
    from tendrils_amd.spawn import spawnBall, spawner
    t = make(64)
    spawnBall(None, dict(uniforms=dict(radius=0.5, speed=0.01))).spawn(t)
    b0 = t.particles.buffers[0]
    ball = t.particles.read(b0)
    # render the ball into the targets texture: no ring rotation (src/particles.js:124-126)
    order = list(t.particles.buffers)
    spawnBall(None, dict(uniforms=dict(radius=0.25, speed=0.0))).spawn(t, t.targets)
    assert t.particles.buffers == order
    assert t.particles.stats(0.01)["respawned"] == 64 * 64          # the ball pass; the pass into `targets` is not a respawn
    assert bits_equal(t.targets.read(), oracle.spawn_ball(64, 64, radius=0.25, speed=0.0)).all()
    assert bits_equal(t.particles.read(b0), ball).all()
    spawner().spawn(t)                                     # default program: all inert
    assert bits_equal(t.particles.read(0), oracle.spawn_init((64, 64))).all()
    t.dispose()


@pytest.mark.parametrize("path", [p for p in golden("spawn") if "image" in p], ids=lambda p: p.split("/")[-1][:-4])
def test_image_spawners(oracle, path):
    """The image spawners of src/demo.main.js:455-515: PixelSpawner with its own buffer (setPixels), index.frag
    (direct) and best-sample.frag (6 samples, colour apply over the vignette pass)."""
    from tendrils_amd.spawn import PixelSpawner, best_sample_frag, pixels_frag
    fx = load(path)
    m, un = fx["meta"], fx["meta"]["uniforms"]
    t = make(m["N"])
    t.particles.upload_texels(fx["state"])
    direct = m["kind"] == "spawn_direct"
    sp = PixelSpawner(None, dict(shader=pixels_frag() if direct else best_sample_frag(),
                                 spawnSize=un["spawnSize"], speed=un["speed"], bias=un["bias"],
                                 jitterRad=2 if any(un["jitter"]) else 0))
    sp.spawnMatrix = list(un["spawnMatrix"])
    sp.setPixels(fx["data"])
    t.timer.time = un["time"] - t.timer.step
    sp.spawn(t)
    assert np.allclose(sp.jitter, un["jitter"])
    got = t.particles.read(0)
    stats = t.particles.stats(0.01)
    t.dispose()
    assert bits_equal(got, oracle_spawn(oracle, fx)).all()
    if direct:
        assert stats["respawned"] == m["N"] * m["N"]
        assert np.abs(got[..., 2:] - fx["out"][..., 2:]).max() <= 2e-7           # against the reference capture


def test_geometry_spawner(oracle):
    """GeometrySpawner (src/spawn/geometry/index.js): shuffle(), the triangle draw into the spawner's buffer
    (bit-identical to the reference capture) and the bright-sample pass over it (bit-identical to the oracle)."""
    import os
    from helpers import GOLDEN
    from tendrils_amd.spawn import GeometrySpawner
    fx = load(os.path.join(GOLDEN, "geometry_triangles_96x54.npz"))
    n = 64
    t = make(n, view=(480, 270))                            # buffer = 0.2 * viewRes = 96 x 54
    rng = np.random.default_rng(2)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.004, .004, (n, n, 2))
    st[rng.random((n, n)) < 0.3] = [-1e6, -1e6, 0, 0]
    t.particles.upload_texels(st)
    sp = GeometrySpawner(None, dict(speed=0.005, bias=1e2 / 5e-3, positions=[0.0] * 42))     # src/demo.main.js:446-447
    seq = iter(np.random.default_rng(7).random(1000))
    sp.random = lambda: float(next(seq))
    sp.shuffle()
    assert all(sp.positions[k] == 0.0 for k in range(0, 42, 6)) and any(sp.positions)         # centre vertices stay put
    sp.positions = [float(v) for v in fx["positions"]]
    t.timer.time = 2000.0
    sp.spawn(t)
    img = sp.buffer.read()
    assert bits_equal(img, fx["out"]).all()                  # the reference's own raster of these triangles
    u = oracle.spawn_sample_uniforms(n, n, t.timer.time, 6, 3, spawnSize=sp.spawnSize, jitter=sp.jitter,
                                     speed=sp.speed, bias=sp.bias, spawnMatrix=sp.spawnMatrix)
    want = oracle.spawn_sample(u, st, img)
    got = t.particles.read(0)
    t.dispose()
    assert bits_equal(got, want).all()
    assert (~bits_equal(got, st).all(-1)).sum() > 100        # a fair number of particles took a candidate
