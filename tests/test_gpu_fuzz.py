"""Differential fuzz: random sequences of the path's operations (step, step_n, draw, ball / flow-sample respawns,
uniform changes) on random shapes, mirrored operation by operation on the CPU restatement.  Everything must stay
bit-identical - this is what shakes out interactions (ring parity after odd step_n, slot layout transitions around
draw(), TARGET / NOISE specialisations, non power-of-two textures, inert and NaN particles)."""
import numpy as np
import pytest

from helpers import bits_equal

pytestmark = pytest.mark.gpu


def oracle_uniforms(oracle, t, n):
    return oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                 **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})


@pytest.mark.parametrize("seed", range(6))
def test_random_operation_sequences(oracle, seed):
    import tendrils_amd as ta
    from tendrils_amd.spawn import PixelSpawner, flow_sample_frag, spawnBall
    from tendrils_amd.tendrils import View
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([32, 48, 64, 100, 128]))
    view = [(96, 54), (64, 64), (80, 60), (120, 50)][int(rng.integers(0, 4))]
    t = ta.Tendrils(View(*view))
    t.resize()
    t.setup(n)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1.1, 1.1, (n, n, 2)) * [1.0, view[1] / view[0]]
    st[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    st[rng.random((n, n)) < 0.15] = [-1e6, -1e6, 0, 0]
    if seed % 2:
        st[1, 1, 0] = np.nan
    tg = np.zeros((n, n, 4), np.float32)
    tg[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    t.particles.upload_texels(st)
    t.targets.set_pixels(tg)
    t.timer.time = float(rng.uniform(0, 5000))
    cur, prev = st.copy(), st.copy()
    flow = np.zeros((view[1], view[0], 4), np.float32)
    view_size = list(t.viewSize)

    def check(what):
        assert bits_equal(t.particles.read(0), cur).all(), what + ": buffers[0]"
        assert bits_equal(t.particles.read(1), prev).all(), what + ": buffers[1]"
        assert bits_equal(t.flow.read(), flow).all(), what + ": flow"

    for op_index in range(14):
        op = rng.choice(["step", "step_n", "draw", "ball", "flow_sample", "uniforms"], p=[.25, .25, .2, .08, .1, .12])
        if op == "uniforms":
            choice = int(rng.integers(0, 5))
            if choice == 0:
                t.state["target"] = float(rng.choice([0.0, 0.0005]))
            elif choice == 1:
                t.state["noiseWeight"] = float(rng.choice([0.0, 0.002, 0.004]))
            elif choice == 2:
                t.state["flowWeight"] = float(rng.choice([0.0, 1.0]))
            elif choice == 3:
                t.state["speedLimit"] = float(rng.choice([0.01, 0.004]))
            else:
                t.state["damping"] = float(rng.choice([0.043, 0.03]))
        elif op == "step":
            t.timer.tick()
            t.step()
            prev, cur = cur, oracle.logic_step(oracle_uniforms(oracle, t, n), cur, flow, tg)
        elif op == "step_n":
            k = int(rng.integers(2, 8))
            t.step_n(k)
            time = t.timer.time - k * t.timer.dt
            for _ in range(k):
                time += t.timer.dt
                u = oracle.logic_uniforms(n, n, time, t.timer.dt, view_size=view_size,
                                          **{a: b for a, b in t.state.items() if isinstance(b, (int, float))})
                prev, cur = cur, oracle.logic_step(u, cur, flow, tg)
        elif op == "draw":
            t.draw()
            flow, frags = oracle.flow_deposit(cur, prev, flow, t.timer.time, view_size=view_size,
                                              speedLimit=t.state["speedLimit"])
            assert t.fragments == frags
        elif op == "ball":
            un = dict(radius=float(rng.uniform(0.2, 0.9)), speed=float(rng.uniform(0, 0.01)))
            spawnBall(None, dict(uniforms=un)).spawn(t)
            prev, cur = cur, oracle.spawn_ball(n, n, **un)
        else:
            sp = PixelSpawner(None, dict(shader=flow_sample_frag(), buffer=t.flow,
                                         spawnSize=[1 / view_size[0], -1 / view_size[1]], speed=1.0, bias=1.0))
            sp.spawn(t)
            u = oracle.spawn_sample_uniforms(n, n, t.timer.time, 5, 0, spawnSize=sp.spawnSize, jitter=sp.jitter,
                                             speed=1.0, bias=1.0, flowDecay=t.state["flowDecay"],
                                             spawnMatrix=sp.spawnMatrix)
            prev, cur = cur, oracle.spawn_sample(u, cur, flow)
        check("seed %d op %d (%s)" % (seed, op_index, op))
    t.dispose()
