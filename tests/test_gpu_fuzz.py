"""Differential fuzz: random sequences of the path's operations (step, step_n, draw, ball / flow-sample respawns,
uniform changes) on random shapes, mirrored operation by operation on the CPU restatement.  Everything must stay
bit-identical - this is what shakes out interactions (ring parity after odd step_n, slot layout transitions around
draw(), TARGET / NOISE specialisations, non power-of-two textures, inert and NaN particles)."""
import os

import numpy as np
import pytest

from helpers import bits_equal

pytestmark = pytest.mark.gpu


def oracle_uniforms(oracle, t, n):
    return oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                 **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})


@pytest.mark.parametrize("seed", range(int(os.environ.get("TH_FUZZ_OPS", "6"))))           # (TH_FUZZ_OPS: longer one-off runs)
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


@pytest.mark.parametrize("seed", range(int(os.environ.get("TH_FUZZ_DRAWS", "24"))))        # (TH_FUZZ_DRAWS: longer one-off runs)
def test_random_draws(oracle, seed):
    """draw() (both passes in one call) on random shapes, views and states against the restatement: lines far longer than
    a record holds, lines that cross the view's edge or lie outside it, endpoints up to the rasteriser's limit of
    |clip| = 1024 and beyond (dropped), inert and NaN particles, crowded texels, non-default colours and a colour map."""
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([17, 32, 50, 64, 100]))
    view = [(96, 54), (64, 64), (33, 47), (120, 50), (16, 9)][int(rng.integers(0, 5))]
    prev = np.zeros((n, n, 4), np.float32)
    spread = float(rng.choice([0.2, 1.0, 1.5, 4.0]))
    prev[..., :2] = rng.uniform(-spread, spread, (n, n, 2)) * [1.0, view[1] / view[0]]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += (rng.uniform(-1, 1, (n, n, 2)) * float(rng.choice([0.02, 0.3, 2.5]))).astype(np.float32)
    cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    far = rng.random((n, n)) < 0.03
    cur[far, 0] = rng.choice([-900.0, 700.0, 1023.0, 1500.0, -4000.0], int(far.sum())).astype(np.float32)
    cur[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    if seed % 2:
        cur[0, 0, 1] = np.nan
        prev[1, 0, 0] = np.inf
    base = np.zeros((view[1], view[0], 4), np.float32)
    base[..., :2] = rng.uniform(-.01, .01, (view[1], view[0], 2))
    base[..., 2] = 800.0
    cmap = rng.uniform(0, 1, (3, 5, 4)).astype(np.float32) if seed % 3 == 0 else None
    render = dict(speedLimit=0.01, flowDecay=0.005, speedAlpha=float(rng.choice([1e-6, 0.5])), colorMapAlpha=0.4 if cmap is not None else 0.0,
                  baseColor=[float(v) for v in rng.uniform(0, 1, 4)], flowColor=[float(v) for v in rng.uniform(0, 1, 4)])
    view_size = [1.0, view[0] / view[1]] if seed % 4 else [0.8, 1.3]
    # seeds from 16 on: a context that honours gl.lineWidth, the two passes with widths of their own
    widths = (float(rng.choice([1, 2, 3.5, 7])), float(rng.choice([1, 1.5, 4]))) if seed >= 16 else (1.0, 1.0)
    opts = ta.defaults()
    opts["lineWidthRange"] = (1, 64) if seed >= 16 else (1, 1)
    t = ta.Tendrils(View(*view), opts)
    t.state["flowWidth"], t.state["lineWidth"] = (widths if seed >= 16 else (5, 1))
    t.resize()
    t.setup(n)
    t.viewSize[:] = view_size
    for k, v in render.items():
        t.state[k] = v
    if cmap is not None:
        t.colorMap.set_pixels(cmap)
    t.particles.upload_texels(cur, 0)
    t.particles.upload_texels(prev, 1)
    t.flow.set_pixels(base)
    t.timer.time = 1000.0
    t.draw()
    got_flow, got_view, frags = t.flow.read(), t.read_view(), t.fragments
    t.dispose()
    want_flow, count = oracle.flow_deposit(cur, prev, base, 1000.0, view_size=view_size, speedLimit=render["speedLimit"], line_width=widths[0])
    want_view, count2 = oracle.view_render(cur, prev, np.zeros((view[1], view[0], 4), np.uint8), 1000.0, view_size=view_size,
                                           colormap=cmap, line_width=widths[1], **render)
    assert frags == count and (widths[0] != widths[1] or count == count2)
    assert bits_equal(got_flow, want_flow).all()
    assert (got_view == want_view).all()
