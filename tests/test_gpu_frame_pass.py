"""The frame pass (TH_OPT_FRAME_FUSE; th_bins.hip): in a step(); draw() loop over tile-sorted slots the step is planned by
th_step and carried out by the draw's first pass over the slots - one kernel moves the particles on and emits their lines
from registers.  Everything must be bit for bit what the two launches leave: particles, flow field, view buffer, fragment
counts - through re-sorts (single-step launches around them stay on their own), with calls in between that need the step
done first (read-backs, spawns, uploads, statistics), with targets and without noise, in fast mode, and on the crowded
targets that make the pass repeat itself with a grown pool."""
import ctypes as C

import numpy as np
import pytest

from helpers import bits_equal

pytestmark = pytest.mark.gpu


def start(n, view, seed, fuse, overrides=None, mode=None, resort=5, pool=None):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    opts = ta.defaults()
    if mode is not None:
        opts["mode"] = mode
    t = ta.Tendrils(View(*view), opts)
    t.resize()
    t.setup(n)
    t.particles.option("bucket", 1)
    t.particles.option("resort_steps", resort)
    t.particles.option("frame_fuse", 1 if fuse else 0)
    if pool:
        t.particles.option("bins_pool", pool)
    t.particles.draw_pipeline("bins")
    rng = np.random.default_rng(seed)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-0.9, 0.9, (n, n, 2)) * [1.0, view[1] / view[0]]
    st[..., 2:] = rng.uniform(-.008, .008, (n, n, 2))
    st[rng.random((n, n)) < 0.05] = [-1e6, -1e6, 0, 0]
    t.particles.upload_texels(st)
    t.state.update(overrides or {})
    t.timer.time = 1000.0
    return t


def frame_passes(t):
    from tendrils_amd import _capi
    info = _capi.DrawInfo()
    _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
    return info.frame_passes


@pytest.mark.parametrize("n,view,overrides,mode", [(256, (96, 54), {}, None), (256, (96, 54), {"noiseWeight": 0.0}, None),
                                                   (128, (64, 64), {"target": 0.0005}, None), (256, (96, 54), {}, "fast"),
                                                   (200, (96, 54), {}, None)])
def test_frame_loop_with_and_without_the_frame_pass(n, view, overrides, mode):
    """n = 200: not a power of two - the frame pass stays out (its integrator variant is the power-of-two one)."""
    import tendrils_amd as ta
    m = None if mode is None else ta.TH_MODE_FAST
    outs = []
    for fuse in (False, True):
        t = start(n, view, 3 * n, fuse, overrides, m)
        if overrides.get("target"):
            tg = np.zeros((n, n, 4), np.float32)
            tg[..., :2] = np.random.default_rng(1).uniform(-0.5, 0.5, (n, n, 2))
            t.targets.set_pixels(tg)
        log = []
        for k in range(14):                         # (re-sorted every 5 single steps: frames 0, 5, 10 ... and their neighbours step on their own)
            t.timer.tick()
            t.step()
            t.draw()
            log.append(t.fragments)
        outs.append((t.particles.read(0), t.particles.read(1), t.flow.read(), t.read_view(), log, frame_passes(t)))
        t.dispose()
    a, b = outs
    assert a[5] == 0 and (b[5] >= 6 if n != 200 else b[5] == 0), (a[5], b[5])
    assert a[4] == b[4] and min(a[4]) > 1000
    assert bits_equal(a[0], b[0]).all() and bits_equal(a[1], b[1]).all()
    assert bits_equal(a[2], b[2]).all() and (a[3] == b[3]).all() and a[3].any()


def test_a_planned_step_is_carried_out_by_whatever_comes_next(oracle):
    """Between step() and draw(): a read-back, statistics, an upload, a spawn, another step, th_step_n, a view pass alone -
    each finds the step done; the results are those of a context that never plans ahead."""
    from tendrils_amd import _capi
    from tendrils_amd.spawn.ball import spawnBall
    n, view = 128, (96, 54)
    outs = []
    for fuse in (False, True):
        t = start(n, view, 11, fuse)
        seen = []
        for k in range(12):
            t.timer.tick()
            t.step()
            if k == 3:
                seen.append(t.particles.read(0))               # read-back
            elif k == 4:
                seen.append(t.particles.stats(t.state["speedLimit"]))
            elif k == 5:
                patch = np.full((4, 8, 4), 0.25, np.float32)     # an upload into the state just stepped: 8 x 4 texels at (16, 32)
                _capi.call("th_upload_state", t.particles._ctx, 0, patch.ctypes.data_as(_capi._fp), 16, 32, 8, 4)
            elif k == 6:
                spawnBall(None, dict(uniforms=dict(radius=0.2, speed=0.01))).spawn(t)
            elif k == 7:
                t.timer.tick()
                t.step()                                        # two steps, one draw
            elif k == 8:
                t.step_n(3)
            elif k == 9:
                t.renderView = False
                t.draw()
                t.renderView = True
                continue
            t.draw()
        outs.append((t.particles.read(0), t.flow.read(), t.read_view(), seen, frame_passes(t)))
        t.dispose()
    a, b = outs
    assert a[4] == 0 and b[4] >= 2
    assert bits_equal(a[0], b[0]).all() and bits_equal(a[1], b[1]).all() and (a[2] == b[2]).all()
    assert bits_equal(a[3][0], b[3][0]).all() and a[3][1] == b[3][1]


def test_frame_pass_repeats_itself_on_a_dry_pool_without_stepping_twice():
    """A pool of 8 pages runs dry in the first frame pass: the pass is repeated with a grown pool - as a plain pass over
    the two states in memory, the step is done."""
    n, view = 256, (96, 54)
    outs = []
    for fuse, pool in ((False, None), (True, 8)):
        t = start(n, view, 5, fuse, pool=pool)
        for k in range(4):
            t.timer.tick()
            t.step()
            t.draw()
        outs.append((t.particles.read(0), t.flow.read(), t.read_view(), frame_passes(t)))
        t.dispose()
    a, b = outs
    assert b[3] >= 2
    assert bits_equal(a[0], b[0]).all() and bits_equal(a[1], b[1]).all() and (a[2] == b[2]).all()


def test_frame_pass_at_c3_size_through_the_node_free_python_host():
    """4096^2 over 1080p, the default policy (nothing forced): after the first frame the loop runs as frame passes; four
    frames equal the two-launch loop bit for bit."""
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    n, frames = 4096, 5
    rng = np.random.default_rng(41)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2)).astype(np.float32)
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2)).astype(np.float32)
    st[rng.random((n, n)) < 0.01] = [-1e6, -1e6, 0, 0]
    outs = []
    for fuse in (False, True):
        t = ta.Tendrils(View(1920, 1080))
        t.resize()
        t.setup(n)
        t.particles.option("frame_fuse", 1 if fuse else 0)
        t.particles.upload_texels(st)
        t.timer.time = 1000.0
        for _ in range(frames):
            t.timer.tick()
            t.step()
            t.draw()
        outs.append((t.particles.read(0)[::7], t.flow.read(), t.read_view(), t.fragments, frame_passes(t)))
        t.dispose()
    a, b = outs
    assert a[4] == 0 and b[4] == frames - 1
    assert a[3] == b[3] > 1_000_000
    assert bits_equal(a[0], b[0]).all() and bits_equal(a[1], b[1]).all() and (a[2] == b[2]).all()
