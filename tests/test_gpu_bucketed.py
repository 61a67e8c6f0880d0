"""The tile-sorted slot layout must be invisible in the results.  The integrator, optical-flow, spawn, deposit and fuzz suites
run once more with it forced on and a 2-step re-sort period (the "bucket" variant of tests/conftest.py): sorting,
re-sorting, the permuted launch and the un-permute on read-back against the golden vectors and the oracle.  Here: the
sizes at which the default policy sorts by itself."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_auto_bucketing_at_c3_size_is_bit_identical(oracle):
    """4096^2 particles over a 1920x1080 field: the default policy buckets.  Compare three steps
    against an unbucketed context (TH_BUCKET cannot change inside one process, so the reference
    run here is a row band of the same state computed by the oracle)."""
    import numpy as np
    import tendrils_amd as ta
    from helpers import bits_equal
    from tendrils_amd.tendrils import View
    n = 4096
    rng = np.random.default_rng(31)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2)).astype(np.float32)
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2)).astype(np.float32)
    st[rng.random((n, n)) < 0.02] = [-1e6, -1e6, 0, 0]
    fl = np.zeros((1080, 1920, 4), np.float32)
    fl[..., :2] = rng.uniform(-.02, .02, (1080, 1920, 2))
    fl[..., 2] = 990.0
    t = ta.Tendrils(View(1920, 1080))
    t.resize()
    t.setup(n)
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.timer.time = 1000.0
    rows = slice(2000, 2064)
    cur = st[rows]
    for _ in range(3):
        t.timer.tick()
        t.step()
        u = oracle.logic_uniforms(n, n, t.timer.time, t.timer.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        cur = oracle.logic_step(u, cur, fl, y0=rows.start)
    got = t.particles.read(0)
    stats = t.particles.stats(t.state["speedLimit"])
    t.dispose()
    assert bits_equal(got[rows], cur).all()
    inert = (st[..., 0] == -1e6) & (st[..., 1] == -1e6)
    assert bits_equal(got[inert], st[inert]).all()
    assert stats["live"] == int((~inert).sum()) and stats["particles"] == n * n


def test_fused_launch_at_c3_size_is_bit_identical(oracle):
    """The launch bench.py times - step_n(20) at 4096^2 over a 1920x1080 field: ONE logic_fused_kernel pass of 20 steps on
    (auto-)tile-sorted slots - against 20 oracle steps of a 64-row band, for both states the ring keeps."""
    import numpy as np
    import tendrils_amd as ta
    from helpers import bits_equal
    from tendrils_amd.tendrils import View
    n, steps = 4096, 20
    rng = np.random.default_rng(32)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2)).astype(np.float32)
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2)).astype(np.float32)
    st[rng.random((n, n)) < 0.01] = [-1e6, -1e6, 0, 0]
    fl = np.zeros((1080, 1920, 4), np.float32)
    fl[..., :2] = rng.uniform(-.02, .02, (1080, 1920, 2))
    fl[..., 2] = 990.0
    t = ta.Tendrils(View(1920, 1080))
    t.resize()
    t.setup(n)
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.timer.time = 1000.0
    rows = slice(1500, 1564)
    cur = st[rows]
    prev = cur
    tm = ta.Timer(0, 0)
    tm.step, tm.time = t.timer.step, t.timer.time
    t.step_n(steps)
    info_sorted = len(t.particles.buffers)
    for _ in range(steps):
        tm.tick()
        u = oracle.logic_uniforms(n, n, tm.time, tm.dt, view_size=t.viewSize,
                                  **{k: v for k, v in t.state.items() if isinstance(v, (int, float))})
        prev = cur
        cur = oracle.logic_step(u, cur, fl, y0=rows.start)
    assert t.timer.time == tm.time
    got0, got1 = t.particles.read(0), t.particles.read(1)
    t.dispose()
    assert info_sorted == 2
    assert bits_equal(got0[rows], cur).all()
    assert bits_equal(got1[rows], prev).all()
