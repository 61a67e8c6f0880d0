"""The library a product host ships - `make release`: no in-process transport, no fault injection (tests/test_release_build.py
checks what it exports) - RUN on the GPU: every other -m gpu test and bench.py load the testing build.  A child process with
TH_LIB pointing at it: the smoke step, a fused 20-step launch and a draw() with both passes, each against the restatement."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RELEASE = os.path.join(ROOT, "tendrils_amd", "lib", "release", "libtendrils_hip.so")

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
import tendrils_amd as ta
from tendrils_amd import _capi
from tendrils_amd.tendrils import View
from helpers import bits_equal
lib = _capi.load()
assert os.path.realpath(lib._name) == os.path.realpath(RELEASE), lib._name
assert not hasattr(lib, "th_comm_loopback_id")
n, view = 256, (96, 54)
rng = np.random.default_rng(2024)
st = np.zeros((n, n, 4), np.float32)
st[..., :2] = rng.uniform(-0.95, 0.95, (n, n, 2)) * [1.0, view[1] / view[0]]
st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
st[rng.random((n, n)) < 0.05] = [-1e6, -1e6, 0, 0]
size = (1.0, view[0] / view[1])
t = ta.Tendrils(View(*view)); t.resize(); t.setup(n)
t.particles.upload_texels(st)
t.timer.time = 3000.0
cur, prev, flow, time, dt = st.copy(), st.copy(), np.zeros((view[1], view[0], 4), np.float32), 3000.0, 1000.0 / 60.0
def step():
    global cur, prev, time
    time += dt
    u = O.logic_uniforms(n, n, time, dt, view_size=size, **O.DEFAULT_STATE)
    prev, cur = cur, O.logic_step(u, cur, flow)
# 1. the smoke step
t.timer.tick(); t.step(); step()
assert bits_equal(t.particles.read(0), cur).all()
# 2. draw() with both passes: the flow pass against the restatement
t.renderView = True
t.draw()
flow, count = O.flow_deposit(cur, prev, flow, time, view_size=size, speedLimit=O.DEFAULT_STATE["speedLimit"])
assert t.fragments == count > 1000 and bits_equal(t.flow.read(), flow).all() and t.read_view().any()
# 3. one fused launch of 20 steps (the bench's), then the statistics it took on the way
t.step_n(20)
for _ in range(20):
    step()
assert bits_equal(t.particles.read(0), cur).all() and bits_equal(t.particles.read(1), prev).all()
s = t.particles.stats(t.state["speedLimit"])
live = (cur[..., 0] != -1e6) | (cur[..., 1] != -1e6)
assert s["live"] == int(live.sum()) and s["particles"] == n * n
# 4. ... and a draw over the result through the binned pipeline
t.particles.draw_pipeline("bins")
t.draw()
flow, count = O.flow_deposit(cur, prev, flow, time, view_size=size, speedLimit=O.DEFAULT_STATE["speedLimit"])
assert t.fragments == count and bits_equal(t.flow.read(), flow).all()
t.dispose()
print("release ok")
'''


def test_release_library_runs_the_hot_path_against_the_restatement(oracle):
    if not os.path.exists(RELEASE):
        subprocess.check_call(["make", "-j3", "-C", os.path.join(ROOT, "tendrils_amd", "csrc"), "release"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, TH_LIB=RELEASE)
    code = "ROOT = %r\nRELEASE = %r\n" % (ROOT, RELEASE) + CHILD
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "release ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
