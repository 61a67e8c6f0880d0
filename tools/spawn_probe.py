#!/usr/bin/env python3
"""Respawn passes at C3 (4096^2 particles, flow 1920x1080): milliseconds per pass (HIP event timer of the context)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import workload as B
import tendrils_amd as ta
from tendrils_amd import _capi
from tendrils_amd.tendrils import View
from tendrils_amd.spawn import PixelSpawner, flow_sample_frag, data_sample_frag, spawnBall

t = ta.Tendrils(View(B.FLOW_W, B.FLOW_H)); t.resize(); t.setup(B.N)
ctx = t.particles._ctx
t.particles.upload_texels(B.synth_state(0))
t.flow.set_pixels(B.synth_flow(1000.0))
t.timer.time = 1000.0
ms = C.c_float()
def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        _capi.call("th_timer_start", ctx); fn(); _capi.call("th_timer_stop", ctx, C.byref(ms)); out.append(ms.value)
    return min(out), float(np.mean(out))
ball = spawnBall(None, dict(uniforms=dict(radius=0.3, speed=0.005)))
print("spawn-ball      best %.3f ms mean %.3f" % timed(lambda: ball.spawn(t)))
fs = PixelSpawner(None, dict(shader=flow_sample_frag(), buffer=t.flow))
print("flow best-sample best %.3f ms mean %.3f" % timed(lambda: fs.spawn(t)))
ds = PixelSpawner(None, dict(shader=data_sample_frag(), buffer=t.particles.buffers[0]))      # (src/demo.main.js:433-441: a ring buffer as the data texture)
try:
    print("data best-sample best %.3f ms mean %.3f" % timed(lambda: ds.spawn(t)))
except Exception as e:
    print("data best-sample:", e)
print("spawn-init      best %.3f ms mean %.3f" % timed(lambda: t.spawn()))
print("stats           best %.3f ms mean %.3f" % timed(lambda: t.particles.stats(0.01)))
t.dispose()
