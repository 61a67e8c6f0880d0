#!/usr/bin/env python3
"""Single-step integrator at C3: kernel time, whole-loop time (re-sorts included) and window misses."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import workload as B
import tendrils_amd as ta
from tendrils_amd import _capi
from tendrils_amd.tendrils import View
from tendrils_amd.optical_flow import OpticalFlow

N = B.N
steps = int(os.environ.get("PROBE_STEPS", "64"))
opts = ta.defaults(); opts.update(device=0, mode=ta.TH_MODE_EXACT)
t = ta.Tendrils(View(B.FLOW_W, B.FLOW_H), opts); t.resize(); t.setup(N)
ctx = t.particles._ctx
st = B.synth_state(0)
if "--in-view" in sys.argv:
    st[..., 1] *= np.float32(0.56)
t.particles.upload_texels(st)
f0, f1 = B.synth_frames()
of = OpticalFlow(t, uniforms=dict(speed=0.08, offset=0.1, scaleUV=[-1, -1])); of.resize([B.FLOW_W, B.FLOW_H])
of.set_pixels(f0); of.step(); of.set_pixels(f1)
t.timer.time = 1000.0
of.update(dict(speedLimit=t.state["speedLimit"], time=1000.0, viewSize=t.viewSize)); of.render()
t.step_n(32); t.step_n(32)      # settle the velocities like the bench does
if "--flow-only" in sys.argv:
    t.state["noiseWeight"] = 0
info = _capi.SlotOrderInfo()
for rep in range(3):
    k_ms, k_n = C.c_float(), C.c_int32()
    _capi.call("th_slot_order", ctx, C.byref(info)); s0 = info.sorts
    t.particles.sync()
    _capi.call("th_kernel_timing", ctx, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        t.timer.tick(); t.step()
    t.particles.sync()
    wall = (time.perf_counter() - t0) / steps * 1e3
    _capi.call("th_kernel_timing_read", ctx, C.byref(k_ms), C.byref(k_n))
    _capi.call("th_kernel_timing", ctx, 0)
    _capi.call("th_slot_order", ctx, C.byref(info))
    print("single step: kernel %.4f ms, loop %.4f ms/step (%d launches); sorted buffers %d, sorts %d, misses since sort %d (%.2f%% of a pass) after %d steps"
          % (k_ms.value, wall, k_n.value, info.sorted_buffers, info.sorts - s0, info.window_misses, 100.0 * info.window_misses / (N * N), info.steps_since_sort), flush=True)
t.dispose()
