#!/usr/bin/env python3
"""Probe of the fused integrator launch at C3: launch length, idle gap, back-to-back rate."""
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
mode = ta.TH_MODE_FAST if "--fast" in sys.argv else ta.TH_MODE_EXACT
opts = ta.defaults(); opts.update(device=0, mode=mode)
t = ta.Tendrils(View(B.FLOW_W, B.FLOW_H), opts); t.resize(); t.setup(N)
ctx = t.particles._ctx
t.particles.upload_texels(B.synth_state(0))
f0, f1 = B.synth_frames()
of = OpticalFlow(t, uniforms=dict(speed=0.08, offset=0.1, scaleUV=[-1, -1])); of.resize([B.FLOW_W, B.FLOW_H])
of.set_pixels(f0); of.step(); of.set_pixels(f1)
t.timer.time = 1000.0
of.update(dict(speedLimit=t.state["speedLimit"], time=1000.0, viewSize=t.viewSize)); of.render()

def timed(n, reps, gap=0.0):
    k_ms, k_n = C.c_float(), C.c_int32()
    _capi.call("th_kernel_timing", ctx, 1)
    for _ in range(reps):
        if gap: t.particles.sync(); time.sleep(gap)
        t.step_n(n)
    _capi.call("th_kernel_timing_read", ctx, C.byref(k_ms), C.byref(k_n))
    _capi.call("th_kernel_timing", ctx, 0)
    return k_ms.value, k_n.value

t.step_n(32); t.step_n(32); t.particles.sync()
short = "--short" in sys.argv
for n in ((32, 20, 32, 20) if short else (32, 20, 16, 8, 4, 2, 32)):
    ms, k = timed(n, 12)
    print("back-to-back n=%2d: %.4f ms/launch  %.4f ms/step (%d launches)" % (n, ms, ms / n, k), flush=True)
for gap in (() if short else (0.001, 0.01, 0.1, 0.5)):
    ms, k = timed(20, 6, gap)
    print("idle %.3f s then n=20: %.4f ms/launch %.4f ms/step" % (gap, ms, ms / 20), flush=True)
# wall clock of lone 20-step launch after pre-roll (driver-like)
for pre in (() if short else (0, 2, 8, 16)):
    for _ in range(pre): t.step_n(32)
    t.particles.sync()
    t0 = time.perf_counter(); t.step_n(20); t.particles.sync(); w = time.perf_counter() - t0
    print("pre-roll %2d x32 steps, sync, wall of step_n(20): %.4f ms/step" % (pre, w / 20 * 1e3), flush=True)
# single-step kernel
k_ms, k_n = C.c_float(), C.c_int32()
for rep in range(2):
    _capi.call("th_kernel_timing", ctx, 1)
    for _ in range(32):
        t.timer.tick(); t.step()
    _capi.call("th_kernel_timing_read", ctx, C.byref(k_ms), C.byref(k_n))
    _capi.call("th_kernel_timing", ctx, 0)
    print("single-step kernel: %.4f ms (%d launches)" % (k_ms.value, k_n.value), flush=True)
t.dispose()
