"""Wall-clock frame time of the reference's loop - timer.tick(); step(); draw() - against the GPU time of its parts (HIP events):
what the host adds (ctypes, launches, the draw's read-back)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import workload as bench  # noqa: E402  (the synthetic C3 inputs)
import tendrils_amd as ta  # noqa: E402
from tendrils_amd import _capi  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

N = int(os.environ.get("TH_N", "4096"))
bench.N = N
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t = ta.Tendrils(View(1920, 1080))
t.resize(); t.setup(N)
t.particles.upload_texels(bench.synth_state(0))
t.timer.time = 1000.0
ctx = t.particles._ctx
for _ in range(5):
    t.timer.tick(); t.step(); t.draw()
t.particles.sync()
t0 = time.perf_counter()
for _ in range(frames):
    t.timer.tick(); t.step(); t.draw()
t.particles.sync()
wall = (time.perf_counter() - t0) / frames * 1e3
# the same number of frames again with the GPU time of every call measured (events on the context's stream)
ms = C.c_float()
gpu = []
for _ in range(frames):
    t.timer.tick()
    _capi.call("th_timer_start", ctx); t.step(); t.draw(); _capi.call("th_timer_stop", ctx, C.byref(ms))
    gpu.append(ms.value)
# host time of the calls alone (no GPU wait beyond what the calls do themselves)
h_step, h_draw = [], []
for _ in range(frames):
    t.timer.tick()
    a = time.perf_counter(); t.step(); b = time.perf_counter(); t.draw(); c = time.perf_counter()
    h_step.append((b - a) * 1e3); h_draw.append((c - b) * 1e3)
print(json.dumps({"frames": frames, "wall_ms_per_frame": wall, "gpu_ms_per_frame": float(np.mean(gpu)),
                  "host_ms_in_step_call": float(np.median(h_step)), "host_ms_in_draw_call": float(np.median(h_draw))}))
t.dispose()
