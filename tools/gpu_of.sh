#!/bin/bash
python -m pytest tests/test_gpu_optical_flow.py tests/test_node_host.py -m gpu -q 2>&1 | tail -3
python - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, 'tests')
import tendrils_amd as ta
from tendrils_amd.optical_flow import OpticalFlow
from tendrils_amd.tendrils import View
from helpers import synth_frame
W, H = 1920, 1080
f0, f1 = synth_frame(W, H, 5), synth_frame(W, H, 5, shift8=(12, 6))
t = ta.Tendrils(View(W, H)); t.resize(); t.setup(64)
of = OpticalFlow(t, uniforms=dict(speed=0.08, scaleUV=[-1, -1]))
of.resize([W, H]); of.set_pixels(f0); of.step(); of.set_pixels(f1)
for name, off in (("offset 0.1 (direct path)", 0.1), ("offset 1/1920 (LDS-tiled path)", 1.0 / 1920), ("offset 4/1920 (LDS-tiled path)", 4.0 / 1920)):
    of.update(dict(speedLimit=0.01, time=1000.0, viewSize=t.viewSize, offset=off))
    for _ in range(5): of.render()
    t.particles.sync(); t0 = time.perf_counter()
    K = 200
    for _ in range(K): of.render()
    t.particles.sync()
    print("%-34s %.2f us per pass" % (name, (time.perf_counter() - t0) / K * 1e6))
t.dispose()
PY
