"""Tendrils.draw() of a sharded job at world size 1 (all a 1-GPU box allows) against the local draw() at C3: what the
library's own exchange (th_draw_sharded over its RCCL communicator: edge rows, counts, fragments, all-gather) costs when
there is nobody to exchange with."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import workload as bench  # noqa: E402  (the synthetic C3 inputs)
import tendrils_amd as ta  # noqa: E402
from tendrils_amd import _capi, sharding  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

N = int(os.environ.get("TH_N", "4096"))
bench.N = N
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
out = {}
for mode in (sys.argv[2:] or ["local", "sharded"]):
    t = ta.Tendrils(View(1920, 1080))
    t.resize(); t.setup(N)
    t.particles.upload_texels(bench.synth_state(0))
    t.timer.time = 1000.0
    ctx = t.particles._ctx
    if mode == "sharded":
        ident = sharding.comm_id()
        buf = (C.c_ubyte * _capi.COMM_ID_BYTES).from_buffer_copy(ident)
        _capi.call("th_comm_init", ctx, buf, 0, 1)
    ms = C.c_float()
    draw = (lambda: t.draw()) if mode == "local" else (lambda: sharding.draw_sharded_native(t, view=True))
    for _ in range(5):
        t.timer.tick(); t.step(); t.line_widths(); draw()
    d, s = [], []
    for _ in range(frames):
        t.timer.tick()
        _capi.call("th_timer_start", ctx); t.step(); _capi.call("th_timer_stop", ctx, C.byref(ms)); s.append(ms.value)
        _capi.call("th_timer_start", ctx); draw(); _capi.call("th_timer_stop", ctx, C.byref(ms)); d.append(ms.value)
    out[mode] = {"draw_both_ms": float(np.median(d)), "step_ms": float(np.median(s))}
    t.dispose()
print(json.dumps(out))
