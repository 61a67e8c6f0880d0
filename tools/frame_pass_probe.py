"""The frame loop - timer.tick(); step(); draw() - at C3 size with the frame pass off / on (TH_OPT_FRAME_FUSE), alternating on one
box: GPU time of step() + draw() per frame (one HIP event pair around both calls), wall time per frame, the first frames and
the crowded regime.   python tools/frame_pass_probe.py [rounds]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tendrils_amd as ta  # noqa: E402
from tendrils_amd import _capi  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
state = bench.synth_state(0)


def run(fuse, settle):
    t = ta.Tendrils(View(1920, 1080))
    t.resize(); t.setup(bench.N)
    t.particles.option("frame_fuse", fuse)
    t.particles.upload_texels(state)
    t.timer.time = 1000.0
    ctx = t.particles._ctx
    ms, gpu = C.c_float(), []
    for _ in range(8 + settle):
        t.timer.tick(); t.step(); t.draw()
    for _ in range(40):
        t.timer.tick()
        _capi.call("th_timer_start", ctx); t.step(); t.draw(); _capi.call("th_timer_stop", ctx, C.byref(ms))
        gpu.append(ms.value)
    t.particles.sync()
    t0 = time.perf_counter()
    for _ in range(40):
        t.timer.tick(); t.step(); t.draw()
    t.particles.sync()
    wall = (time.perf_counter() - t0) / 40 * 1e3
    info = _capi.DrawInfo()
    _capi.call("th_draw_query", ctx, C.byref(info))
    t.dispose()
    return dict(frame_fuse=fuse, after_frames=8 + settle, gpu_ms_per_frame_median=float(np.median(gpu)), gpu_ms_mean=float(np.mean(gpu)),
                wall_ms_per_frame=wall, frame_passes=int(info.frame_passes), fragments=int(info.fragments))


for r in range(rounds):
    for settle in (0, 250):
        for fuse in (0, 1):
            print(json.dumps(run(fuse, settle)), flush=True)
