#!/usr/bin/env python3
"""Where a bin's workgroup of bins_blend_kernel spends its life: a diagnostic build of th_bins.hip (-DTH_BLEND_STAMPS:
cycle stamps at the kernel's barriers, summed over the launch; tools/build_variant_libs.sh builds it into tools/bin/, tools/gpu_ab_prev.sh runs
this with TH_LIB pointing there) under the C3 frame loop.  Prints cycles per phase per workgroup and their shares.

    TH_LIB=tools/bin/libtendrils_hip_stamps.so python3 tools/blend_stamps.py [frames] [settle frames]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import workload as bench  # noqa: E402
import tendrils_amd as ta  # noqa: E402
from tendrils_amd import _capi  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t = ta.Tendrils(View(1920, 1080))
t.resize()
t.setup(bench.N)
t.particles.upload_texels(bench.synth_state(0))
t.timer.time = 1000.0
t.renderView = True
lib = _capi.load()
stamps = (C.c_ulonglong * 16)()
for _ in range(settle):
    t.timer.tick(); t.step(); t.draw()
_capi.call("th_sync", t.particles._ctx)
assert lib.th_debug_blend_stamps(stamps) == 0
for _ in range(frames):
    t.timer.tick(); t.step(); t.draw()
_capi.call("th_sync", t.particles._ctx)
assert lib.th_debug_blend_stamps(stamps) == 0
v = np.array(list(stamps), np.float64)
names = ["cursors + barrier", "places, page table, keys (global loads)", "count atomics + scan", "stream indices into runs", "rank by counting",
         "places into blend order", "thread 0's own run blended", "... the workgroup's longest run", "store"]
wgs = max(v[12], 1.0)
total = v[:9].sum()
print("frames %d (after %d): %.0f workgroups with fragments per draw, %.0f fragments per such bin, longest run %.1f on average" %
      (frames, settle, wgs / frames, v[13] / wgs, v[14] / wgs))
for k, name in enumerate(names):
    print("  %-44s %9.0f cycles per workgroup  %5.1f %%" % (name, v[k] / wgs, 100.0 * v[k] / total))
print("  %-44s %9.0f cycles per workgroup (%.1f us at 2.4 GHz; s_memtime counts at 100 MHz on gfx950 if these look 24x small)" % ("total", total / wgs, total / wgs / 2400.0))
