#!/usr/bin/env python3
"""Long frame loops over the shapes whose vertex lookup drifts (tools/gpu_soak.sh's companion for round 6's paths): tick(); step();
draw() with both passes for `frames` frames through the default policy (tile-sorted slots, the binned pipeline: LineSources, the span
kernel's queue, pools and page tables growing as the wake crowds the target) and through the stream-ordered pipeline in texel
order - fragments of every frame, then flow field, view buffer and particles, bit for bit.

    python3 tools/soak_shapes.py [n=3000] [frames=300] [f32|f16]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib.workload import synth_rows  # noqa: E402
import tendrils_amd as ta  # noqa: E402
from tendrils_amd import _capi  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
fmt = sys.argv[3] if len(sys.argv) > 3 else "f32"
view = (1920, 1080) if n >= 2048 else (480, 270)
st = synth_rows(n, n, 4321)
outs = []
for pipeline in ("auto", "stream"):
    opts = ta.defaults()
    opts.update(stateFormat=ta.TH_STATE_F16 if fmt == "f16" else ta.TH_STATE_F32)
    t = ta.Tendrils(View(*view), opts)
    t.resize()
    t.setup(n)
    t.particles.option("bucket", 1 if pipeline == "auto" else 0)
    t.particles.draw_pipeline(pipeline)
    t.particles.upload_texels(st)
    t.timer.time = 1000.0
    t.renderView = True
    info, frags, used, crowd = _capi.DrawInfo(), [], set(), 0
    for _ in range(frames):
        t.timer.tick()
        t.step().draw()
        frags.append(t.fragments)
        _capi.call("th_draw_query", t.particles._ctx, C.byref(info))
        used.add(int(info.pipeline)); crowd = max(crowd, int(info.crowded_fragments))
    outs.append((frags, used, crowd, t.flow.read(), t.read_view(), t.particles.read(0), t.particles.stats(t.state["speedLimit"])))
    t.dispose()
a, b = outs
same = lambda x, y: bool(((x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))).all())
ok = a[0] == b[0] and same(a[3], b[3]) and bool((a[4] == b[4]).all()) and same(a[5], b[5])
print("n %d %s, %d frames: pipelines %s / %s, fragments per draw %.2f M (last), most fragments in crowded bins %d; live %d nan %d; bins == stream-ordered: %s"
      % (n, fmt, frames, sorted(a[1]), sorted(b[1]), a[0][-1] / 1e6, a[2], a[6]["live"], a[6]["nan"], ok))
sys.exit(0 if ok and a[1] == {1} and b[1] == {0} else 1)
