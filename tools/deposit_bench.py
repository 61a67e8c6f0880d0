"""Frame loop of the reference at C3: Tendrils.step() + Tendrils.draw() (flow deposit) per frame.
Prints ms per frame part (HIP events on the context's stream) and fragments per frame."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import workload as bench  # noqa: E402  (the synthetic C3 inputs)
import tendrils_amd as ta  # noqa: E402
from tendrils_amd import _capi  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

N = int(os.environ.get("TH_N", "4096"))
bench.N = N
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
in_view = "--in-view" in sys.argv

opts = ta.defaults()
WIDTH = float(os.environ.get("TH_WIDTH", "0"))     # > 0: a GL that honours gl.lineWidth - flowWidth = lineWidth = TH_WIDTH
if WIDTH > 0:
    opts["lineWidthRange"] = (1, 64)
    opts["state"]["flowWidth"] = opts["state"]["lineWidth"] = WIDTH
t = ta.Tendrils(View(1920, 1080), opts)
t.resize()
t.setup(N)
st = bench.synth_state(0)
if in_view:
    st[..., 1] *= np.float32(0.56)
t.particles.upload_texels(st)
t.particles.draw_pipeline(os.environ.get("TH_PIPE", "auto"))     # "stream" | "bins" | "auto"
t.timer.time = 1000.0
ctx = t.particles._ctx
ms = C.c_float()


def timed(fn):
    _capi.call("th_timer_start", ctx)
    fn()
    _capi.call("th_timer_stop", ctx, C.byref(ms))
    return ms.value


for _ in range(5):                      # warm-up: the wake builds up, buffers are sized
    t.timer.tick(); t.step(); t.draw()
step_ms, draw_ms, view_ms, both_ms, frags, pipes, crowded = [], [], [], [], [], [], []
info = _capi.DrawInfo()
both = "--both" in sys.argv              # both passes in one call (Tendrils.draw() with renderView) instead of one after the other
for _ in range(frames):
    t.timer.tick()
    step_ms.append(timed(t.step))
    if both:
        t.renderView = True
        both_ms.append(timed(t.draw))
        frags.append(t.fragments)
        _capi.call("th_draw_query", ctx, C.byref(info))
        pipes.append(info.pipeline); crowded.append(info.crowded_fragments / max(info.fragments, 1))
        continue
    t.renderView = False
    draw_ms.append(timed(t.draw))           # the flow pass (what feeds the next step)
    frags.append(t.fragments)
    t.renderView = True
    u, n = t.render_uniforms(), C.c_uint64(0)
    view_ms.append(timed(lambda: _capi.call("th_view_draw", ctx, C.byref(u), C.byref(n))))     # the view pass
if os.environ.get("TH_BENCH_TRACE") and both_ms:        # how the frame changes as the wake crowds the particles: averages per 50 frames
    for k in range(0, len(both_ms), 50):
        print("frames %4d-%4d  step %.3f  draw(both) %.3f  fragments %.2f M  binned %3d%%  crowded share %.2f" % (k, min(k + 50, len(both_ms)) - 1, np.mean(step_ms[k:k + 50]),
              np.mean(both_ms[k:k + 50]), np.mean(frags[k:k + 50]) / 1e6, 100 * np.mean(pipes[k:k + 50]), np.mean(crowded[k:k + 50])))
wall_ms = None
if "--wall" in sys.argv:                 # ... and 200 more frames as a host runs them (no event pair, no sync but the draw's own read-back)
    import time
    _capi.call("th_sync", ctx)
    t0 = time.perf_counter()
    for _ in range(200):
        t.timer.tick(); t.step(); t.draw()
    _capi.call("th_sync", ctx)
    wall_ms = (time.perf_counter() - t0) / 200 * 1e3
if os.environ.get("TH_BENCH_FRAMES") and both_ms:       # e.g. 640:710 - those frames one by one
    lo, hi = (int(x) for x in os.environ["TH_BENCH_FRAMES"].split(":"))
    for k in range(lo, min(hi, len(both_ms))):
        print("frame %4d  step %.3f  draw(both) %.3f  fragments %.2f M  crowded share %.2f" % (k, step_ms[k], both_ms[k], frags[k] / 1e6, crowded[k]))
stats = t.particles.stats(t.state["speedLimit"])
print(json.dumps({"pipeline": os.environ.get("TH_PIPE", "auto"), "particles": N * N, "flow": [1920, 1080], "frames": frames, "in_view": in_view,
                  "step_ms": float(np.mean(step_ms)), "draw_ms": float(np.mean(draw_ms)) if draw_ms else None, "view_ms": float(np.mean(view_ms)) if view_ms else None,
                  "draw_both_ms": float(np.mean(both_ms)) if both_ms else None,
                  "fragments_per_frame": float(np.mean(frags)), "frames_per_s": 1e3 / float(np.mean(step_ms) + (np.mean(both_ms) if both_ms else np.mean(draw_ms))),
                  "wall_ms_per_frame": wall_ms, "live": stats["live"], "nan": stats["nan"]}))
t.dispose()
