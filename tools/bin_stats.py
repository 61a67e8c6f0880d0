"""How crowded the draw target gets in bench.py's frame loop: particles per flow texel / per 16x16 bin after k frames."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import tendrils_amd as ta
from tendrils_amd.tendrils import View
t = ta.Tendrils(View(1920, 1080)); t.resize(); t.setup(4096)
t.particles.upload_texels(bench.synth_state(0)); t.timer.time = 1000.0
t.particles.draw_pipeline("stream")
done = 0
for frames in (5, 15, 35, 65, 150):
    while done < frames:
        t.timer.tick(); t.step(); t.draw(); done += 1
    st = t.particles.read(0)
    x = (st[..., 0] * t.viewSize[0] + 1) * 960; y = (st[..., 1] * t.viewSize[1] + 1) * 540
    ok = (x >= 0) & (x < 1920) & (y >= 0) & (y < 1080) & (st[:2048, :, 0] > -1e5 if False else True)
    ok[2048:] = False                      # only rows < H/2 draw lines
    tx = x[ok].astype(np.int64); ty = y[ok].astype(np.int64)
    tex = np.bincount(ty * 1920 + tx, minlength=1920 * 1080)
    bins = np.bincount((ty >> 4) * 120 + (tx >> 4), minlength=120 * 68)
    big = bins > 4096
    per_bin_max = np.zeros(120 * 68, np.int64)
    np.maximum.at(per_bin_max, (np.arange(1920 * 1080) // 1920 >> 4) * 120 + (np.arange(1920 * 1080) % 1920 >> 4), tex)
    long_ = per_bin_max > 256
    print(json.dumps({"frames": frames, "fragments": t.fragments, "in_view_drawing": int(ok.sum()), "max_per_texel": int(tex.max()),
                      "max_per_bin": int(bins.max()), "bins_over_4096": int(big.sum()), "share_in_big_bins": float(bins[big].sum() / max(bins.sum(), 1)),
                      "bins_with_run_over_256": int(long_.sum()), "bins_with_run_over_64": int((per_bin_max > 64).sum()),
                      "p50_bin": int(np.median(bins)), "p99_bin": int(np.percentile(bins, 99))}))
t.dispose()
