"""FAST mode against EXACT mode over fused launches: drift of positions / velocities after k steps (calibrates the bound in
tests/test_gpu_logic_parity.py::test_fast_mode_drift_over_fused_launches)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_logic_parity as T
import tendrils_amd as ta

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st, fl = T.seeded_case(n, 99, inert=0.02, pos_range=1.0)
outs = {}
for mode in (ta.TH_MODE_EXACT, ta.TH_MODE_FAST):
    t = T.make_tendrils(n, (96, 54), (96, 54), {}, mode)
    t.particles.upload_texels(st); t.flow.set_pixels(fl); t.timer.time = 4000.0
    seq = []
    for k in (1, 4, 15, 20, 60, 156):
        t.step_n(k); seq.append((k, t.particles.read(0).copy()))
    outs[mode] = seq; t.dispose()
done = 0
for (k, a), (_, b) in zip(outs[ta.TH_MODE_EXACT], outs[ta.TH_MODE_FAST]):
    done += k
    nan_same = (np.isnan(a) == np.isnan(b)).all(-1)
    d = np.abs(np.nan_to_num(a) - np.nan_to_num(b))
    far = (np.abs(a[..., :2]) > 1e5).any(-1) | (np.abs(b[..., :2]) > 1e5).any(-1)
    dd = d[~far & nan_same]
    print("steps %4d: nan pattern differs %d, parked differs %d; |d pos| max %.3g p99.99 %.3g median %.3g; |d vel| max %.3g p99.99 %.3g" % (
        done, (~nan_same).sum(), (far & (np.abs(a[..., 0] - b[..., 0]) > 1)).sum(), dd[:, :2].max(), np.quantile(dd[:, :2], 0.9999), np.median(dd[:, :2]),
        dd[:, 2:].max(), np.quantile(dd[:, 2:], 0.9999)))
