#!/bin/bash
python -m pytest tests/test_gpu_logic_parity.py tests/test_gpu_bucketed.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY'
import time, numpy as np, tendrils_amd as ta
from tendrils_amd.tendrils import View
for n in (256, 1024, 4096):
    for graph in (0, 1):
        t = ta.Tendrils(View(480, 270)); t.resize(); t.setup(n)
        rng = np.random.default_rng(1)
        st = np.empty((n, n, 4), np.float32); st[..., :2] = rng.uniform(-1, 1, (n, n, 2)); st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
        t.particles.upload_texels(st)
        K, B = 64, 20
        def run():
            if graph: t.step_n(K)
            else:
                for _ in range(K): t.timer.tick(); t.step()
        run(); t.particles.sync()
        t0 = time.perf_counter()
        for _ in range(B): run()
        t.particles.sync()
        dt = (time.perf_counter() - t0) / (B * K)
        print("N=%4d^2  %s  %.2f us/step  %.2f G particle-steps/s" % (n, "graph" if graph else "loop ", dt * 1e6, n * n / dt / 1e9))
        t.dispose()
PY
