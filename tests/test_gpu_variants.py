"""The alternative paths the environment switches select must give the same results as the defaults: the
suites that exercise them rerun with each switch set (round 2 ran these by hand from tools/gpu_variants.sh)."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

VARIANTS = {
    # no temporal fusion: th_step_n replays a captured graph of single-step launches
    "TH_FUSE=0": ["test_gpu_logic_parity.py", "test_gpu_packed_state.py"],
    # never a tile-sorted slot order: texel order everywhere, the stream-ordered draw() pipeline
    "TH_BUCKET=0": ["test_gpu_logic_parity.py", "test_gpu_deposit.py", "test_gpu_view.py"],
    # the stream-ordered draw() pipeline although the integrator steps over sorted slots (every draw restores texel order)
    "TH_DRAW=stream TH_BUCKET=1 TH_RESORT_STEPS=3": ["test_gpu_deposit.py", "test_gpu_view.py", "test_gpu_scene.py"],
    # the stream-ordered view pass rasterises and sorts for itself even right after the flow pass
    "TH_DRAW=stream TH_DRAW_REUSE=0": ["test_gpu_view.py"],
    # every step through the reference-order kernel
    "TH_FORCE_GENERIC=1": ["test_gpu_logic_parity.py"],
}


@pytest.mark.parametrize("variant", sorted(VARIANTS), ids=lambda v: v.replace(" ", ","))
def test_suites_under_switch(variant):
    if os.environ.get("TH_VARIANT_RUN"):
        pytest.skip("already inside a variant run")
    env = dict(os.environ, TH_VARIANT_RUN="1")
    for kv in variant.split():
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x"] + [os.path.join(ROOT, "tests", s) for s in VARIANTS[variant]],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
