import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure) - builds oracle/libtendrils_oracle.so if needed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    return O


# ---- the library's alternative paths ----------------------------------------------------------------------
# The switches are per-context options that start from the environment when a context is created (th_create), so a
# variant is just an environment around one test: the suites listed here run once by default and once more under every
# variant that names them - in this process, through the same hosts (the Node host's child process inherits the
# environment).  No variant changes a result: the assertions are the suites' own.
VARIANTS = {
    # no temporal fusion: th_step_n replays a captured graph of single-step launches
    "no-fuse": ({"TH_FUSE": "0"}, ["test_gpu_logic_parity", "test_gpu_packed_state"]),
    # never a tile-sorted slot order: texel order everywhere, the stream-ordered draw() pipeline
    "no-bucket": ({"TH_BUCKET": "0"}, ["test_gpu_logic_parity", "test_gpu_deposit", "test_gpu_view"]),
    # tile-sorted slots forced on at every size, re-sorted every 2 fused steps
    "bucket": ({"TH_BUCKET": "1", "TH_REBUCKET_STEPS": "2"},
               ["test_gpu_logic_parity", "test_gpu_optical_flow", "test_gpu_spawn", "test_gpu_deposit", "test_gpu_fuzz"]),
    # the stream-ordered draw() pipeline although the integrator steps over sorted slots (every draw restores texel order)
    "stream-on-sorted": ({"TH_DRAW": "stream", "TH_BUCKET": "1", "TH_RESORT_STEPS": "3"},
                         ["test_gpu_deposit", "test_gpu_view", "test_gpu_scene"]),
    # the stream-ordered view pass rasterises and sorts for itself even right after the flow pass
    "no-draw-reuse": ({"TH_DRAW": "stream", "TH_DRAW_REUSE": "0"}, ["test_gpu_view"]),
    # every step through the reference-order kernel
    "generic": ({"TH_FORCE_GENERIC": "1"}, ["test_gpu_logic_parity"]),
    # the binned draw() pipeline forced, in texel order ...
    "bins": ({"TH_DRAW": "bins"}, ["test_gpu_deposit", "test_gpu_view", "test_gpu_fuzz", "test_gpu_scene"]),
    # ... and over tile-sorted slots re-sorted every few steps
    "bins-on-sorted": ({"TH_DRAW": "bins", "TH_BUCKET": "1", "TH_RESORT_STEPS": "3", "TH_REBUCKET_STEPS": "2"},
                       ["test_gpu_deposit", "test_gpu_view", "test_gpu_fuzz", "test_gpu_scene"]),
    # ... with the frame loop's re-sort inside its steps (a COUNT pass, a SCATTER pass) instead of beside its draws
    "bins-on-sorted-in-steps": ({"TH_DRAW": "bins", "TH_BUCKET": "1", "TH_RESORT_STEPS": "3", "TH_REBUCKET_STEPS": "2", "TH_ASYNC_SORT": "0"},
                                ["test_gpu_deposit", "test_gpu_scene"]),
}


def pytest_generate_tests(metafunc):
    if "th_variant" not in metafunc.fixturenames:
        return
    module = metafunc.module.__name__.rsplit(".", 1)[-1]
    names = [None] + [k for k, (_, suites) in VARIANTS.items() if module in suites]
    if len(names) > 1:
        metafunc.parametrize("th_variant", names, indirect=True, ids=[n or "default" for n in names])


@pytest.fixture(autouse=True)
def th_variant(request, monkeypatch):
    name = getattr(request, "param", None)
    if name:
        for k, v in VARIANTS[name][0].items():
            monkeypatch.setenv(k, v)
    return name
