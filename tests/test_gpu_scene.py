"""Scene replay on the GPU against frames captured from the REFERENCE: its own Player driving its own
Tendrils - preset "Flow", then keyframes easing into "Turbulence" and "Wings"; preset "Fluid" (the view wiped every frame)
into "Ghostly" and "Rorschach"; "Flow" into "Funhouse" and "Rave" (`target` and `varyTarget` eased up from 0 over a targets
texture: the integrator's TARGET kernels under a moving uniform) - over 24 frames of tick / play / step / draw (oracle/gen_fixtures.py:gen_scene ->
tests/golden/scene_*.npz).  The same script runs through
tendrils_amd/scenes.py and through the Node host's js/scenes.js: the state object must follow the reference's double for
double, the particle texture, the flow field and the view image within tests/test_scene_script.py:scene_close."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT
from test_scene_script import SCENES, TABLE, check_states, scene_close

pytestmark = pytest.mark.gpu


@pytest.fixture(params=SCENES, ids=lambda c: c.name)
def captured(request):
    return request.param


def test_python_scene_against_the_reference_frames(captured):
    import copy
    FX, META = captured.FX, captured.META
    import tendrils_amd as ta
    from tendrils_amd.scenes import Scene
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(*META["viewRes"]))
    t.resize()
    t.setup(META["N"])
    assert list(t.viewSize) == META["viewSize"]
    scene = Scene(t).preset(TABLE[META["first"]])
    for k in META["script"]:
        scene.keyframe(TABLE[k["preset"]], k["time"], k["duration"], k["ease"])
    t.particles.upload_texels(FX["state"])
    if "targets" in FX.files:                      # (the scene with `target` > 0: tendrils.targets, src/index.js:105,207)
        t.targets.set_pixels(FX["targets"])
    t.timer.time = META["time0"]
    hosts, states, flows, views = [], [], [], []

    def each(k, tn):
        assert tn.timer.time == META["times"][k]
        hosts.append(copy.deepcopy(tn.state))
        states.append(tn.particles.read(0))
        if k in META["grab"]:
            flows.append(tn.flow.read())
            views.append(tn.read_view())
    scene.run(META["frames"], each=each)
    t.dispose()
    check_states(hosts, META)
    scene_close(states, flows, views, captured)


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_node_scene_against_the_reference_frames(tmp_path, captured):
    FX, META = captured.FX, captured.META
    FX["state"].astype(np.float32).tofile(tmp_path / "state.bin")
    inputs = {"state": "state.bin"}
    if "targets" in FX.files:
        FX["targets"].astype(np.float32).tofile(tmp_path / "targets.bin")
        inputs["targets"] = "targets.bin"
    spec = dict(kind="scene", N=META["N"], viewRes=META["viewRes"], inputs=inputs, time0=META["time0"],
                frames=META["frames"], grab=META["grab"], first=META["first"], script=META["script"], table=TABLE)
    (tmp_path / "case.json").write_text(json.dumps(spec))
    r = subprocess.run([shutil.which("node"), os.path.join(ROOT, "tests", "js", "run_case.js"), str(tmp_path / "case.json")],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = json.loads((tmp_path / "result.json").read_text())
    assert res["times"] == META["times"]
    check_states(res["states"], META)
    n, (fw, fh) = META["N"], META["viewRes"]
    states = [np.fromfile(tmp_path / ("state_%d.bin" % k), np.float32).reshape(n, n, 4) for k in range(META["frames"])]
    flows = [np.fromfile(tmp_path / ("flow_%d.bin" % k), np.float32).reshape(fh, fw, 4) for k in META["grab"]]
    views = [np.fromfile(tmp_path / ("view_%d.bin" % k), np.uint8).reshape(fh, fw, 4) for k in META["grab"]]
    scene_close(states, flows, views, captured)


def test_replay_tool_runs(tmp_path):
    """tools/replay_scene.py: the batch renderer over the preset table writes its frames."""
    out = str(tmp_path / "scene")
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "replay_scene.py"), os.path.join(GOLDEN, "presets.json"), "Flow",
                        "Turbulence", "Wings", "--frames", "24", "--every", "12", "--root", "64", "--view", "96x54", "--out", out],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(out + "_0012.ppm") == os.path.getsize(out + "_0024.ppm") == len(b"P6 96 54 255\n") + 96 * 54 * 3
