"""Scene replay (tendrils_amd/scenes.py): presets from the fixture table keyframed into `tendrils.state` by the Player
while step() + draw() run; the state must follow the tracks and the view must fill."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def test_two_presets_eased_over_a_short_run():
    import tendrils_amd as ta
    from tendrils_amd.scenes import Scene, apply_preset, preset_targets
    from tendrils_amd.spawn.ball import spawnBall
    from tendrils_amd.tendrils import View
    table = json.load(open(os.path.join(GOLDEN, "presets.json")))
    assert len(table) >= 30 and "Flow" in table and "Wings" in table
    t = ta.Tendrils(View(96, 54))
    t.resize()
    t.setup(64)
    apply_preset(t, table["Flow"])
    assert t.state["colorMapAlpha"] == 0 and t.state["flowColor"][:3] == [1.0, 1.0, 1.0] and t.state["baseColor"][3] == 0
    scene = Scene(t)
    end = 24 * t.timer.step
    scene.keyframe(table["Wings"], time=end, duration=0.75 * end, ease=[0, 0.95, 1])
    seen = []
    scene.run(24, each=lambda k, tn: seen.append((tn.state["flowDecay"], tn.state["baseColor"][3], tn.view_fragments)),
              spawner=spawnBall(None, dict(uniforms=dict(radius=0.25, speed=0.01))))
    view = t.read_view()
    stats = t.particles.stats(t.state["speedLimit"])
    t.dispose()
    want = preset_targets(table["Wings"], t.state)
    assert t.state["flowDecay"] == want["tendrils"]["flowDecay"] == 0          # the last frame lands on the keyframe
    assert t.state["baseColor"] == want["baseColor"] and t.state["baseColor"][3] == 0.8
    decay = [s[0] for s in seen]
    assert decay[0] == 0.005 and all(a >= b for a, b in zip(decay, decay[1:])) and decay[-1] == 0      # eased, monotonic here
    assert 0 < seen[10][1] < 0.8                                                # the colour track moves too
    assert stats["live"] == 64 * 64 and seen[-1][2] > 0 and view.any()
