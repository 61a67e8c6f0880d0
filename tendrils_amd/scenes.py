"""Headless scene replay: presets and keyframed tracks driving `tendrils.state` while the particle path
runs - what the reference's demo does interactively (src/demo.main.js:1342-1354 colour proxy, :1483-3238 presets,
:836-857 + :1027-1031 the players ticked every frame), as a batch job: step, draw, hand the view image to the caller.

A preset is DATA: {"state": {...}, "spawn": {...}, "colorProxy": {...}} (the shape of tests/golden/presets.json, taken
from the demo's preset table when the fixtures are generated; any dict of that shape works).
"""
from .animate import Player


def preset_targets(preset, state):
    """What a preset sets, as track targets: {"tendrils": {scalar state keys}, "baseColor": [r, g, b, a], ...}.
    Colours go through the demo's colour proxy rule (src/demo.main.js:1342-1354): rgb / 255, alpha apart; a preset
    that names only part of a colour keeps the current rest."""
    out = {"tendrils": dict(preset.get("state", {}))}
    proxy = preset.get("colorProxy", {})
    for name in ("base", "flow", "fade"):
        cur = list(state[name + "Color"])
        if name + "Color" in proxy:
            cur[:3] = [c / 255 for c in proxy[name + "Color"]]
        if name + "Alpha" in proxy:
            cur[3] = proxy[name + "Alpha"]
        if name + "Color" in proxy or name + "Alpha" in proxy:
            out[name + "Color"] = cur
    return out


def apply_preset(tendrils, preset):
    """Set a preset at once (what clicking it does in the demo)."""
    for key, val in preset_targets(preset, tendrils.state).items():
        if key == "tendrils":
            tendrils.state.update(val)
        else:
            tendrils.state[key][:] = val
    return tendrils


class Scene:
    """A player whose tracks write straight into `tendrils.state` (scalars through the "tendrils" track, the three colours
    through a track each), and a frame loop around it."""

    def __init__(self, tendrils):
        self.t = tendrils
        s = tendrils.state
        self.player = Player({"tendrils": [], "baseColor": [], "flowColor": [], "fadeColor": []},
                             {"tendrils": s, "baseColor": s["baseColor"], "flowColor": s["flowColor"], "fadeColor": s["fadeColor"]})

    def keyframe(self, preset, time, duration=0, ease=None):
        """Reach `preset` at `time` (ms), easing over `duration` ms before it (0: the values switch at `time`)."""
        for key, val in preset_targets(preset, self.t.state).items():
            track = self.player.tracks[key]
            frame = {"to": val, "time": time, "ease": list(ease) if ease else None}
            if duration:
                track.smooth_over(duration, frame)
            else:
                track.to(frame)
        return self

    def run(self, frames, each=None, spawner=None):
        """`frames` x (timer.tick(); player.play(time); step(); draw()); each(frame_index, tendrils) after every frame."""
        t = self.t
        if spawner is not None:
            spawner.spawn(t)
        for k in range(frames):
            t.timer.tick()
            self.player.play(t.timer.time)
            t.step()
            t.draw()
            if each is not None:
                each(k, t)
        return self
