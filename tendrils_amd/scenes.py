"""Headless scene replay: presets and keyframed tracks driving `tendrils.state` while the particle path
runs - what the reference's demo does interactively (src/demo.main.js:1335-1354 colour proxy, :1483-3238 presets,
:836-857 + :1027-1031 the players ticked every frame), as a batch job: step, draw, hand the frame to the caller.

A preset is DATA: {"state": {...}, "spawn": {...}, "colorProxy": {...}} (the shape of tests/golden/presets.json, taken
from the demo's preset table when the fixtures are generated; any dict of that shape works).  Pinned by
tests/golden/scene_flow_turbulence_wings_64.npz: the reference's own Player driving the reference's own Tendrils through
the same script, frame by frame (tests/test_scene_script.py on CPU, tests/test_gpu_scene.py on the GPU).
"""
from .animate import Player

COLOURS = ("base", "flow", "fade")


class ColourProxy:
    """The demo's colour proxy: colours as 0..255 rgb + an alpha apart; a preset assigns some of the six entries, the rest
    stays as the last preset left it, and a state colour is [r / 255, g / 255, b / 255, alpha]."""

    def __init__(self, state):
        self.rgb = {c: [v * 255 for v in state[c + "Color"][:3]] for c in COLOURS}
        self.alpha = {c: state[c + "Color"][3] for c in COLOURS}

    def take(self, preset):
        """Assign what `preset` names; returns the state colours it touched, {"baseColor": [r, g, b, a], ...}."""
        named = preset.get("colorProxy", {})
        touched = {}
        for c in COLOURS:
            if c + "Color" in named:
                self.rgb[c] = list(named[c + "Color"])
            if c + "Alpha" in named:
                self.alpha[c] = named[c + "Alpha"]
            if c + "Color" in named or c + "Alpha" in named:
                touched[c + "Color"] = self.colour(c)
        return touched

    def colour(self, c):
        return [v / 255 for v in self.rgb[c]] + [self.alpha[c]]


def apply_preset(tendrils, preset, proxy=None):
    """Set a preset at once (what clicking it does in the demo): its state entries, and all three colours from the proxy."""
    proxy = proxy or ColourProxy(tendrils.state)
    proxy.take(preset)
    tendrils.state.update(preset.get("state", {}))
    for c in COLOURS:
        tendrils.state[c + "Color"][:] = proxy.colour(c)
    return tendrils


class Scene:
    """A player whose tracks write straight into `tendrils.state` (scalars through the "tendrils" track, the three colours
    through a track each - the demo's track table), a colour proxy that runs through the presets in the order they are
    given, and the frame loop around both."""

    def __init__(self, tendrils):
        self.t = tendrils
        s = tendrils.state
        self.proxy = ColourProxy(s)
        self.player = Player({"tendrils": [], "baseColor": [], "flowColor": [], "fadeColor": []},
                             {"tendrils": s, "baseColor": s["baseColor"], "flowColor": s["flowColor"], "fadeColor": s["fadeColor"]})

    def preset(self, preset):
        apply_preset(self.t, preset, self.proxy)
        return self

    def keyframe(self, preset, time, duration=0, ease=None):
        """Reach `preset` at `time` (ms): eased over the `duration` ms before it, or - duration 0 - from the key before.
        Either way the curve is joined to the one that leads into the previous key (smoothOver / smoothTo)."""
        targets = {"tendrils": dict(preset.get("state", {})), **self.proxy.take(preset)}
        for name, to in targets.items():
            frame = {"to": to, "time": time, "ease": list(ease) if ease else None}
            track = self.player.tracks[name]
            if duration:
                track.smooth_over(duration, frame)
            else:
                track.smooth_to(frame)
        return self

    def frame(self):
        """One pass of the demo's loop body: timer.tick(); player.play(time); step(); draw()."""
        t = self.t
        t.timer.tick()
        self.player.play(t.timer.time)
        t.step()
        t.draw()
        return self

    def run(self, frames, each=None, spawner=None):
        if spawner is not None:
            spawner.spawn(self.t)
        for k in range(frames):
            self.frame()
            if each is not None:
                each(k, self.t)
        return self
