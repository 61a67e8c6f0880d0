"""Player / Timeline / tween against the REFERENCE's own compiled classes (src/animate, taken out of the demo bundle's
module table by oracle/harness_animate.js): tests/golden/animate_script.json holds a scripted set of tracks - start
frames, eased keyframes (smoothTo / flipTo / smoothOver / easeOver / flipOver with curves of 2-5 control values), array
and object outputs, call labels - and, after every play / seek / playFrom of the player, all outputs, every track's
playhead and the calls made so far.  The Python mirror and the Node mirror must reproduce every double exactly."""
import copy
import json
import os
import shutil
import subprocess

import pytest

from helpers import GOLDEN, ROOT

FIX = json.load(open(os.path.join(GOLDEN, "animate_script.json")))


def same(a, b, path=""):
    if isinstance(b, dict):
        assert isinstance(a, dict), path
        ka = {k for k, v in a.items() if v is not None}
        kb = {k for k, v in b.items() if v is not None}
        assert ka == kb, "%s: keys %s vs %s" % (path, sorted(ka), sorted(kb))
        for k in kb:
            same(a[k], b[k], path + "." + str(k))
    elif isinstance(b, list):
        assert isinstance(a, list) and len(a) == len(b), path
        for k, (x, y) in enumerate(zip(a, b)):
            same(x, y, "%s[%d]" % (path, k))
    else:
        assert a == b and type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float)) and float(a) == float(b) and not isinstance(a, bool) and not isinstance(b, bool)), \
            "%s: %r, the reference has %r" % (path, a, b)


def test_python_player_replays_the_reference_script():
    from tendrils_amd.animate import Player
    log = []

    def fix(frame):
        if isinstance(frame, dict) and frame.get("call"):
            frame["call"] = [(lambda out, span, label=label: log.append(label)) for label in frame["call"]]
        return frame
    tracks = {k: [fix(copy.deepcopy(f)) for f in v] for k, v in FIX["tracks"].items()}
    player = Player(tracks, copy.deepcopy(FIX["outputs"]))
    names = dict(to="to", smoothTo="smooth_to", flipTo="flip_to", over="over", easeOver="ease_over", smoothOver="smooth_over",
                 flipOver="flip_over", easeTo="ease_to")
    got = []
    for op in copy.deepcopy(FIX["ops"]):
        if op[0] == "track":
            getattr(player.tracks[op[1]], names[op[2]])(*[fix(a) if isinstance(a, dict) else a for a in op[3:]])
            continue
        if op[0] == "play":
            player.play(op[1])
        elif op[0] == "seek":
            player.seek(op[1])
        else:
            player.play_from(op[1], op[2])
        got.append({"outputs": copy.deepcopy(player.outputs),
                    "heads": {k: [t.time, t.gap, len(t.frames)] for k, t in player.tracks.items()}, "calls": list(log)})
    assert len(got) == len(FIX["expected"])
    for k, (g, w) in enumerate(zip(got, FIX["expected"])):
        same(g, w, "call %d" % k)
    for name, frames in FIX["frames"].items():          # the timelines themselves: times and joined ease curves
        mine = player.tracks[name].frames
        assert len(mine) == len(frames)
        for f, w in zip(mine, frames):
            t = f["time"]
            assert (t == float("inf") and w["time"] == "inf") or (t == float("-inf") and w["time"] == "-inf") or t == w["time"]
            assert (f.get("ease") or None) == w["ease"]


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_js_player_replays_the_reference_script():
    script = """
    const {Player} = require('./tendrils_amd/js/animate');
    const fx = JSON.parse(require('fs').readFileSync(process.argv[1]));
    const log = [];
    const fix = (frame) => { if (frame && frame.call) frame.call = frame.call.map((label) => () => log.push(label)); return frame; };
    const tracks = {}; for (const k in fx.tracks) tracks[k] = fx.tracks[k].map(fix);
    const player = new Player(tracks, JSON.parse(JSON.stringify(fx.outputs)));
    const out = [];
    for (const op of fx.ops) {
      if (op[0] === 'track') { const tl = player.tracks[op[1]]; tl[op[2]](...op.slice(3).map((a) => (a && typeof a === 'object' && !Array.isArray(a)) ? fix(a) : a)); continue; }
      if (op[0] === 'play') player.play(op[1]); else if (op[0] === 'seek') player.seek(op[1]); else player.playFrom(op[1], op[2]);
      const heads = {}; for (const k in player.tracks) heads[k] = [player.tracks[k].time, player.tracks[k].gap, player.tracks[k].frames.length];
      out.push({outputs: JSON.parse(JSON.stringify(player.outputs)), heads, calls: log.slice()});
    }
    console.log(JSON.stringify(out));
    """
    r = subprocess.run([shutil.which("node"), "-e", script, os.path.join(GOLDEN, "animate_script.json")], cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert len(got) == len(FIX["expected"])
    for k, (g, w) in enumerate(zip(got, FIX["expected"])):
        same(g, w, "call %d" % k)
