"""Player / Timeline / tween against the REFERENCE's own compiled classes (src/animate, taken out of the demo bundle's
module table by oracle/harness_animate.js): tests/golden/animate_script.json holds a scripted set of tracks - start
frames, eased keyframes (smoothTo / flipTo / smoothOver / easeOver / flipOver / easeTo with curves of 2-6 control
values), array and object outputs, call labels, keys added and taken out WHILE the playhead stands inside the timeline
(the demo's `keyframe()` and <backspace>, src/demo.main.js:1267-1274, :3472-3474), the return values of the timeline's
queries (gapAt, indexOf, spanGapAt, minFrame, splice*, start / end / duration, valid) - and, after every play / seek /
playFrom of the player, all outputs, every track's playhead and the calls made so far.  The Python and the Node
implementation must reproduce every double exactly."""
import copy
import json
import math
import os
import shutil
import subprocess

import pytest

from helpers import GOLDEN, ROOT

FIX = json.load(open(os.path.join(GOLDEN, "animate_script.json")))
SNAKE = dict(to="to", smoothTo="smooth_to", flipTo="flip_to", over="over", easeOver="ease_over", smoothOver="smooth_over",
             flipOver="flip_over", easeTo="ease_to", spliceAt="splice_at", spliceSpan="splice_span", spliceIndex="splice_index",
             splice="splice", gapAt="gap_at", indexOf="index_of", spanGapAt="span_gap_at", minFrame="min_frame", start="start",
             end="end", duration="duration", valid="valid")


def same(a, b, path=""):
    """a == b down to the last double; None entries of dicts count as absent (JSON drops `undefined`)."""
    if isinstance(b, dict):
        assert isinstance(a, dict), path
        ka = {str(k) for k, v in a.items() if v is not None}
        kb = {k for k, v in b.items() if v is not None}
        assert ka == kb, "%s: keys %s vs %s" % (path, sorted(ka), sorted(kb))
        for k in kb:
            same(a[k] if k in a else a[int(k)], b[k], path + "." + str(k))
    elif isinstance(b, list):
        assert isinstance(a, list) and len(a) == len(b), path
        for k, (x, y) in enumerate(zip(a, b)):
            same(x, y, "%s[%d]" % (path, k))
    else:
        assert a == b and type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float)) and float(a) == float(b) and not isinstance(a, bool) and not isinstance(b, bool)), \
            "%s: %r, the reference has %r" % (path, a, b)


def spelled(v):
    """What JSON.stringify leaves of a value (functions and undefined drop out, infinities spelled as the fixture does)."""
    if isinstance(v, dict):
        return {str(k): spelled(x) for k, x in v.items() if x is not None and not callable(x)}
    if isinstance(v, (list, tuple)):
        return [None if callable(x) else spelled(x) for x in v]
    if isinstance(v, float) and math.isinf(v):
        return "inf" if v > 0 else "-inf"
    return v


def test_python_player_replays_the_reference_script():
    from tendrils_amd.animate import Player
    log = []

    def fix(frame):
        if isinstance(frame, dict) and frame.get("call"):
            frame["call"] = [(lambda out, span, label=label: log.append(label)) for label in frame["call"]]
        return frame
    tracks = {k: [fix(copy.deepcopy(f)) for f in v] for k, v in FIX["tracks"].items()}
    assert list(tracks) == sorted(tracks)                   # the order the reference's page saw (see oracle/gen_fixtures.py)
    player = Player(tracks, copy.deepcopy(FIX["outputs"]))
    got, queries = [], []
    for op in copy.deepcopy(FIX["ops"]):
        if op[0] in ("track", "query"):
            val = getattr(player.tracks[op[1]], SNAKE[op[2]])(*[fix(a) if isinstance(a, dict) else a for a in op[3:]])
            if op[0] == "query":
                queries.append(spelled(val))
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
    assert len(queries) == len(FIX["queries"]) > 15
    for k, (g, w) in enumerate(zip(queries, FIX["queries"])):
        if w is None:
            assert not g, "query %d" % k
        else:
            same(g, w, "query %d" % k)
    for name, frames in FIX["frames"].items():          # the timelines themselves: times and joined ease curves
        mine = player.tracks[name].frames
        assert len(mine) == len(frames)
        for f, w in zip(mine, frames):
            assert spelled(f["time"]) == w["time"]
            assert (f.get("ease") or None) == w["ease"]
            assert (f.get("to") if f.get("to") is not None else None) == w["to"]
    assert {k: spelled(getattr(player, k)()) for k in ("start", "end", "duration")} == FIX["player"]


JS_REPLAY = """
// replays one script (an object with ops / outputs and, optionally, tracks) or a list of them through js/animate.js
const {Player} = require('./tendrils_amd/js/animate');
const spell = (key, v) => (v === Infinity ? 'inf' : (v === -Infinity ? '-inf' : (typeof v === 'number' && v !== v ? 'nan' : v)));
function replay(fx) {
  const log = [];
  const fix = (frame) => { if (frame && frame.call) frame.call = frame.call.map((label) => () => log.push(label)); return frame; };
  const tracks = {};
  for (const k of Object.keys(fx.tracks || fx.outputs).sort()) tracks[k] = ((fx.tracks || {})[k] || []).map(fix);
  const player = new Player(tracks, JSON.parse(JSON.stringify(fx.outputs)));
  const out = [], queries = [];
  for (const op of fx.ops) {
    if (op[0] === 'track' || op[0] === 'query') {
      const tl = player.tracks[op[1]];
      const val = tl[op[2]](...op.slice(3).map((a) => (a && typeof a === 'object' && !Array.isArray(a)) ? fix(a) : a));
      if (op[0] === 'query') queries.push(JSON.parse(JSON.stringify(val === undefined ? null : val, spell)));
      continue;
    }
    if (op[0] === 'play') player.play(op[1]); else if (op[0] === 'seek') player.seek(op[1]); else player.playFrom(op[1], op[2]);
    const heads = {}; for (const k in player.tracks) heads[k] = [player.tracks[k].time, player.tracks[k].gap, player.tracks[k].frames.length];
    out.push({outputs: JSON.parse(JSON.stringify(player.outputs)), heads, calls: log.slice()});
  }
  const frames = {};
  for (const k in player.tracks) frames[k] = player.tracks[k].frames.map((f) => ({time: spell('', f.time), ease: f.ease || null, to: (f.to === undefined ? null : f.to)}));
  return {out, queries, frames, player: {start: spell('', player.start()), end: spell('', player.end()), duration: spell('', player.duration())}};
}
const given = JSON.parse(require('fs').readFileSync(process.argv[1]));
console.log(JSON.stringify(Array.isArray(given) ? given.map(replay) : replay(given)));
"""


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_js_player_replays_the_reference_script():
    r = subprocess.run([shutil.which("node"), "-e", JS_REPLAY, os.path.join(GOLDEN, "animate_script.json")], cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert len(got["out"]) == len(FIX["expected"])
    for k, (g, w) in enumerate(zip(got["out"], FIX["expected"])):
        same(g, w, "call %d" % k)
    assert got["queries"] == FIX["queries"] and got["frames"] == FIX["frames"] and got["player"] == FIX["player"]


FUZZ = json.load(open(os.path.join(GOLDEN, "animate_fuzz.json")))


def replay_python(case):
    from tendrils_amd.animate import Player
    log = []

    def fix(frame):
        if isinstance(frame, dict) and frame.get("call"):
            frame["call"] = [(lambda out, span, label=label: log.append(label)) for label in frame["call"]]
        return frame
    player = Player({k: [] for k in sorted(case["outputs"])}, copy.deepcopy(case["outputs"]))
    got, queries = [], []
    for op in copy.deepcopy(case["ops"]):
        if op[0] in ("track", "query"):
            val = getattr(player.tracks[op[1]], SNAKE[op[2]])(*[fix(a) if isinstance(a, dict) else a for a in op[3:]])
            if op[0] == "query":
                queries.append(spelled(val))
            continue
        if op[0] == "play":
            player.play(op[1])
        elif op[0] == "seek":
            player.seek(op[1])
        else:
            player.play_from(op[1], op[2])
        got.append({"outputs": copy.deepcopy(player.outputs),
                    "heads": {k: [t.time, t.gap, len(t.frames)] for k, t in player.tracks.items()}, "calls": list(log)})
    frames = {k: [{"time": spelled(f["time"]), "ease": f.get("ease") or None, "to": f.get("to")} for f in t.frames] for k, t in player.tracks.items()}
    return got, queries, frames


@pytest.mark.parametrize("case", FUZZ, ids=lambda c: "seed%d" % c["seed"])
def test_python_player_replays_random_reference_scripts(case):
    """Twelve seeded random scripts run on the reference's own Player (oracle/gen_fixtures.py:random_animate_script): keyframes of
    every kind added before and during playback, removals, plays forwards and backwards, seeks, playFroms - every output, every
    playhead, every call, every query result and the final timelines, double for double."""
    got, queries, frames = replay_python(case)
    assert len(got) == len(case["expected"]) > 5
    for k, (g, w) in enumerate(zip(got, case["expected"])):
        same(g, w, "call %d" % k)
    for k, (g, w) in enumerate(zip(queries, case["queries"])):
        same(g, w, "query %d" % k)
    for name, want in case["frames"].items():
        mine = frames[name]
        assert len(mine) == len(want)
        for f, w in zip(mine, want):
            assert f["time"] == w["time"] and f["ease"] == w["ease"] and f["to"] == w["to"], (name, f, w)


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_js_player_replays_random_reference_scripts():
    r = subprocess.run([shutil.which("node"), "-e", JS_REPLAY, os.path.join(GOLDEN, "animate_fuzz.json")], cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert len(got) == len(FUZZ)
    for case, g in zip(FUZZ, got):
        assert len(g["out"]) == len(case["expected"])
        for k, (a, w) in enumerate(zip(g["out"], case["expected"])):
            same(a, w, "seed %d call %d" % (case["seed"], k))
        assert g["queries"] == case["queries"] and g["frames"] == case["frames"], case["seed"]


def test_tween_forms_and_curves():
    """The blend's closed forms: curve values by de Casteljau equal the Bernstein sum, the two-point curve is a lerp, and
    the collection form blends numbers only (anything else in a key lands when the playhead passes the key, through
    `apply`) and takes a number it has nothing to blend from at t = 1."""
    from tendrils_amd.animate import curve_at, join_curve, tween, tween_props, tween_value
    for pts in ([0.3], [0, 1], [0, 0.95, 1], [0, -0.3, 1.2, 1], [0, 0.1, 0.4, 0.9, 1], [0, 0, 0.3, 0.6, 0.8, 0.9, 1]):
        n = len(pts) - 1
        for u in (0, 0.125, 0.5, 0.99, 1):
            want = sum(math.comb(n, i) * (1 - u) ** (n - i) * u ** i * p for i, p in enumerate(pts))
            assert abs(curve_at(pts, u) - want) < 1e-15 * max(1, n * n)
    assert join_curve(None) == 0 and join_curve([0.25]) == 0.25 and join_curve([0, 0.25, 1], -1) == -0.75
    assert tween_value(2, 4, 0.25) == 2.5 and tween_value(2, 4, 0.5, [0, 1, 1]) == 2 + 2 * 0.75 and tween_value("x", 4, 0.1) == 4
    out = {"k": 1, "flag": True, "n": 5}
    tween_props({"k": 3}, {"k": 5, "flag": False, "n": 7, "new": 2}, 0.5, None, out)
    assert out == {"k": 4, "flag": True, "n": 6, "new": None}            # `new`: nothing to come from before t = 1
    tween_props({"k": 3}, {"k": 5, "flag": False, "new": 2}, 1, None, out)
    assert out == {"k": 5, "flag": True, "n": 6, "new": 2}
    assert tween({"a": 1, "b": 2, "t": 0.5, "ease": None}) == 1.5 and tween({"a": [0, 0], "b": [1, 2], "t": 0.5}, [9, 9]) == [0.5, 1]
