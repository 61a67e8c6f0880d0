"""Timer semantics against the REFERENCE's own class: tests/golden/timer_script.json holds the state of
`tendrils.timer.constructor` (src/timer.js:1-80, run from the reference bundle by oracle/gen_fixtures.py) after every
operation of a script - fixed step, wall clock, pause absorbing into `offset`, end / loop, negative rates, seek / scrub
/ reset.  The Python mirror and the JS mirror of the Node host must reproduce every double exactly."""
import json
import os
import shutil
import subprocess

import pytest

from helpers import GOLDEN, ROOT

FIXTURE = json.load(open(os.path.join(GOLDEN, "timer_script.json")))
COLS = FIXTURE["columns"]


def expected():
    return [[float(v) for v in row] for row in FIXTURE["out"]]


def compare(got):
    want = expected()
    assert len(got) == len(want)
    for k, (op, g, w) in enumerate(zip(FIXTURE["ops"], got, want)):
        for name, a, b in zip(COLS, g, w):
            if name == "now" and op[0] == "tick" and op[1] is None:
                continue            # the harness probes now(null) there; the mirrors' default argument is the wall clock
            assert a == b or (a != a and b != b), "op %d %r: %s = %r, the reference has %r" % (k, op, name, a, b)


def test_python_timer_replays_the_reference_script():
    from tendrils_amd.timer import Timer
    tm, got = None, []
    for op in FIXTURE["ops"]:
        if op[0] == "new":
            tm = Timer(op[1], op[2])
        elif op[0] == "set":
            setattr(tm, op[1], op[2])
        elif op[0] == "tick":
            tm.tick(op[1])
        elif op[0] == "seek":
            tm.seek(op[1])
        elif op[0] == "scrub":
            tm.scrub(op[1])
        elif op[0] == "reset":
            tm.reset(op[1], op[2])
        probe = op[1] if (op[0] == "tick" and op[1] is not None) else 12345
        got.append([float(tm.time), float(tm.dt), float(tm.offset), float(tm.since), 1.0 if tm.paused else 0.0, float(tm.now(probe))])
    compare(got)


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_js_timer_replays_the_reference_script():
    script = """
    const {Timer} = require('./tendrils_amd/js/timer');
    const ops = JSON.parse(process.argv[1]);
    let tm = null; const out = [];
    for (const op of ops) {
      if (op[0] === 'new') tm = new Timer(op[1], op[2]);
      else if (op[0] === 'set') tm[op[1]] = op[2];
      else if (op[0] === 'tick') tm.tick(op[1] === null ? undefined : op[1]);
      else if (op[0] === 'seek') tm.seek(op[1]);
      else if (op[0] === 'scrub') tm.scrub(op[1]);
      else if (op[0] === 'reset') tm.reset(op[1], op[2]);
      const probe = (op[0] === 'tick' && op[1] !== null) ? op[1] : 12345;
      out.push([tm.time, tm.dt, tm.offset, tm.since, tm.paused ? 1 : 0, tm.now(probe)].map((v) => Object.is(v, -0) ? '-0.0' : String(v)));
    }
    console.log(JSON.stringify(out));
    """
    r = subprocess.run([shutil.which("node"), "-e", script, json.dumps(FIXTURE["ops"])], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    compare([[float(v) for v in row] for row in json.loads(r.stdout)])
