"""Particles.spawn(map, pixels, offset) (src/particles.js:94-117) against captures of the REFERENCE: a position-dependent
map on the default staging array and on a non-square one at an offset.  The [w, h, 4] array is filled x-outer / y-inner
and handed to setPixels; which texel ends up with map(x, y) - and that every ring buffer gets it - is what is pinned."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, bits_equal

pytestmark = pytest.mark.gpu
CASES = ["mapspawn_full_16", "mapspawn_rect_24"]


def fixture(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return d["buffers"], [float(v) for v in d["coef"]], [int(v) for v in d["pixels"]], [int(v) for v in d["offset"]]


@pytest.mark.parametrize("name", CASES)
def test_python_spawn_map_matches_the_reference(name):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    want, c, pixels, offset = fixture(name)
    n = want.shape[1]
    t = ta.Tendrils(View(32, 32))
    t.resize()
    t.setup(n)

    def fn(data, x, y):
        data[0] = c[0] + c[1] * x + c[2] * y
        data[1] = c[3] + c[4] * x + c[5] * y
        data[2] = x
        data[3] = y
    if name.endswith("full_16"):
        t.particles.spawn(fn)
    else:
        t.particles.spawn(fn, np.zeros((pixels[0], pixels[1], 4), np.float32), offset)
    got = [t.particles.read(k) for k in range(len(t.particles.buffers))]
    t.dispose()
    assert len(got) == want.shape[0]
    for k in range(len(got)):
        assert bits_equal(got[k], want[k]).all(), "ring buffer %d" % k
    # (the rectangle case: everything outside it is still the inert fill of setup())
    assert (want[0][..., 0] == -1e6).sum() == n * n - pixels[0] * pixels[1]


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
@pytest.mark.parametrize("name", CASES)
def test_node_spawn_map_matches_the_reference(name):
    want, c, pixels, offset = fixture(name)
    n = want.shape[1]
    script = """
    const T = require('./tendrils_amd/js');
    const cfg = JSON.parse(process.argv[1]);
    const t = new T.Tendrils({drawingBufferWidth: 32, drawingBufferHeight: 32}, {});
    t.resize(); t.setup(cfg.n);
    const c = cfg.c;
    const fn = (data, x, y) => { data[0] = c[0] + c[1]*x + c[2]*y; data[1] = c[3] + c[4]*x + c[5]*y; data[2] = x; data[3] = y; };
    if (cfg.full) t.particles.spawn(fn);
    else t.particles.spawn(fn, {shape: [cfg.pixels[0], cfg.pixels[1], 4], data: new Float32Array(cfg.pixels[0]*cfg.pixels[1]*4)}, cfg.offset);
    const out = [];
    for (let k = 0; k < t.particles.buffers.length; ++k) out.push(Buffer.from(t.particles.read(k).buffer).toString('base64'));
    t.dispose();
    console.log(JSON.stringify(out));
    """
    cfg = dict(n=n, c=c, pixels=pixels, offset=offset, full=name.endswith("full_16"))
    r = subprocess.run([shutil.which("node"), "-e", script, json.dumps(cfg)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import base64
    got = [np.frombuffer(base64.b64decode(b), np.float32).reshape(n, n, 4) for b in json.loads(r.stdout)]
    assert len(got) == want.shape[0]
    for k in range(len(got)):
        assert bits_equal(got[k], want[k]).all(), "ring buffer %d" % k
