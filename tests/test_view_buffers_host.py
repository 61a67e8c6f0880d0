"""Tendrils.buffers on the host side alone (no context yet): what the constructor and setupBuffers / stepBuffers do to the
list before setupParticles() has made a device context (src/index.js:109, 172-184, 385-391)."""
import shutil
import subprocess

import pytest

from helpers import ROOT


def test_python_host_keeps_the_ring_without_a_context():
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(8, 8), dict(ta.defaults(), numBuffers=3))
    assert len(t.buffers) == 3 and all(b.shape == [1, 1] for b in t.buffers)
    a, b, c = t.buffers
    assert t.stepBuffers() is t and t.buffers == [c, a, b]          # src/utils/index.js:1-7: unshift(pop())
    t.resize()
    assert all(x.shape == [8, 8] for x in t.buffers)
    t.setupBuffers(1)
    assert t.buffers == [c] and t.stepBuffers().buffers == [c]
    assert t.viewport() is t
    t.setupBuffers()
    assert t.buffers == []
    assert ta.Tendrils(View(8, 8)).buffers == []                    # numBuffers defaults to 0 (src/index.js:68)


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_node_host_keeps_the_ring_without_a_context():
    script = """
    const T = require('./tendrils_amd/js');
    const t = new T.Tendrils({drawingBufferWidth: 8, drawingBufferHeight: 8}, {numBuffers: 3});
    const [a, b, c] = t.buffers;
    const ok = [t.buffers.length === 3, t.stepBuffers() === t, t.buffers[0] === c && t.buffers[1] === a && t.buffers[2] === b,
                t.resize() === t && t.buffers.every((x) => x.shape[0] === 8 && x.shape[1] === 8),
                t.setupBuffers(1).buffers.length === 1 && t.buffers[0] === c, t.viewport() === t,
                t.setupBuffers().buffers.length === 0, new T.Tendrils(null, {}).buffers.length === 0];
    console.log(JSON.stringify(ok));
    """
    r = subprocess.run([shutil.which("node"), "-e", script], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == "[" + ",".join(["true"] * 8) + "]", r.stdout
