"""The library's own RCCL side (th_comm.hip) on one GPU: a world of one rank through ctypes and through the Node host -
id, collective init, the counter all-reduce on the context's stream, teardown.  (World sizes above one need more than one
GPU: bench.py --gpus N runs them; the id exchange between ranks is covered on CPU in tests/test_sharding_gloo.py.)"""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(n=128):
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(96, 54))
    t.resize()
    t.setup(n)
    rng = np.random.default_rng(3)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.02, .02, (n, n, 2))
    st[rng.random((n, n)) < 0.1] = [-1e6, -1e6, 0, 0]
    t.particles.upload_texels(st)
    return t


def test_world_of_one_through_the_c_abi():
    from tendrils_amd import _capi
    from tendrils_amd.sharding import comm_id, comm_query
    t = make()
    ctx = t.particles._ctx
    assert comm_query(ctx)["active"] is False
    local = t.particles.stats(0.01)
    # without a communicator the global block is the local one
    g = _capi.Counters()
    _capi.call("th_stats_global", ctx, C.c_float(0.01), C.byref(g))
    assert {k: getattr(g, k) for k, _ in _capi.Counters._fields_} == local
    ident = comm_id()
    buf = (C.c_ubyte * _capi.COMM_ID_BYTES).from_buffer_copy(ident)
    _capi.call("th_comm_init", ctx, buf, 0, 1)
    q = comm_query(ctx)
    assert q["active"] and q["rank"] == 0 and q["world"] == 1 and q["rccl_version"] > 0
    with pytest.raises(_capi.TendrilsHipError):           # one communicator per context
        _capi.call("th_comm_init", ctx, buf, 0, 1)
    for _ in range(3):                                     # the reduction rides the context's stream behind the pass
        t.timer.tick()
        t.step()
        _capi.call("th_stats_async", ctx, C.c_float(0.01), None)
        _capi.call("th_stats_allreduce", ctx)
    _capi.call("th_stats_global", ctx, C.c_float(0.01), C.byref(g))
    assert {k: getattr(g, k) for k, _ in _capi.Counters._fields_} == t.particles.stats(0.01)
    _capi.call("th_comm_destroy", ctx)
    assert comm_query(ctx)["active"] is False
    with pytest.raises(_capi.TendrilsHipError):
        _capi.call("th_comm_init", ctx, buf, 1, 1)          # rank outside the world
    t.dispose()


def test_world_of_one_from_the_node_host():
    node = shutil.which("node")
    if node is None or not os.path.exists(os.path.join(ROOT, "tendrils_amd", "lib", "tendrils_hip.node")):
        pytest.skip("no Node host here")
    script = """
const { Particles } = require('./tendrils_amd/js/particles');
const p = new Particles(null, { shape: [64, 64] });
p.setup(2);
p.spawn((d) => { d[0] = 0.25; d[1] = -0.5; d[2] = 0.003; d[3] = 0.004; });
const id = Particles.commUniqueId();
const before = p.commQuery();
p.commInit(id, 0, 1);
const q = p.commQuery();
const g = p.statsGlobal(0.01), l = p.stats(0.01);
p.commDestroy();
const { Tendrils } = require('./tendrils_amd/js');
const t = new Tendrils({ drawingBufferWidth: 32, drawingBufferHeight: 18 }, { rows: 16, row0: 32, globalHeight: 64 });   // a row band of a 64 x 64 texture
t.resize();
t.setup(64);
const band = { shape: t.particles.shape, live: t.particles.stats(0.01).particles };
t.dispose();
console.log(JSON.stringify({ idBytes: id.length, before, q, g, l, after: p.commQuery(), band }));
p.dispose();
"""
    r = subprocess.run([node, "-e", script], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["idBytes"] == 128 and out["before"]["active"] == 0 and out["after"]["active"] == 0
    assert out["q"]["active"] == 1 and out["q"]["world"] == 1 and out["q"]["rcclVersion"] > 0
    assert out["g"] == out["l"] and out["g"]["live"] == 64 * 64 and abs(out["g"]["sumSpeed"] - 64 * 64 * 0.005) < 1e-3
    assert out["band"] == {"shape": [64, 16], "live": 64 * 16}


def test_state_gather_over_the_librarys_communicator():
    """th_state_gather (the whole particle texture on every rank, for the spawners that sample arbitrary particles) over a
    world of one: the copy equals the band, and a particle-texture best-sample pass on the 'shard' equals the plain one."""
    import torch
    from tendrils_amd import _capi, sharding
    from tendrils_amd.sharding import comm_id
    from tendrils_amd.spawn import PixelSpawner, data_sample_frag
    from helpers import bits_equal
    outs = []
    for gathered in (False, True):
        t = make(96)
        ctx = t.particles._ctx
        if gathered:
            buf = (C.c_ubyte * _capi.COMM_ID_BYTES).from_buffer_copy(comm_id())
            _capi.call("th_comm_init", ctx, buf, 0, 1)
            sharding.gather_state(t, 0)
            ptr = C.c_void_p()
            _capi.call("th_state_gather_ptr", ctx, 0, C.byref(ptr))      # (the same buffer: what the gather left there)
            copy = sharding.device_view(ptr.value, (96, 96, 4), "<f4").cpu().numpy()
            assert bits_equal(copy, t.particles.read(0)).all()
        sp = PixelSpawner(None, dict(shader=data_sample_frag(), buffer=t.particles.buffers[0], spawnSize=[0.8, 0.8], speed=0.01, bias=0.3))
        t.timer.time = 480.0
        sp.spawn(t)
        outs.append(t.particles.read(0))
        t.dispose()
    assert bits_equal(outs[0], outs[1]).all()


def test_draw_sharded_by_the_library_world_of_one(oracle):
    """th_draw_sharded - edge rows, emit, fragment all-to-all, merge, all-gather, all issued by the library on its own
    communicator (ncclSend / ncclRecv groups) - at world size 1 (one GPU): both passes against the local th_draw."""
    from tendrils_amd import _capi, sharding
    from tendrils_amd.sharding import comm_id
    from helpers import bits_equal
    outs = []
    for native in (False, True):
        t = make(96)
        for _ in range(3):
            t.timer.tick()
            t.step()
        if native:
            buf = (C.c_ubyte * _capi.COMM_ID_BYTES).from_buffer_copy(comm_id())
            _capi.call("th_comm_init", t.particles._ctx, buf, 0, 1)
            t.state["autoClearView"] = False
            t.drawFade()
            frags = sharding.draw_sharded_native(t, view=True)
        else:
            t.draw()
            frags = t.fragments
        outs.append((frags, t.flow.read(), t.read_view()))
        t.dispose()
    assert outs[0][0] == outs[1][0] > 100
    assert bits_equal(outs[0][1], outs[1][1]).all()
    assert (outs[0][2] == outs[1][2]).all() and outs[0][2].any()
