"""The Node host (tendrils_amd/js + the N-API shim): CPU-side checks that it loads and mirrors the
reference's host logic; GPU-side parity through the JS API against the golden vectors."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import ROOT, bits_equal, golden, load, of_expected, of_inputs, state_overrides

NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")
ADDON = os.path.join(ROOT, "tendrils_amd", "lib", "tendrils_hip.node")


def node(script, *args, cwd=ROOT):
    return subprocess.run([NODE, "-e", script, *args], cwd=cwd, capture_output=True, text=True, timeout=300)


def test_addon_loads_and_exports_the_abi():
    if not os.path.exists(ADDON):
        import __graft_entry__ as g
        g.build()
    r = node("const a=require('./tendrils_amd/lib/tendrils_hip.node');"
             "console.log(JSON.stringify({n:Object.keys(a).length,abi:a.abiVersion(),ring:a.TARGET_RING}))")
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    assert info == {"n": 97, "abi": 14, "ring": -1}


@pytest.mark.gpu
def test_options_and_communicator_ids_through_the_shim():
    """Particles.option(name[, value]) reads / sets a per-context switch; commLoopbackId() makes the id of an in-process world
    (th_comm_init tells it from an RCCL id by itself: a context joins it as rank 0 of 1)."""
    r = node("""
    const T = require('./tendrils_amd/js');
    const native = require('./tendrils_amd/js/native');
    const t = new T.Tendrils({drawingBufferWidth: 32, drawingBufferHeight: 32}, {});
    t.resize(); t.setup(32);
    const h = t.particles.handle, out = {};
    out.bucket0 = t.particles.option('bucket');
    out.bucket1 = t.particles.option('bucket', 1);
    out.resort = t.particles.option('resortSteps', 7);
    let threw = false; try { native.option(h, native.OPT_BUCKET, 5); } catch (e) { threw = /TH_OPT_BUCKET/.test(String(e)); }
    out.threw = threw;
    const id = T.Particles.commLoopbackId();
    out.idBytes = id.length;
    native.commInit(h, id, 0, 1);
    out.comm = native.commQuery(h);
    out.stats = native.statsGlobal(h, 0.01).particles;
    native.commDestroy(h);
    t.dispose();
    console.log(JSON.stringify(out));
    """)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["bucket0"] == -1 and out["bucket1"] == 1 and out["resort"] == 7 and out["threw"] and out["idBytes"] == 128
    assert out["comm"]["active"] == 2 and out["comm"]["world"] == 1 and out["stats"] == 32 * 32


def test_timer_matches_reference_semantics():
    """Fixed-step, wall-clock, pause, end and loop behaviour of src/timer.js, JS mirror vs Python mirror."""
    from tendrils_amd.timer import Timer
    script = """
    const {Timer}=require('./tendrils_amd/js/timer');
    const out=[];
    let t=Object.assign(new Timer(1000,1000),{step:1000/60});
    for(let k=0;k<3;++k){t.tick(); out.push([t.time,t.dt]);}
    t.paused=true; t.tick(); out.push([t.time,t.dt,t.offset]); t.paused=false;
    t.end=80; t.tick(); out.push([t.time,t.dt,t.paused]); t.tick(); out.push([t.time,t.dt,t.paused]);
    t=Object.assign(new Timer(0,0),{step:30,end:100,loop:true}); for(let k=0;k<5;++k){t.tick(); out.push([t.time,t.dt]);}
    t=new Timer(5000,5000); t.tick(5250); out.push([t.time,t.dt]); t.rate=2; t.tick(5300); out.push([t.time,t.dt]);
    console.log(JSON.stringify(out));
    """
    r = node(script)
    assert r.returncode == 0, r.stderr
    js = json.loads(r.stdout)
    out = []
    t = Timer(1000, 1000)
    t.step = 1000 / 60
    for _ in range(3):
        t.tick()
        out.append([t.time, t.dt])
    t.paused = True
    t.tick()
    out.append([t.time, t.dt, t.offset])
    t.paused = False
    t.end = 80
    t.tick()
    out.append([t.time, t.dt, t.paused])
    t.tick()
    out.append([t.time, t.dt, t.paused])
    t = Timer(0, 0)
    t.step, t.end, t.loop = 30, 100, True
    for _ in range(5):
        t.tick()
        out.append([t.time, t.dt])
    t = Timer(5000, 5000)
    t.tick(5250)
    out.append([t.time, t.dt])
    t.rate = 2
    t.tick(5300)
    out.append([t.time, t.dt])
    assert js == json.loads(json.dumps(out))


def test_no_gpu_throws_a_js_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = node("const T=require('./tendrils_amd/js');const t=new T.Tendrils({drawingBufferWidth:8,drawingBufferHeight:8});"
             "t.resize();try{t.setup(8);console.log('no error')}catch(e){console.log('threw '+e.message)}")
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("threw tendrils_hip th_create")


def run_case(tmp_path, spec, arrays):
    for name, arr in arrays.items():
        np.ascontiguousarray(arr).tofile(str(tmp_path / name))
    (tmp_path / "case.json").write_text(json.dumps(spec))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "run_case.js"), str(tmp_path / "case.json")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["logic_default_64", "logic_target_strong_64", "logic_multistep_64",
                                  "logic_npot_48", "logic_c1_256"])
def test_js_step_matches_reference_bits(tmp_path, name):
    fx = load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    m = fx["meta"]
    n = m["N"]
    arrays = {"state.bin": fx["state"], "flow.bin": fx["flow"]}
    spec = dict(kind="logic", N=n, viewRes=m["viewRes"], flowShape=m["flowShape"], viewSize=m["viewSize"],
                state=state_overrides(m), steps=m["steps"], time0=m["times"][0] - m["dts"][0],
                times=m["times"], dts=m["dts"],
                inputs=dict(state="state.bin", flow="flow.bin"))
    if "targets" in fx:
        arrays["targets.bin"] = fx["targets"]
        spec["inputs"]["targets"] = "targets.bin"
    if m["steps"] > 1:
        spec["follow"] = []
        for k in range(m["steps"] - 1):
            arrays["follow_%d.bin" % k] = fx["out"][k]
            spec["follow"].append("follow_%d.bin" % k)
    run_case(tmp_path, spec, arrays)
    for k in range(m["steps"]):
        got = np.fromfile(str(tmp_path / ("out_%d.bin" % k)), np.float32).reshape(n, n, 4)
        ok = bits_equal(got, fx["out"][k]).all(-1)
        assert (ok | ~fx["valid"][k]).all(), "%s step %d" % (name, k)
    res = json.loads((tmp_path / "result.json").read_text())
    assert res["time"] == m["times"][-1]


@pytest.mark.gpu
def test_js_optical_flow_and_spawners(tmp_path, oracle):
    from test_spawn_oracle import oracle_spawn
    # optical flow
    fx = load(os.path.join(ROOT, "tests", "golden", "of_demo_240x135.npz"))
    m = fx["meta"]
    f0, f1, dst = of_inputs(m)
    d = tmp_path / "of"
    d.mkdir()
    run_case(d, dict(kind="optical_flow", N=8, viewRes=m["out"], frame=m["frame"], uniforms=m["uniforms"],
                     inputs=dict(flow="flow.bin", last="last.bin", view="view.bin")),
             {"flow.bin": dst, "last.bin": f0, "view.bin": f1})
    got = np.fromfile(str(d / "out_0.bin"), np.float32).reshape(m["out"][1], m["out"][0], 4)
    assert bits_equal(of_expected(fx, got), fx["out"]).all()
    # spawners: bit-equal to the oracle (same pinned sin/cos)
    for name in ("spawn_ball_demo_128", "spawn_flow_sample_64", "spawn_data_sample_64"):
        fx = load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        m = fx["meta"]
        d = tmp_path / name
        d.mkdir()
        spec = dict(kind=m["kind"], N=m["N"], viewRes=[96, 54], uniforms=m["uniforms"], inputs={})
        arrays = {}
        if m["kind"] == "spawn_sample":
            spec.update(apply=m["apply"], state={"flowDecay": m["uniforms"].get("flowDecay", 0.005)})
            arrays["state.bin"] = fx["state"]
            spec["inputs"]["state"] = "state.bin"
            if m["apply"] == 0:
                arrays["flow.bin"] = fx["data"]
                spec["inputs"]["flow"] = "flow.bin"
        run_case(d, spec, arrays)
        got = np.fromfile(str(d / "out_0.bin"), np.float32).reshape(m["N"], m["N"], 4)
        assert bits_equal(got, oracle_spawn(oracle, fx)).all(), name


@pytest.mark.gpu
def test_js_step_n_matches_python_single_steps(tmp_path):
    """Tendrils.stepN in the Node host (hipGraph replay) == the same steps issued one by one from Python."""
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    n, steps = 128, 6
    rng = np.random.default_rng(8)
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.01, .01, (n, n, 2))
    fl = np.zeros((54, 96, 4), np.float32)
    fl[..., :2] = rng.uniform(-.01, .01, (54, 96, 2))
    fl[..., 2] = 1990.0
    run_case(tmp_path, dict(kind="logic", N=n, viewRes=[96, 54], steps=steps, time0=2000.0, stepN=True,
                            inputs=dict(state="state.bin", flow="flow.bin")), {"state.bin": st, "flow.bin": fl})
    got = np.fromfile(str(tmp_path / ("out_%d.bin" % (steps - 1))), np.float32).reshape(n, n, 4)
    t = ta.Tendrils(View(96, 54))
    t.resize()
    t.setup(n)
    t.particles.upload_texels(st)
    t.flow.set_pixels(fl)
    t.timer.time = 2000.0
    for _ in range(steps):
        t.timer.tick()
        t.step()
    want = t.particles.read(0)
    t.dispose()
    assert bits_equal(got, want).all()


@pytest.mark.gpu
def test_js_frame_loop_step_and_draw(tmp_path, oracle):
    """Node host: Tendrils.step().draw() per frame (flow deposit closing the loop) == the oracle's step + deposit."""
    n, frames = 48, 4
    rng = np.random.default_rng(21)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-0.9, 0.9, (n, n, 2)) * [1.0, 0.5]
    st[..., 2:] = rng.uniform(-.008, .008, (n, n, 2))
    run_case(tmp_path, dict(kind="frames", N=n, viewRes=[96, 54], frames=frames, time0=3000.0,
                            inputs=dict(state="state.bin")), {"state.bin": st})
    got_state = np.fromfile(str(tmp_path / "state.out.bin"), np.float32).reshape(n, n, 4)
    got_flow = np.fromfile(str(tmp_path / "flow.out.bin"), np.float32).reshape(54, 96, 4)
    res = json.load(open(str(tmp_path / "result.json")))
    DEFAULT_STATE = oracle.DEFAULT_STATE
    cur, prev, flow, time = st.copy(), st.copy(), np.zeros((54, 96, 4), np.float32), 3000.0
    dt = 1000.0 / 60.0
    counts = []
    for _ in range(frames):
        time += dt
        u = oracle.logic_uniforms(n, n, time, dt, view_size=(1.0, 96 / 54), **DEFAULT_STATE)
        prev, cur = cur, oracle.logic_step(u, cur, flow)
        flow, k = oracle.flow_deposit(cur, prev, flow, time, view_size=(1.0, 96 / 54), speedLimit=DEFAULT_STATE["speedLimit"])
        counts.append(k)
    assert res["fragments"] == counts and abs(res["time"] - time) < 1e-9
    assert bits_equal(got_state, cur).all() and bits_equal(got_flow, flow).all()
    lines = np.fromfile(str(tmp_path / "lines.out.bin"), np.float32).reshape(-1, 12)
    assert bits_equal(lines, oracle.export_lines(cur, prev, time, view_size=(1.0, 96 / 54))).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["spawn_image_direct_64", "spawn_image_best_sample_64"])
def test_js_image_spawners(tmp_path, oracle, name):
    from test_spawn_oracle import oracle_spawn
    fx = load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    m, un = fx["meta"], fx["meta"]["uniforms"]
    ih, iw = fx["data"].shape[:2]
    run_case(tmp_path, dict(kind="spawn_image", N=m["N"], viewRes=[96, 54], direct=m["kind"] == "spawn_direct",
                            jitterRad=2 if any(un["jitter"]) else 0, imageShape=[iw, ih], uniforms=un,
                            inputs=dict(state="state.bin", image="image.bin")),
             {"state.bin": fx["state"], "image.bin": fx["data"]})
    got = np.fromfile(str(tmp_path / "out_0.bin"), np.float32).reshape(m["N"], m["N"], 4)
    assert bits_equal(got, oracle_spawn(oracle, fx)).all()


@pytest.mark.gpu
def test_js_geometry_spawner(tmp_path, oracle):
    fx = load(os.path.join(ROOT, "tests", "golden", "geometry_triangles_96x54.npz"))
    n = 64
    rng = np.random.default_rng(3)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    st[..., 2:] = rng.uniform(-.004, .004, (n, n, 2))
    st[rng.random((n, n)) < 0.3] = [-1e6, -1e6, 0, 0]
    run_case(tmp_path, dict(kind="geometry", N=n, viewRes=[480, 270], speed=0.005, bias=1e2 / 5e-3, time0=2000.0,
                            positions=[float(v) for v in fx["positions"]], inputs=dict(state="state.bin")),
             {"state.bin": st})
    got = np.fromfile(str(tmp_path / "out_0.bin"), np.float32).reshape(n, n, 4)
    res = json.load(open(str(tmp_path / "result.json")))
    assert res["moved"]
    u = oracle.spawn_sample_uniforms(n, n, res["time"], 6, 3, spawnSize=[1, 1], jitter=res["jitter"], speed=0.005,
                                     bias=1e2 / 5e-3, spawnMatrix=[1, 0, 0, 0, 1, 0, 0, 0, 1])
    assert bits_equal(got, oracle.spawn_sample(u, st, fx["out"])).all()


@pytest.mark.gpu
def test_js_exchange_primitives_world_size_1(oracle):
    """The multi-GPU exchange primitives through the N-API shim (device addresses as BigInt): at one owner, emit followed
    by merge of the emitted buffers is the flow pass - the flow texture must equal the oracle's deposit bit for bit."""
    import base64
    import shutil
    import subprocess
    n, view = 40, (96, 54)               # (the state travels on node's command line)
    rng = np.random.default_rng(31)
    prev = np.zeros((n, n, 4), np.float32)
    prev[..., :2] = rng.uniform(-0.9, 0.9, (n, n, 2)) * [1.0, 54 / 96]
    prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
    cur = prev.copy()
    cur[..., :2] += rng.uniform(-.06, .06, (n, n, 2)).astype(np.float32)
    script = """
    const T = require('./tendrils_amd/js');
    const native = require('./tendrils_amd/js/native');
    const cfg = JSON.parse(process.argv[1]);
    const f32 = (b) => new Float32Array(new Uint8Array(Buffer.from(b, 'base64')).buffer);
    const t = new T.Tendrils({drawingBufferWidth: cfg.view[0], drawingBufferHeight: cfg.view[1]}, {});
    t.resize(); t.setup(cfg.n);
    t.particles.uploadTexels(f32(cfg.cur), 0); t.particles.uploadTexels(f32(cfg.prev), 1);
    t.timer.time = cfg.time;
    const h = t.particles.handle;
    native.depositSetOwners(h, 1);
    native.depositSetHalo(h, null, null);
    const e = native.depositEmit(h, new Float32Array([t.viewSize[0], t.viewSize[1], t.timer.time, t.state.speedLimit]));
    if (typeof e.keys !== 'bigint' || typeof native.flowDevicePtr(h) !== 'bigint') throw new Error('addresses are BigInt');
    native.depositMerge(h, e.keys, e.colors, e.count);
    const out = {flow: Buffer.from(t.flow.read().buffer).toString('base64'), count: e.count,
                 state0: native.stateDevicePtr(h, 0) !== native.stateDevicePtr(h, 1)};
    t.dispose();
    console.log(JSON.stringify(out));
    """
    cfg = dict(n=n, view=list(view), time=800.0,
               cur=base64.b64encode(cur.tobytes()).decode(), prev=base64.b64encode(prev.tobytes()).decode())
    r = subprocess.run([shutil.which("node"), "-e", script, json.dumps(cfg)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout)
    got = np.frombuffer(base64.b64decode(res["flow"]), np.float32).reshape(view[1], view[0], 4)
    want, frags = oracle.flow_deposit(cur, prev, np.zeros((view[1], view[0], 4), np.float32), 800.0, view_size=(1.0, 96 / 54))
    assert res["count"] == frags and res["state0"]
    assert bits_equal(got, want).all()
