#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - golden-vector generator (build container only).

Runs the reference's OWN compiled shaders (oracle/ref_runner.py: the prebuilt
bundle under /root/reference/docs/js, executed unmodified by headless Chromium +
SwiftShader) on seeded synthetic textures and writes the inputs, the exact
uniform values and the reference outputs as .npz fixtures under tests/golden/.
Only numbers are written; no reference text reaches the repository.

    python oracle/gen_fixtures.py [--only NAME_SUBSTRING]

Every fixture holds:  kind, the inputs (state/flow/targets/...), `uniforms`
(json), `out` (reference output, [K,N,N,4] for logic cases) and `valid`
(boolean mask of texels whose reference value is defined; see QUAD NOTE).

QUAD NOTE.  logic.frag samples `flow` inside the non-uniform `if(pos != inert)`
branch, which GLSL ES 1.0 leaves undefined (implicit derivatives in divergent
control flow).  SwiftShader's behaviour there: in the left-most 2x2 pixel quads
(x in {0,1}) whose lane 0 (x = 0, even y) is inert, the other three lanes read a
zero flow texel.  Those (at most 3 per quad) texels are masked out of `valid`;
everything else is compared bit-for-bit.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_runner import RefRunner  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
INERT = np.float32(-1e6)


def rand_state(rng, n, inert_frac=0.0, pos_range=1.0, vel_range=0.01):
    st = np.empty((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-pos_range, pos_range, (n, n, 2))
    st[..., 2:] = rng.uniform(-vel_range, vel_range, (n, n, 2))
    if inert_frac > 0:
        st[rng.random((n, n)) < inert_frac] = [INERT, INERT, 0, 0]
    return st


def rand_flow(rng, w, h, time, mag=0.01, age=120.0):
    fl = np.zeros((h, w, 4), np.float32)
    fl[..., :2] = rng.uniform(-mag, mag, (h, w, 2))
    fl[..., 2] = time - rng.uniform(0, age, (h, w))     # deposit time (ms), some fully decayed
    fl[..., 3] = rng.uniform(0, 1, (h, w))
    return fl


def valid_mask(state):
    """See QUAD NOTE."""
    n = state.shape[0]
    inert = (state[..., 0] == INERT) & (state[..., 1] == INERT)
    valid = np.ones((n, n), bool)
    for y in range(0, n - 1, 2):
        if inert[y, 0]:
            for (yy, xx) in ((y, 1), (y + 1, 0), (y + 1, 1)):
                if not inert[yy, xx]:
                    valid[yy, xx] = False
    return valid


def save(name, **arrs):
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-34s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


def logic_case(r, name, n, seed, uniforms=None, time0=5000.0, steps=1, view=(48, 48),
               flow_shape=None, inert_frac=0.09, pos_range=1.0, vel_range=0.01, flow_mag=0.01,
               with_targets=False, zero_flow=False, state=None, view_size=None):
    rng = np.random.default_rng(seed)
    st = rand_state(rng, n, inert_frac, pos_range, vel_range) if state is None else state
    fw, fh = flow_shape if flow_shape else view
    t_first = time0 + 1000.0 / 60.0
    fl = np.zeros((fh, fw, 4), np.float32) if zero_flow else rand_flow(rng, fw, fh, t_first, flow_mag)
    tg = None
    if with_targets:
        tg = np.zeros((n, n, 4), np.float32)
        tg[..., :2] = rng.uniform(-1, 1, (n, n, 2))
    outs, res = r.logic(st, flow=fl, targets=tg, uniforms=uniforms or {}, time0=time0, steps=steps,
                        view=view, flow_shape=(fw, fh), view_size=view_size, return_each=True)
    # chain the validity mask through the trajectory: an undefined texel stays undefined
    valid = np.ones((steps, n, n), bool)
    cur, v = st, np.ones((n, n), bool)
    for k in range(steps):
        v = v & valid_mask(cur)
        valid[k] = v
        cur = outs[k]
    meta = dict(kind="logic", N=n, steps=steps, times=res["times"], dts=res["dts"],
                viewSize=res["viewSize"], viewRes=res["viewRes"], flowShape=res["flowShape"],
                state={k: v for k, v in res["state"].items() if isinstance(v, (int, float))},
                overrides=uniforms or {}, seed=seed, ref_ms=res["ms"])
    arrs = dict(state=st, flow=fl, out=np.stack(outs), valid=valid, uniforms=json.dumps(meta))
    if tg is not None:
        arrs["targets"] = tg
    save(name, **arrs)


def gen_logic(r, only):
    cases = [
        # name, kwargs
        ("logic_default_64", dict(n=64, seed=101)),
        ("logic_flow_only_64", dict(n=64, seed=102, uniforms={"noiseWeight": 0})),
        # preset 'Flow Only' src/demo.main.js:1865-1870
        ("logic_flow_only_preset_64", dict(n=64, seed=103, uniforms={
            "flowDecay": 0.001, "forceWeight": 0.014, "noiseWeight": 0})),
        # preset 'Noise Only' src/demo.main.js:1828-1838
        ("logic_noise_only_preset_64", dict(n=64, seed=104, uniforms={
            "flowWeight": 0, "noiseWeight": 0.003, "noiseScale": 1.5, "varyNoiseScale": -30,
            "noiseSpeed": 0.00025, "varyNoiseSpeed": -0.3})),
        # tracksStart.tendrils3 target pull (src/demo.main.js:905-909), exaggerated too
        ("logic_target_64", dict(n=64, seed=105, uniforms={"target": 0.000005, "varyTarget": 1},
                                 with_targets=True)),
        ("logic_target_strong_64", dict(n=64, seed=106, uniforms={"target": 0.003, "varyTarget": -0.5},
                                        with_targets=True)),
        # speed cap active: large flow forces
        ("logic_speedcap_64", dict(n=64, seed=107, uniforms={"forceWeight": 0.2}, flow_mag=0.05,
                                   vel_range=0.05)),
        ("logic_all_inert_64", dict(n=64, seed=108, inert_frac=1.1)),
        # non-square view: viewSize = coverAspect([96,54]) = [1, 1.7778]; particles beyond the view
        ("logic_viewsize_64", dict(n=64, seed=109, view=(96, 54), pos_range=1.5)),
        # trajectories: K = 8 steps, flow held fixed
        ("logic_multistep_64", dict(n=64, seed=110, steps=8, time0=1000.0)),
        # non power-of-two state texture (pins `/dataRes`)
        ("logic_npot_48", dict(n=48, seed=111, view=(40, 30))),
        # C1: 256x256, fresh (zero) flow, default state, first tick from t=0
        ("logic_c1_256", dict(n=256, seed=12345, time0=0.0, view=(256, 256), zero_flow=True,
                              inert_frac=0.0)),
        # late time (time*noiseSpeed large) + long-decayed flow
        ("logic_latetime_64", dict(n=64, seed=112, time0=3.6e6)),
    ]
    for name, kw in cases:
        if only and only not in name:
            continue
        logic_case(r, name, **kw)

    # zero speed -> 0/0 = NaN (src/logic.frag:92-94): zero velocity, no forces
    if not only or only in "logic_zero_speed_32":
        rng = np.random.default_rng(113)
        st = rand_state(rng, 32, 0.0)
        st[::2, :, 2:] = 0.0
        logic_case(r, "logic_zero_speed_32", n=32, seed=113, uniforms={"noiseWeight": 0},
                   zero_flow=True, state=st, view=(32, 32))


def gen_logic_denormal(r, only):
    """Velocities whose squares underflow: speed = length(newVel) passes through the denormal range and reaches
    0 -> 0/0 = NaN (src/logic.frag:92-94).  Pins the flush-to-zero behaviour of the reference's arithmetic."""
    name = "logic_denormal_16"
    if only and only not in name:
        return
    rng = np.random.default_rng(114)
    n = 16
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-0.9, 0.9, (n, n, 2))
    mag = 10.0 ** rng.uniform(-24, -17, (n, n))
    ang = rng.uniform(0, 2 * np.pi, (n, n))
    st[..., 2] = mag * np.cos(ang)
    st[..., 3] = mag * np.sin(ang)
    logic_case(r, name, n=n, seed=114, uniforms={"noiseWeight": 0}, zero_flow=True, state=st, view=(16, 16), steps=3,
               time0=1000.0)


def gen_logic_4096(r, only):
    """C3-sized state texture: the index i = (x+.5 + (y+.5)*W)/(W*H) loses low bits in fp32 once
    (y+.5)*W >= 2^24 (src/logic.frag:57-58) - must be mirrored, not fixed.  The state is generated
    inside the page from an integer hash (tests/helpers.py:hashed_state is the same generator); only
    row bands of the reference output are stored."""
    name = "logic_4096_bands"
    if only and only not in name:
        return
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from helpers import hashed_state
    n, seed, inert_mod = 4096, 20240, 37
    rng = np.random.default_rng(seed)
    fw, fh = 240, 135
    time0 = 60000.0
    fl = rand_flow(rng, fw, fh, time0 + 1000.0 / 60.0, 0.01)
    bands = [(0, 2), (1022, 1026), (2047, 2049), (3070, 3074), (4094, 4096)]
    outs, res = r.logic(None, flow=fl, time0=time0, steps=1, view=(fw, fh), flow_shape=(fw, fh), rows=bands,
                        state_gen=dict(N=n, seed=seed, inertMod=inert_mod))
    # validity (QUAD NOTE) per band from the regenerated input
    valids = []
    for (a, b) in bands:
        st = hashed_state(n, seed, inert_mod, rows=(a - a % 2, b + b % 2))
        inert = (st[..., 0] == INERT) & (st[..., 1] == INERT)
        v = np.ones(inert.shape, bool)
        for yy in range(0, inert.shape[0] - 1, 2):
            if inert[yy, 0]:
                for (dy, dx) in ((0, 1), (1, 0), (1, 1)):
                    if not inert[yy + dy, dx]:
                        v[yy + dy, dx] = False
        valids.append(v[a % 2: a % 2 + (b - a)])
    meta = dict(kind="logic_bands", N=n, seed=seed, inertMod=inert_mod, bands=bands, times=res["times"], dts=res["dts"],
                viewSize=res["viewSize"], viewRes=res["viewRes"], flowShape=res["flowShape"],
                state={k: v for k, v in res["state"].items() if isinstance(v, (int, float))}, ref_ms=res["ms"])
    save(name, flow=fl, out=np.concatenate(outs[0]), valid=np.concatenate(valids), uniforms=json.dumps(meta))


def band_valid(n, seed, inert_mod, bands):
    """QUAD NOTE validity of the rows in `bands`, from the regenerated hashed input."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from helpers import hashed_state
    valids = []
    for (a, b) in bands:
        st = hashed_state(n, seed, inert_mod, rows=(a - a % 2, b + b % 2))
        inert = (st[..., 0] == INERT) & (st[..., 1] == INERT)
        v = np.ones(inert.shape, bool)
        for yy in range(0, inert.shape[0] - 1, 2):
            if inert[yy, 0]:
                for (dy, dx) in ((0, 1), (1, 0), (1, 1)):
                    if not inert[yy + dy, dx]:
                        v[yy + dy, dx] = False
        valids.append(v[a % 2: a % 2 + (b - a)])
    return np.concatenate(valids)


def gen_logic_config_bands(r, only):
    """BASELINE.json configs C2 and C4 as reference captures (row bands, hashed state generated in the page).
    C2: 1024^2, "curl-noise" flow 1024x1024 (tests/helpers.py:curl_flow - regenerated by the tests, not stored),
        K = 4 steps with every step's bands kept.
    C4: 8192^2 (= MAX_TEXTURE_SIZE of the reference's GL here), bands around the 8-way shard boundaries."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from helpers import curl_flow
    name = "logic_c2_1024_bands"
    if not only or only in name:
        n, seed, inert_mod, steps = 1024, 31337, 41, 4
        time0 = 5000.0
        fw = fh = 1024
        flow_gen = dict(kind="curl", w=fw, h=fh, seed=77, time=time0 + 1000.0 / 60.0)
        fl = curl_flow(fw, fh, flow_gen["seed"], flow_gen["time"])
        bands = [(0, 4), (510, 514), (1020, 1024)]
        outs, res = r.logic(None, flow=fl, time0=time0, steps=steps, view=(fw, fh), flow_shape=(fw, fh), rows=bands,
                            state_gen=dict(N=n, seed=seed, inertMod=inert_mod), return_each=True)
        meta = dict(kind="logic_bands", N=n, seed=seed, inertMod=inert_mod, bands=bands, steps=steps, flowGen=flow_gen,
                    times=res["times"], dts=res["dts"], viewSize=res["viewSize"], viewRes=res["viewRes"],
                    flowShape=res["flowShape"],
                    state={k: v for k, v in res["state"].items() if isinstance(v, (int, float))}, ref_ms=res["ms"])
        save(name, out=np.stack([np.concatenate(o) for o in outs]), valid=band_valid(n, seed, inert_mod, bands),
             uniforms=json.dumps(meta))
    name = "logic_c4_8192_bands"
    if not only or only in name:
        n, seed, inert_mod = 8192, 8192017, 29
        rng = np.random.default_rng(seed)
        fw, fh = 240, 135
        time0 = 120000.0
        fl = rand_flow(rng, fw, fh, time0 + 1000.0 / 60.0, 0.01)
        bands = [(0, 1), (1023, 1025), (2047, 2049), (4095, 4097), (6143, 6145), (7167, 7169), (8191, 8192)]
        outs, res = r.logic(None, flow=fl, time0=time0, steps=1, view=(fw, fh), flow_shape=(fw, fh), rows=bands,
                            state_gen=dict(N=n, seed=seed, inertMod=inert_mod))
        meta = dict(kind="logic_bands", N=n, seed=seed, inertMod=inert_mod, bands=bands, steps=1,
                    times=res["times"], dts=res["dts"], viewSize=res["viewSize"], viewRes=res["viewRes"],
                    flowShape=res["flowShape"],
                    state={k: v for k, v in res["state"].items() if isinstance(v, (int, float))}, ref_ms=res["ms"])
        save(name, flow=fl, out=np.stack([np.concatenate(outs[0])]), valid=band_valid(n, seed, inert_mod, bands),
             uniforms=json.dumps(meta))


def gen_deposit(r, only):
    """Tendrils.draw()'s flow pass (src/index.js:278-303) on the reference: current / previous state textures in,
    flow FBO out.  The output is stored sparsely (texels that differ from the initial flow).  Every vertex pair
    drawn here has two live vertices (the reference's behaviour with an inert vertex is undefined, see the oracle)."""
    def case(name, n, view, seed, pos_range=1.0, step=0.03, with_flow=False, inert=0.0, layout=None, time=5016.67,
             uniforms=None):
        if only and only not in name:
            return
        rng = np.random.default_rng(seed)
        fw, fh = view
        prev = np.zeros((n, n, 4), np.float32)
        if layout == "hashed":           # inputs regenerated by the tests (tests/helpers.py:deposit_hashed_inputs), not stored
            sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
            from helpers import deposit_hashed_inputs
            cur, prev = deposit_hashed_inputs(n, seed, pos_range, step, inert)
            out, res = r.deposit(cur, prev, uniforms=uniforms, time=time, view=view)
            idx = np.flatnonzero((out != 0).any(-1)).astype(np.int32)
            meta = dict(kind="deposit", N=n, viewRes=[fw, fh], viewSize=res["viewSize"], time=time,
                        speedLimit=res["state"]["speedLimit"], seed=seed, lineWidthRange=res["lineWidthRange"],
                        overrides=uniforms or {}, hashed=dict(pos_range=pos_range, step=step, inert=inert))
            save(name, idx=idx, val=out.reshape(-1, 4)[idx], uniforms=json.dumps(meta))
            return
        if layout == "cells":            # one short line per 16x16-texel cell: isolated lines, ties provoked
            cell = fw // n
            for y in range(n):
                for x in range(n):
                    p0 = np.array([x * cell + 5 + rng.uniform(0, 6), y * cell + 5 + rng.uniform(0, 6)])
                    if rng.random() < 0.15:
                        p0 = np.round(p0 * 16) / 16
                    prev[y, x, :2] = p0 / fw * 2 - 1
            ang = rng.uniform(0, 2 * np.pi, (n, n))
            length = rng.uniform(0, step, (n, n))
            d = np.stack([np.cos(ang), np.sin(ang)], -1) * length[..., None] / fw * 2
        elif layout == "border":         # endpoints concentrated around the edge of the view
            side = rng.integers(0, 4, (n, n))
            u = rng.uniform(-1.05, 1.05, (n, n))
            e = rng.uniform(0.93, 1.07, (n, n)) * np.where(rng.random((n, n)) < 0.5, 1, -1)
            prev[..., 0] = np.where(side < 2, e, u)
            prev[..., 1] = np.where(side < 2, u, e) * (fh / fw)
            d = rng.uniform(-step, step, (n, n, 2))
        else:
            prev[..., :2] = rng.uniform(-pos_range, pos_range, (n, n, 2))
            d = rng.uniform(-step, step, (n, n, 2))
        prev[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
        cur = prev.copy()
        cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2))
        cur[..., :2] = (prev[..., :2] + d).astype(np.float32)
        if inert > 0:
            k = rng.random((n, n)) < inert
            cur[k] = [INERT, INERT, 0, 0]
            prev[k] = [INERT, INERT, 0, 0]
        fl = None
        if with_flow:
            fl = rand_flow(rng, fw, fh, time, 0.01)
        out, res = r.deposit(cur, prev, flow=fl, uniforms=uniforms, time=time, view=view)
        base = fl if fl is not None else np.zeros((fh, fw, 4), np.float32)
        idx = np.flatnonzero((out != base).any(-1)).astype(np.int32)
        meta = dict(kind="deposit", N=n, viewRes=[fw, fh], viewSize=res["viewSize"], time=time,
                    speedLimit=res["state"]["speedLimit"], seed=seed, lineWidthRange=res["lineWidthRange"],
                    overrides=uniforms or {})
        arrs = dict(current=cur, previous=prev, idx=idx, val=out.reshape(-1, 4)[idx], uniforms=json.dumps(meta))
        if fl is not None:
            arrs["flow"] = fl
        save(name, **arrs)

    case("deposit_isolated_64", 64, (1024, 1024), 301, layout="cells", step=4.5, time=1234.0)
    case("deposit_subtexel_64", 64, (1024, 1024), 302, layout="cells", step=0.15, time=1234.0)
    case("deposit_overlap_32", 32, (64, 64), 303)
    case("deposit_nonsquare_flow_32", 32, (96, 54), 304, with_flow=True, inert=0.1)
    case("deposit_border_48", 48, (96, 54), 305, layout="border", step=0.06)
    case("deposit_long_lines_32", 32, (64, 64), 306, pos_range=1.3, step=0.2)
    case("deposit_speedlimit_32", 32, (80, 60), 307, uniforms={"speedLimit": 0.004})
    # a frame-like load: 65 536 particles over a 480x270 field, steps of ~1-2 texels, a fifth outside the view,
    # blended over an existing field: heavy overlap, every kind of tie and edge crossing at once
    case("deposit_frame_256", 256, (240, 135), 308, layout="hashed", pos_range=1.15, step=0.012, inert=19)


def gen_loop(r, only):
    """The reference's frame loop, closed: Tendrils.step() then Tendrils.draw() for K frames from an empty flow
    field - the deposited wake of every frame steers the next step.  No inert particles (see the deposit's
    documented deviation)."""
    name = "loop_frames_64"
    if only and only not in name:
        return
    n, view, frames = 64, (96, 54), 6
    rng = np.random.default_rng(31)
    st = np.zeros((n, n, 4), np.float32)
    st[..., :2] = rng.uniform(-0.95, 0.95, (n, n, 2)) * [1.0, 0.55]
    st[..., 2:] = rng.uniform(-.008, .008, (n, n, 2))
    outs, res = r.logic(st, flow=None, time0=1000.0, steps=frames, view=view, return_each=True, draw=True)
    fl = res["flow_out"]
    idx = np.flatnonzero((fl != 0).any(-1)).astype(np.int32)
    meta = dict(kind="loop", N=n, frames=frames, viewRes=list(view), viewSize=res["viewSize"], times=res["times"],
                dts=res["dts"], state={k: v for k, v in res["state"].items() if isinstance(v, (int, float))})
    save(name, state=st, out=np.stack(outs), flow_idx=idx, flow_val=fl.reshape(-1, 4)[idx], uniforms=json.dumps(meta))


def gen_optical_flow(r, only):
    """One blended pass of the reference's optical-flow shader (docs/js/demo.js:73) per case.
    Frames are regenerated from seeds by tests/helpers.py:synth_frame; only parameters and the
    reference output (whole texture, or row bands for the 1080p case) are stored."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from helpers import of_inputs
    cases = [
        # name, frame (w,h), out (w,h), uniforms, bands
        ("of_default_64", (64, 48), (64, 48), dict(viewSize=[1, 1], scaleUV=[1, -1], offset=1.0 / 64, speed=1,
                                                   speedLimit=1, time=1234.5), None),
        # demo settings src/demo.main.js:526-530 (speed .08, offset .1, scaleUV [-1,-1]) + tendrils speedLimit
        ("of_demo_96x64", (64, 48), (96, 64), dict(viewSize=[1, 1.5], scaleUV=[-1, -1], offset=0.1, speed=0.08,
                                                   speedLimit=0.01, time=777.0), None),
        ("of_demo_240x135", (160, 90), (240, 135), dict(viewSize=[1, 240 / 135], scaleUV=[-1, -1], offset=0.1,
                                                        speed=0.08, speedLimit=0.01, time=5016.67), None),
        ("of_texel_offset_240x135", (240, 135), (240, 135), dict(viewSize=[1, 240 / 135], scaleUV=[-1, -1],
                                                                 offset=1.0 / 240, speed=0.08, speedLimit=0.01,
                                                                 time=5016.67), None),
        # C3: 1080p frames and flow; only three 8-row bands are kept
        ("of_c3_1080p", (1920, 1080), (1920, 1080), dict(viewSize=[1, 1920 / 1080], scaleUV=[-1, -1], offset=0.1,
                                                         speed=0.08, speedLimit=0.01, time=1000.0),
         [(0, 8), (536, 544), (1072, 1080)]),
    ]
    for name, fr, out, un, bands in cases:
        if only and only not in name:
            continue
        un = dict(un)
        un["lambda"] = 0.001
        seed = sum(map(ord, name))
        meta = dict(kind="optical_flow", frame=list(fr), out=list(out), uniforms=un, seed=seed,
                    shift8=[12, 6], bands=bands)          # frame1 = frame0 translated by (1.5, 0.75) px
        f0, f1, dst = of_inputs(meta)
        ow, oh = out
        ref = r.shader("optical_flow", (ow, oh), textures={"view": f1, "last": f0}, uniforms=un, blend=True, dst=dst)
        if bands:
            ref = np.concatenate([ref[a:b] for a, b in bands])
        save(name, out=ref, uniforms=json.dumps(meta))


def gen_spawn(r, only):
    """Respawn shaders (docs/js/demo.js:71-72): ball, flow-sample, data-sample.  Their hashes
    amplify the platform's sin(), so these captures pin statistics, not bits."""
    if not only or only in "spawn_ball_default_64 spawn_ball_demo_128":
        for name, n, un in (("spawn_ball_default_64", 64, dict(radius=1.0, speed=0.0)),
                            ("spawn_ball_demo_128", 128, dict(radius=0.3, speed=0.005))):   # src/demo.main.js:1402-1405
            ref = r.shader("spawn_ball", (n, n), uniforms=un)
            save(name, out=ref, uniforms=json.dumps(dict(kind="spawn_ball", N=n, uniforms=un)))
    rng = np.random.default_rng(4711)
    n, (fw, fh) = 64, (96, 54)
    st = rand_state(rng, n, 0.3, 1.0, 0.004)
    time = 2016.67
    fl = rand_flow(rng, fw, fh, time, 0.01, 100.0)
    ident = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    view_size = [1.0, 96 / 54]
    if not only or only in "spawn_flow_sample_64":
        # spawnFlow(): spawnSize = [1,-1]/viewSize, buffer = tendrils.flow  (src/demo.main.js:403-424);
        # jitter = aspect(viewRes, jitterRad=2) = 2/viewRes  (src/spawn/pixels/index.js:19,53)
        un = dict(dataRes=[n, n], geomRes=[n, 2 * n], spawnSize=[1 / view_size[0], -1 / view_size[1]],
                  jitter=[2 / fw, 2 / fh], time=time, speed=1.0, bias=1.0, flowDecay=0.005, spawnMatrix=ident)
        ref = r.shader("spawn_flow_sample", (n, n), textures={"particles": st, "spawnData": fl}, uniforms=un)
        save("spawn_flow_sample_64", state=st, data=fl, out=ref,
             uniforms=json.dumps(dict(kind="spawn_sample", N=n, samples=5, apply=0, uniforms=un)))
    if not only or only in "spawn_data_sample_64":
        # spawnFastest(): buffer = particles.buffers[0], spawnSize = particles.shape (src/demo.main.js:437-441)
        un = dict(dataRes=[n, n], geomRes=[n, 2 * n], spawnSize=[n, n], jitter=[2 / fw, 2 / fh], time=time,
                  speed=1.0, bias=1.0, spawnMatrix=ident)
        ref = r.shader("spawn_data_sample", (n, n), textures={"particles": st, "spawnData": st}, uniforms=un)
        save("spawn_data_sample_64", state=st, data=st, out=ref,
             uniforms=json.dumps(dict(kind="spawn_sample", N=n, samples=2, apply=1, uniforms=un)))


def gen_spawn_image(r, only):
    """Image spawners (src/demo.main.js:455-515): index.frag (direct) and best-sample.frag (6 samples) over an RGBA
    image held in a float texture (PixelSpawner's default buffer, src/spawn/pixels/index.js:17).  The image is a
    seeded 8-bit pattern / 255.  Direct spawn with zero jitter has no hash in it: positions are exact and the
    velocities differ from this build only through cos/sin."""
    rng = np.random.default_rng(4712)
    n, (iw, ih) = 64, (80, 60)
    img8 = rng.integers(0, 256, (ih, iw, 4), dtype=np.uint8)
    img8[..., 3] = rng.integers(128, 256, (ih, iw))
    img8[5:20, 10:30, :3] = [200, 30, 60]           # flat patches: grey (d = 0) and saturated
    img8[30:40, 40:60, :3] = 128
    img = (img8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    st = rand_state(rng, n, 0.3, 1.0, 0.004)
    time = 2016.67
    flip = [-1, 0, 0, 0, 1, 0, 0, 0, 1]              # mat3.scale(identity, [-1, 1]) (src/demo.main.js:462-463)
    for name, frag, jitter, speed in (("spawn_image_direct_64", "spawn_direct", [0.0, 0.0], 0.3),
                                      ("spawn_image_direct_jitter_64", "spawn_direct", [2 / 96, 2 / 54], 0.3),
                                      ("spawn_image_best_sample_64", "spawn_best_sample", [2 / 96, 2 / 54], 1.0)):
        if only and only not in name:
            continue
        un = dict(dataRes=[n, n], geomRes=[n, 2 * n], spawnSize=[1.0, 0.75], jitter=jitter, time=time, speed=speed,
                  bias=1.0, spawnMatrix=flip)
        ref = r.shader(frag, (n, n), textures={"particles": st, "spawnData": img}, uniforms=un)
        save(name, state=st, data=img, out=ref,
             uniforms=json.dumps(dict(kind="spawn_direct" if frag == "spawn_direct" else "spawn_sample", N=n,
                                      samples=6, apply=2, uniforms=un)))


def gen_geometry(r, only):
    """GeometrySpawner (src/spawn/geometry/index.js): the triangle draw into the spawner's buffer and the
    bright-sample.frag pass over it.  Triangles as shuffle() builds them (centre + two rim vertices), from a seed."""
    rng = np.random.default_rng(4713)
    fw, fh = 96, 54
    view_size = [1.0, fw / fh]
    pos = []
    for _ in range(7):
        ang = rng.uniform(0, 2 * np.pi)
        arc = 2 * np.pi * (0.01 + rng.uniform(0, 0.03) + (rng.random() < 0.5) * 0.25)
        r1, r2 = 0.25 + rng.uniform(0, 1.3), 0.25 + rng.uniform(0, 1.3)
        pos += [0.0, 0.0, np.cos(ang - arc) * r1, np.sin(ang - arc) * r1, np.cos(ang + arc) * r2, np.sin(ang + arc) * r2]
    pos = [float(np.float32(v)) for v in pos]
    if not only or only in "geometry_triangles_96x54":
        for name, col in (("geometry_triangles_96x54", [1, 1, 1, 1]), ("geometry_triangles_translucent_96x54", [0.9, 0.5, 0.2, 0.6])):
            img = r.shader("geometry_frag", (fw, fh), uniforms={"color": col, "viewSize": view_size}, blend=True,
                           vert="geometry_vert", positions=pos)
            save(name, positions=np.array(pos, np.float32), out=img,
                 uniforms=json.dumps(dict(kind="geometry", shape=[fw, fh], viewSize=view_size, color=col)))
    if not only or only in "spawn_geometry_bright_sample_64":
        n = 64
        st = rand_state(rng, n, 0.3, 1.0, 0.004)
        img = r.shader("geometry_frag", (fw, fh), uniforms={"color": [1, 1, 1, 1], "viewSize": view_size}, blend=True,
                       vert="geometry_vert", positions=pos)
        # src/demo.main.js:446-447: speed 0.005, bias 1e2/5e-3
        un = dict(dataRes=[n, n], geomRes=[n, 2 * n], spawnSize=[1.0, 1.0], jitter=[2 / fw, 2 / fh], time=2016.67,
                  speed=0.005, bias=1e2 / 5e-3, spawnMatrix=[1, 0, 0, 0, 1, 0, 0, 0, 1])
        ref = r.shader("spawn_bright_sample", (n, n), textures={"particles": st, "spawnData": img}, uniforms=un)
        save("spawn_geometry_bright_sample_64", state=st, data=img, out=ref,
             uniforms=json.dumps(dict(kind="spawn_sample", N=n, samples=6, apply=3, uniforms=un)))


def gen_view(r, only):
    """Tendrils.draw()'s VIEW pass (src/index.js:315-337, src/render/index.vert:58-100) on the reference, for the input
    states of some of the deposit fixtures: the default framebuffer (RGBA8) read back after draw(), on a context without
    multisampling (the view's width-1 lines then follow the flow pass's rasteriser).  Stored sparsely."""
    def case(name, source, uniforms=None, time=None):
        if only and only not in name:
            return
        d = np.load(os.path.join(GOLDEN, source + ".npz"))
        meta = json.loads(str(d["uniforms"]))
        cur, prev = d["current"], d["previous"]
        fw, fh = meta["viewRes"]
        t = meta["time"] if time is None else time
        out, res = r.deposit(cur, prev, uniforms=dict(meta.get("overrides", {}), **(uniforms or {})), time=t, view=(fw, fh), want_view=True)
        view = res["view_out"]
        idx = np.flatnonzero(view.any(-1)).astype(np.int32)
        st = res["state"]
        m = dict(kind="view", source=source, N=meta["N"], viewRes=[fw, fh], viewSize=res["viewSize"], time=t, samples=res["samples"],
                 render={k: st[k] for k in ("speedLimit", "flowDecay", "speedAlpha", "colorMapAlpha", "baseColor", "flowColor", "lineWidth")})
        save(name, idx=idx, val=view.reshape(-1, 4)[idx], uniforms=json.dumps(m))

    case("view_isolated_64", "deposit_isolated_64")
    case("view_overlap_32", "deposit_overlap_32")
    case("view_nonsquare_32", "deposit_nonsquare_flow_32")
    case("view_border_48", "deposit_border_48")
    case("view_colours_32", "deposit_long_lines_32", time=2718.0,
         uniforms=dict(baseColor=[0.9, 0.4, 0.1, 0.7], flowColor=[0.2, 0.7, 1.0, 0.35], speedAlpha=2.5, flowDecay=0.012))


BUFFER_SCRIPTS = {
    # the demo's loop (src/demo.main.js:89, 1082-1101): one buffer; per frame tick, step().draw(), then - to the screen -
    # drawFade() [the blur pass that reads buffers[0] there is out of scope], stepBuffers()
    "buffers_demo_loop_48": dict(num=1, ops=[["tickStep"], ["draw"], ["bind", -1], ["viewport"], ["drawFade"], ["stepBuffers"],
                                             ["tickStep"], ["draw"], ["bind", -1], ["drawFade"], ["stepBuffers"],
                                             ["read", 0], ["read", -1]]),
    # two buffers: the view lands in buffers[0], the ring rotates under it, copyBuffer blends a buffer into whatever is
    # bound, drawBuffer clears (autoClearView) or not, copies to the screen and rotates
    "buffers_copy_and_rotate_48": dict(num=2, ops=[["draw"], ["read", 0], ["read", 1], ["read", -1],
                                                   ["bind", 1], ["copyBuffer", 0], ["read", 1],
                                                   ["bind", -1], ["drawFill", [0.2, 0.5, 0.1, 0.7]], ["drawBuffer", 0], ["read", -1],
                                                   ["read", 0], ["read", 1],
                                                   ["stepBuffers"], ["tickStep"], ["draw"], ["read", 0], ["read", 1],
                                                   ["copyBuffer", 5], ["read", 0],
                                                   ["set", "autoClearView", True], ["drawBuffer", None], ["read", -1],
                                                   ["tickStep"], ["draw"], ["read", 0], ["read", 1], ["read", -1],
                                                   ["setupBuffers", 1], ["read", 0], ["setupBuffers", 3], ["read", 2]]),
}


def gen_buffers(r, only):
    """Tendrils.buffers (src/index.js:66-68, 172-184, 318-325, 359-391) on the reference: scripts of its own draw /
    copyBuffer / drawBuffer / stepBuffers / setupBuffers / clearView calls, and every image the script reads on the way
    (RGBA8; a context without multisampling).  Stored sparsely per read."""
    for name, sc in BUFFER_SCRIPTS.items():
        if only and only not in name:
            continue
        rng = np.random.default_rng(4711 + len(name))
        n, view = 48, (48, 32)
        prev = rand_state(rng, n, pos_range=0.9, vel_range=0.012)
        prev[..., 1] *= view[1] / view[0]
        cur = prev.copy()
        cur[..., :2] += rng.uniform(-.1, .1, (n, n, 2)).astype(np.float32)
        cur[..., 2:] = rng.uniform(-.012, .012, (n, n, 2)).astype(np.float32)
        uniforms = dict(baseColor=[1, 0.7, 0.3, 0.6], flowColor=[0.2, 1, 0.9, 0.3], fadeColor=[0.1, 0.2, 0.3, 0.25], speedAlpha=1.5)
        res = r.buffers(cur, prev, sc["ops"], num_buffers=sc["num"], uniforms=uniforms, time=2500.0, view=view)
        assert res["samples"] == 0 and len(res["images"]) == sum(1 for op in sc["ops"] if op[0] == "read")
        arrs = {}
        for k, img in enumerate(res["images"]):
            idx = np.flatnonzero(img.any(-1)).astype(np.int32)
            arrs["idx%d" % k], arrs["val%d" % k] = idx, img.reshape(-1, 4)[idx]
        m = dict(kind="buffers", N=n, viewRes=list(view), viewSize=res["viewSize"], time0=2500.0, numBuffers=sc["num"], ops=sc["ops"],
                 lengths=res["lengths"], reads=len(res["images"]), state=uniforms, samples=res["samples"])
        save(name, current=cur, previous=prev, uniforms=json.dumps(m), **arrs)
        print("wrote %s (%d reads; non-empty: %s)" % (name, len(res["images"]), [int(i.any()) for i in res["images"]]))


def gen_spawn_map(r, only):
    """Particles.spawn(map, pixels, offset) (src/particles.js:94-117): the [w, h, 4] staging array is filled x-outer /
    y-inner and handed to setPixels - which texel ends up with map(x, y) is what these captures pin."""
    if only and only != "spawn_map":
        return
    coef = [0.125, 0.0078125, -0.00048828125, -0.25, -0.001953125, 0.015625]        # exact in fp32
    full, _ = r.spawn_map(16, coef)
    save("mapspawn_full_16", buffers=np.stack(full), coef=np.array(coef), pixels=np.array([16, 16]), offset=np.array([0, 0]))
    part, _ = r.spawn_map(24, coef, pixels=(5, 9), offset=(3, 11))
    save("mapspawn_rect_24", buffers=np.stack(part), coef=np.array(coef), pixels=np.array([5, 9]), offset=np.array([3, 11]))


TIMER_SCRIPT = [
    ["new", 1000, 1000], ["set", "step", 1000 / 60], ["tick", None], ["tick", None], ["tick", None],
    ["set", "paused", True], ["tick", None], ["tick", None], ["set", "paused", False], ["tick", None],
    ["set", "end", 90], ["tick", None], ["tick", None], ["tick", None],                      # runs into `end`, pauses itself
    ["new", 0, 0], ["set", "step", 30], ["set", "end", 100], ["set", "loop", True],
    ["tick", None], ["tick", None], ["tick", None], ["tick", None], ["tick", None],          # wraps
    ["new", 0, 0], ["set", "step", 30], ["set", "rate", -1], ["set", "end", -100], ["tick", None], ["tick", None],
    ["tick", None], ["tick", None], ["tick", None],                                          # negative rate towards a negative end
    ["new", 0, 0], ["set", "step", 25], ["set", "rate", -1.5], ["set", "end", 100], ["set", "loop", True],
    ["tick", None], ["tick", None], ["tick", None],                                          # % keeps the dividend's sign
    ["new", 5000, 5000], ["tick", 5250], ["set", "rate", 2], ["tick", 5300], ["seek", 100], ["tick", 5400],
    ["scrub", 250], ["tick", 5450], ["set", "paused", True], ["tick", 5600], ["tick", 5700], ["set", "paused", False],
    ["tick", 5800], ["reset", 6000, 5900], ["tick", 6100], ["set", "rate", -0.5], ["tick", 6300],
    ["set", "end", -400], ["tick", 6500], ["tick", 7500], ["tick", 7600],                    # wall clock into a negative end
]


def gen_timer(r, only):
    """src/timer.js:24-60 through the reference's own class: fixed step, wall clock, pause absorbing into `offset`,
    end / loop, negative rates, seek / scrub / reset."""
    if only and only != "timer":
        return
    out = r.timer(TIMER_SCRIPT)
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, "timer_script.json")
    with open(path, "w") as f:
        json.dump({"ops": TIMER_SCRIPT, "columns": ["time", "dt", "offset", "since", "paused", "now"],
                   "out": [[repr(float(v)) for v in row] for row in out]}, f, indent=0)
    print("wrote timer_script.json (%d ops)" % len(TIMER_SCRIPT))


# in alphabetical order: the job reaches the page with its keys sorted (kaleido's JSON encoder), and the order of the
# tracks is the order in which a player call visits them - visible in the call log when two tracks fire in one call
ANIMATE_TRACKS = {"calls": [], "colour": [], "spawn": [], "tendrils": []}
ANIMATE_OUTPUTS = {"calls": {}, "colour": [1, 1, 1, 0.5], "spawn": {"radius": 1, "speed": 0},
                   "tendrils": {"autoFade": True, "damping": 0.043, "flowWeight": 1, "noiseScale": 2.125}}
ANIMATE_OPS = [
    # a start frame on every track, like the demo's tracksStart (src/demo.main.js:928-949)
    ["track", "tendrils", "to", {"to": {"flowWeight": 1, "noiseScale": 1.5, "forceWeight": 0.017}, "time": 60}],
    ["track", "colour", "to", {"to": [0, 0, 0, 0.9], "time": 60}],
    ["track", "spawn", "to", {"to": {"radius": 0.6, "speed": 0.1}, "time": 60}],
    ["track", "calls", "to", {"call": ["reset"], "time": 60}],
    ["track", "calls", "to", {"call": ["restart", "dark"], "time": 200}],
    # eased keyframes: smoothTo (ease joined to the previous curve), flipTo, spans with their null start frame
    ["track", "tendrils", "smoothTo", {"to": {"flowWeight": 0.2, "noiseScale": 3.25}, "time": 1000, "ease": [0, 0.95, 1]}],
    ["track", "tendrils", "smoothTo", {"to": {"flowWeight": 0.9, "damping": 0.1}, "time": 1800, "ease": [0, 0.2, 0.7, 1]}],
    ["track", "tendrils", "flipTo", {"to": {"noiseScale": 0.5, "autoFade": False}, "time": 2600, "ease": [0, -0.3, 1.2, 1]}],
    ["track", "tendrils", "smoothOver", 300, {"to": {"flowWeight": 2, "forceWeight": 0.03}, "time": 4000, "ease": [0, 1]}],
    ["track", "tendrils", "easeOver", 250, -0.5, {"to": {"flowWeight": 0.1}, "time": 5000, "ease": [0, 0.1, 0.4, 0.9, 1], "call": ["burst"]}],
    ["track", "colour", "smoothTo", {"to": [1, 0.5, 0.25, 0.1], "time": 1500, "ease": [0, 0.5, 1]}],
    ["track", "colour", "over", 400, {"to": [0.2, 0.2, 0.9, 1], "time": 3000}],
    ["track", "spawn", "flipOver", 500, {"to": {"radius": 0.05, "speed": 0.5}, "time": 2200, "ease": [0, 0.8, 1]}],
    # playback: before the first frame, through spans, skipping several frames at once, exactly on frames, backwards
    ["play", 0], ["play", 60], ["play", 61], ["play", 500], ["play", 1000], ["play", 1250.5], ["play", 1799.999],
    ["play", 2300], ["play", 2600], ["play", 3650], ["play", 3700.25], ["play", 3999], ["play", 4800], ["play", 5000],
    ["play", 6000], ["play", 4900], ["play", 2000], ["seek", 100], ["play", 150], ["playFrom", 2900, 0], ["playFrom", 1100, 4000],
    ["seek", 2750], ["play", 2760],
    # editing while the playhead stands inside the timeline (the demo's `keyframe()` adds a frame at the current track time,
    # src/demo.main.js:1267-1274; <backspace> removes the one before it, :3472-3474), then playing on
    ["track", "tendrils", "smoothTo", {"to": {"flowWeight": 0.5, "damping": 0.2}, "time": 2760, "ease": [0, 0.95, 1]}],
    ["play", 2770], ["play", 2800], ["play", 3690], ["play", 3705],
    ["track", "tendrils", "to", {"to": {"noiseScale": 7}, "time": 3600}], ["play", 3710], ["play", 4100],
    ["track", "colour", "spliceAt", 2000], ["play", 4200], ["playFrom", 1800, 0],
    ["track", "spawn", "spliceSpan", 600, 1500], ["playFrom", 2500, 0],
    ["track", "tendrils", "spliceAt", 4500, 1], ["track", "tendrils", "spliceIndex", 2], ["playFrom", 5200, 0],
    ["track", "colour", "over", 0, {"to": [0.5, 0.5, 0.5, 0.5], "time": 3000}], ["playFrom", 3000, 2900], ["play", 3001],
    ["track", "spawn", "easeTo", 0.25, {"to": {"radius": 2}, "time": 100, "ease": [0, 0.3, 0.6, 0.8, 0.9, 1]}], ["playFrom", 90, 0],
    # return values of the timeline's queries, on the timelines as they stand now
    ["query", "tendrils", "gapAt", 3650], ["query", "tendrils", "gapAt", 60], ["query", "tendrils", "gapAt", -1e9],
    ["query", "tendrils", "gapAt", 1e9], ["query", "tendrils", "indexOf", {"time": 1800}], ["query", "tendrils", "indexOf", {"time": 1e9}],
    ["query", "tendrils", "indexOf", {"time": 59}], ["query", "colour", "spanGapAt", 2800], ["query", "colour", "spanGapAt", 3000],
    ["query", "tendrils", "minFrame", {"to": {"flowWeight": 0.9, "damping": 0.3, "noiseScale": 0.5}, "time": 2000}],
    ["query", "tendrils", "minFrame", {"flowWeight": 1, "time": 7}, 30, [0, 1]], ["query", "tendrils", "minFrame", {"to": {"noiseScale": 7}, "time": 1e9}],
    ["query", "spawn", "start"], ["query", "spawn", "end"], ["query", "spawn", "duration"], ["query", "calls", "valid"],
    ["query", "spawn", "splice", 0, 0], ["query", "spawn", "spliceIndex", 3], ["query", "spawn", "spliceIndex", -1],
    ["query", "spawn", "gapAt", 500], ["playFrom", 2300, 0],
]


def gen_animate(r_unused, only):
    """src/animate (Player, Timeline, tween) through the reference's own compiled classes, taken out of the demo bundle's
    module table: a scripted set of tracks and a play / seek sequence; the outputs after every player call."""
    if only and only != "animate":
        return
    r = RefRunner("demo-modules")
    res = r.animate(ANIMATE_TRACKS, ANIMATE_OPS, ANIMATE_OUTPUTS)
    os.makedirs(GOLDEN, exist_ok=True)
    with open(os.path.join(GOLDEN, "animate_script.json"), "w") as f:
        json.dump({"tracks": ANIMATE_TRACKS, "outputs": ANIMATE_OUTPUTS, "ops": ANIMATE_OPS, "expected": res["out"],
                   "frames": res["frames"], "queries": res["queries"], "player": {k: res[k] for k in ("start", "end", "duration")}}, f)
    print("wrote animate_script.json (%d player calls)" % len(res["out"]))


def random_animate_script(seed, n_ops=48):
    """A seeded random script for the reference's Player: keyframes of every kind added before and DURING playback, removals,
    plays forwards and backwards, seeks, playFroms.  Values are dyadic rationals (exact in binary), times multiples of 1/4."""
    rng = np.random.default_rng(seed)
    keys = ["alpha", "beta", "gamma", "delta"]

    def value():
        return float(rng.integers(-64, 65)) / 16.0

    def frame(track):
        if track == "vec":
            to = [value() for _ in range(4)]
        else:
            to = {k: value() for k in keys if rng.random() < 0.6} or {"alpha": value()}
            if rng.random() < 0.15:
                to["flag"] = bool(rng.random() < 0.5)
        f = {"to": to, "time": float(rng.integers(0, 16000)) / 4.0}
        if rng.random() < 0.7:
            f["ease"] = [0.0] + [float(rng.integers(-8, 25)) / 16.0 for _ in range(int(rng.integers(0, 5)))] + [1.0]
        if rng.random() < 0.2:
            f["call"] = ["c%d" % int(rng.integers(0, 100))]
        return f
    ops, now = [], 0.0
    for k in range(n_ops):
        track = "vec" if rng.random() < 0.3 else "obj"
        r = rng.random()
        if k < 6 or r < 0.35:
            kind = ["to", "smoothTo", "flipTo", "easeTo", "over", "smoothOver", "flipOver", "easeOver"][int(rng.integers(0, 8))]
            args = []
            if kind.endswith("Over") or kind == "over":
                args.append(float(rng.integers(0, 2400)) / 4.0)
            if kind in ("easeTo", "easeOver"):
                args.append(float(rng.integers(-8, 9)) / 8.0)
            ops.append(["track", track, kind] + args + [frame(track)])
        elif r < 0.42:
            ops.append(["track", track, "spliceAt", float(rng.integers(0, 16000)) / 4.0] + ([1] if rng.random() < 0.5 else []))
        elif r < 0.47:
            ops.append(["track", track, "spliceSpan", float(rng.integers(0, 4000)) / 4.0, float(rng.integers(0, 16000)) / 4.0])
        elif r < 0.50:
            ops.append(["track", track, "spliceIndex", int(rng.integers(-3, 8))])
        elif r < 0.55:
            ops.append(["query", track, ["gapAt", "indexOf"][int(rng.integers(0, 2))]] + ([float(rng.integers(0, 16000)) / 4.0] if rng.random() < 2 else []))
            if ops[-1][2] == "indexOf":
                ops[-1][3] = {"time": ops[-1][3]}
        else:
            step = float(rng.integers(1, 3000)) / 4.0
            now = max(0.0, now - step) if rng.random() < 0.2 else now + step
            if r < 0.85:
                ops.append(["play", now])
            elif r < 0.93:
                ops.append(["seek", now])
            else:
                ops.append(["playFrom", now, float(rng.integers(0, 8000)) / 4.0])
    return ops


def gen_animate_fuzz(r_unused, only):
    """Twelve seeded random scripts on the reference's own Player (see random_animate_script): what tests/golden/animate_script.json
    pins by hand, in bulk."""
    if only and only not in ("animate_fuzz", "animate"):
        return
    r = RefRunner("demo-modules")
    cases = []
    for seed in range(12):
        ops = random_animate_script(1000 + seed)
        outputs = {"obj": {"alpha": 0.5, "beta": -1, "flag": True, "label": "x"}, "vec": [0, 0.25, 0.5, 1]}
        res = r.animate({"obj": [], "vec": []}, ops, outputs)
        cases.append({"seed": 1000 + seed, "ops": ops, "outputs": outputs, "expected": res["out"], "queries": res["queries"], "frames": res["frames"]})
    with open(os.path.join(GOLDEN, "animate_fuzz.json"), "w") as f:
        json.dump(cases, f)
    print("wrote animate_fuzz.json (%d scripts, %d player calls)" % (len(cases), sum(len(c["expected"]) for c in cases)))


def scene_colour(proxy, name, preset):
    """The demo's colour proxy (src/demo.main.js:1335-1354): a preset assigns `<name>Color` (0..255) and / or `<name>Alpha`
    to a proxy that keeps everything else, and the state colour is then [r / 255, g / 255, b / 255, alpha]."""
    cp = preset.get("colorProxy", {})
    if name + "Color" in cp:
        proxy[name + "Color"] = list(cp[name + "Color"])
    if name + "Alpha" in cp:
        proxy[name + "Alpha"] = cp[name + "Alpha"]
    return (name + "Color" in cp) or (name + "Alpha" in cp)


SCENES = {
    # name: (first preset, [(preset, frame it is reached at, frames of easing before it (0: from the key before), ease)], seed)
    "scene_flow_turbulence_wings_64": ("Flow", [("Turbulence", 12, 10, [0, 0.95, 1]), ("Wings", 24, 0, [0, 0.2, 1])], 41),
    # autoClearView (the view is wiped every frame), a translucent fade, the colour map on: "Fluid", "Ghostly", "Rorschach"
    "scene_fluid_ghostly_rorschach_64": ("Fluid", [("Ghostly", 9, 6, [0, 0.1, 0.9, 1]), ("Rorschach", 24, 12, [0, 0.95, 1])], 43),
    # `target` > 0 with a targets texture (a ring of positions): "Funhouse" (target 0.005, varyTarget 5) eased in, then "Rave"
    "scene_flow_funhouse_rave_targets_64": ("Flow", [("Funhouse", 10, 8, [0, 0.9, 1]), ("Rave", 24, 10, [0, 0.3, 1])], 47),
}


def gen_scene(r_unused, only):
    """a preset set at once, then keyframes easing into others on the reference's own Player, 24 frames of the
    demo's loop body on the reference's own Tendrils.  Stores the script, the state object after every frame, the particle
    texture of every frame and flow + view of five frames."""
    table = json.load(open(os.path.join(GOLDEN, "presets.json")))
    r = None
    for name, (first_name, keys, seed) in SCENES.items():
        if only and only not in name and only != "scene":
            continue
        n, view, frames, time0, step = 64, (96, 54), 24, 1000.0, 1000 / 60
        rng = np.random.default_rng(seed)
        st = np.zeros((n, n, 4), np.float32)
        st[..., :2] = rng.uniform(-0.95, 0.95, (n, n, 2)) * [1.0, 0.55]
        st[..., 2:] = rng.uniform(-.008, .008, (n, n, 2))
        defaults = {"baseColor": [1, 1, 1, 0.5], "flowColor": [1, 1, 1, 0.04], "fadeColor": [0.1333, 0.1333, 0.1333, 0]}
        proxy = {}
        for c in ("base", "flow", "fade"):
            proxy[c + "Color"] = [v * 255 for v in defaults[c + "Color"][:3]]
            proxy[c + "Alpha"] = defaults[c + "Color"][3]

        def colours(preset, only_named):
            out = {}
            for c in ("base", "flow", "fade"):
                if scene_colour(proxy, c, preset) or not only_named:
                    out[c + "Color"] = [v / 255 for v in proxy[c + "Color"]] + [proxy[c + "Alpha"]]
            return out
        first = table[first_name]
        colors0 = colours(first, False)
        script = [dict(preset=p, time=time0 + at * step, duration=over * step, ease=ease) for p, at, over, ease in keys]
        ops = []
        for k in script:
            p = table[k["preset"]]
            tracks = dict(tendrils=dict(p.get("state", {})), **colours(p, True))
            for tr, to in tracks.items():
                frame = {"to": to, "time": k["time"], "ease": list(k["ease"])}
                ops.append(["track", tr, "smoothOver", k["duration"], frame] if k["duration"] else ["track", tr, "smoothTo", frame])
        r = r or RefRunner("demo-modules")
        grab = [0, 7, 11, 15, 23]
        extra = {}
        if "targets" in name:            # every particle is pulled towards a point of a ring (velocity part unused by logic.frag)
            ang = rng.uniform(0, 2 * np.pi, (n, n))
            tg = np.zeros((n, n, 4), np.float32)
            tg[..., 0], tg[..., 1] = 0.6 * np.cos(ang), 0.3 * np.sin(ang)
            extra["targets"] = tg
        res = r.scene(st, ops, state0=first.get("state", {}), colors0=colors0, time0=time0, frames=frames, view=view, grab=grab, **extra)
        assert res["samples"] == 0
        meta = dict(kind="scene", N=n, frames=frames, viewRes=list(view), viewSize=res["viewSize"], time0=time0, times=res["times"],
                    dts=res["dts"], first=first_name, script=script, ops=ops, colors0=colors0, grab=grab, states=res["states"])
        save(name, state=st, out=res["particles"], flows=np.stack([res["flows"][g] for g in grab]),
             views=np.stack([res["views"][g] for g in grab]), uniforms=json.dumps(meta), **extra)


def _js_object_entries(text):
    """`key: value` pairs of a JS object literal's inside (comments removed), split at top-level commas"""
    import re
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    for part in parts:
        if ":" in part:
            k, v = part.split(":", 1)
            if re.fullmatch(r"[A-Za-z_][A-Za-z0-9_]*", k.strip()):
                yield k.strip(), " ".join(v.split())


def _js_value(expr, names):
    """A literal (number, boolean, array of numbers) or a little arithmetic over known names (`state.flowDecay`,
    `audioDefaults.trackFlowAt`, Math.max / Math.min): -> (value, computed?) or None when it is neither"""
    import ast
    import re
    if re.fullmatch(r"-?[0-9.]+(?:e-?[0-9]+)?|true|false|\[[-0-9., e]*\]", expr):
        return json.loads(expr), False
    py = expr.replace("Math.max", "max").replace("Math.min", "min")
    try:
        tree = ast.parse(py, mode="eval")
    except SyntaxError:
        return None

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return n.value
        if isinstance(n, ast.BinOp) and isinstance(n.op, (ast.Mult, ast.Div, ast.Add, ast.Sub)):
            a, b = ev(n.left), ev(n.right)
            return {ast.Mult: a * b, ast.Add: a + b, ast.Sub: a - b}.get(type(n.op)) if not isinstance(n.op, ast.Div) else a / b
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, ast.USub):
            return -ev(n.operand)
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in ("max", "min"):
            return (max if n.func.id == "max" else min)(*[ev(a) for a in n.args])
        if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name):
            return names[n.value.id][n.attr]
        raise KeyError(ast.dump(n))
    try:
        return ev(tree), True
    except (KeyError, TypeError):
        return None


def gen_presets(r_unused, only):
    """The demo's presets (src/demo.main.js:1483-3238) as DATA: for every preset the numbers / booleans / colour arrays its
    function assigns to `state`, the reset spawner's uniforms, the colour proxy - and, for completeness, to the blur, blend,
    optical-flow and audio-trigger tables the out-of-scope passes read.  A preset runs on top of the defaults
    (wrapPresetter, src/demo.main.js:3244-3264); an entry written as arithmetic over values known here (`Math.max(state.
    flowDecay, 0.1)`, `audioDefaults.trackSpawnAt*0.8`) is evaluated and listed under "computed"; whatever is neither lands
    under "skipped".  Only values travel - no code."""
    if only and only != "presets":
        return
    import re
    src = open("/root/reference/src/demo.main.js").read()
    start = src.index("const presets = {")
    body = src[start:]
    heads = [(m.start(), m.group(1)) for m in re.finditer(r"\n    '([^']+)'\(\) \{", body)]
    end = body.index("\n  };", heads[-1][0])

    def object_after(text, at):
        """the inside of the object literal whose `{` is the first one at or after `at`"""
        a = text.index("{", at)
        depth, k = 0, a
        while True:
            depth += text[k] == "{"
            depth -= text[k] == "}"
            if depth == 0:
                return text[a + 1:k]
            k += 1
    audio_defaults = {}
    for k, v in _js_object_entries(object_after(src, src.index("const audioDefaults = {"))):
        got = _js_value(v, {})
        if got is not None and k not in audio_defaults:
            audio_defaults[k] = got[0]
    # (the microphone's thresholds sit in a conditional spread: the branch taken when `settings.mic_track` is not set)
    for k, v in _js_object_entries(object_after(src, src.index("settings.mic_track !== 'true')?"))):
        got = _js_value(v, {})
        if got is not None and k not in audio_defaults:
            audio_defaults[k] = got[0]
    default_state = {k: _js_value(v, {})[0] for k, v in _js_object_entries(object_after(open("/root/reference/src/index.js").read(),
                     open("/root/reference/src/index.js").read().index("state: {"))) if _js_value(v, {}) is not None}
    targets = (("state", "state"), ("resetSpawner.uniforms", "spawn"), ("colorProxy", "colorProxy"), ("blurState", "blur"),
               ("blendProxy", "blend"), ("opticalFlowState", "opticalFlow"), ("audioState", "audio"))
    out = {}
    for n, (pos, name) in enumerate(heads):
        block = body[pos:(heads[n + 1][0] if n + 1 < len(heads) else end)]
        entry, skipped, computed = {}, [], []
        assigns = []
        for target, key in targets:
            for m in re.finditer(r"Object\.assign\(" + re.escape(target) + r",\s*\{", block):
                assigns.append((m.start(), key, object_after(block, m.end() - 1)))
        names = {"state": dict(default_state), "audioDefaults": audio_defaults}
        for _, key, inside in sorted(assigns):                      # in source order: later entries see earlier state
            for k, v in _js_object_entries(inside):
                got = _js_value(v, names)
                if got is None:
                    skipped.append(key + "." + k)
                    continue
                entry.setdefault(key, {})[k] = got[0]
                if got[1]:
                    computed.append(key + "." + k)
                if key == "state":
                    names["state"][k] = got[0]
        if computed:
            entry["computed"] = computed
        if skipped:
            entry["skipped"] = skipped
        out[name] = entry
    os.makedirs(GOLDEN, exist_ok=True)
    with open(os.path.join(GOLDEN, "presets.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote presets.json (%d presets; %d computed entries, %d skipped)" % (len(out), sum(len(e.get("computed", [])) for e in out.values()),
                                                                             sum(len(e.get("skipped", [])) for e in out.values())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    r = RefRunner()
    print("oracle:", r.probe())
    gen_logic(r, args.only)
    gen_logic_denormal(r, args.only)
    gen_logic_4096(r, args.only)
    gen_logic_config_bands(r, args.only)
    gen_deposit(r, args.only)
    gen_loop(r, args.only)
    gen_optical_flow(r, args.only)
    gen_spawn(r, args.only)
    gen_spawn_image(r, args.only)
    gen_geometry(r, args.only)
    gen_view(r, args.only)
    gen_buffers(r, args.only)
    gen_animate(r, args.only)
    gen_animate_fuzz(r, args.only)
    gen_presets(r, args.only)
    gen_scene(r, args.only)
    gen_spawn_map(r, args.only)
    gen_timer(r, args.only)


if __name__ == "__main__":
    main()
