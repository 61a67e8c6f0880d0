"""TEST INFRASTRUCTURE - runs the reference itself (build container only).

Drives the reference's own prebuilt bundle (/root/reference/docs/js/index.js: UMD
`Tendrils`, compiled logic.frag inlined) and the compiled shader strings of
/root/reference/docs/js/demo.js inside kaleido's headless Chromium + SwiftShader
(software WebGL 1, fp32 highp).  Used only by oracle/gen_fixtures.py to produce
the numeric golden vectors under tests/golden/.  The reference text is read at
run time from /root/reference, concatenated with oracle/harness.js in /tmp and
never written into this repository.  Not importable on the GPU box (no
/root/reference there) - nothing in tests/, bench.py or the product imports it.
"""
import base64
import json
import os
import re
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _b64(a, dtype):
    return base64.b64encode(np.ascontiguousarray(a, dtype=dtype).tobytes()).decode()


def _f32(s, shape):
    return np.frombuffer(base64.b64decode(s), dtype=np.float32).reshape(shape).copy()


class RefRunner:
    def __init__(self, bundle="index"):
        """bundle "index": docs/js/index.js (the Tendrils library) + harness.js; "demo-modules": docs/js/demo.js with its
        bootstrap handing out the module loader instead of starting the app (one token of the webpack prologue changed in
        the /tmp copy) + harness_animate.js - gives access to the compiled animation classes."""
        from kaleido.scopes.plotly import PlotlyScope

        self._tmp = tempfile.mkdtemp(prefix="tendrils_oracle_")
        stub = os.path.join(self._tmp, "stub.js")
        with open(os.path.join(REF, "docs/js/index.js" if bundle == "index" else "docs/js/demo.js")) as f:
            bundle_text = f.read()
        if bundle != "index":
            boot = 't.p="",t(0)}(['
            assert bundle_text.find(boot) == 405            # the outer bundle's prologue (a nested copy further in stays as it is)
            bundle_text = bundle_text.replace(boot, 't.p="",t}([', 1)
        with open(os.path.join(HERE, "harness.js" if bundle == "index" else "harness_animate.js")) as f:
            harness = f.read()
        bundle = bundle_text
        with open(stub, "w") as f:
            f.write(bundle + "\n" + harness)
        self._scope = PlotlyScope(plotlyjs="file://" + stub)
        self._demo_shaders = None

    def _run(self, job):
        raw = self._scope.transform({"data": [], "layout": {"job": job}}, format="svg")
        res = json.loads(raw.decode())
        if "error" in res:
            raise RuntimeError("oracle harness: %s\n%s" % (res["error"], res.get("stack")))
        return res

    def probe(self):
        return self._run({"kind": "probe"})

    # -- reference Tendrils.step() ------------------------------------------------
    def logic(self, state, flow=None, targets=None, uniforms=None, time0=0.0, steps=1,
              view=(64, 64), flow_shape=None, view_size=None, return_each=False, rows=None, state_gen=None, draw=False):
        """state: [N,N,4] f32 indexed [y][x][c]; flow: [H,W,4]; returns list of [N,N,4]."""
        if state_gen is not None:            # {"N":..., "seed":..., "inertMod":...}: generated inside the page
            N = int(state_gen["N"])
            inputs = {"stateGen": {"seed": int(state_gen["seed"]), "inertMod": int(state_gen.get("inertMod", 0))}}
        else:
            N = state.shape[0]
            assert state.shape == (N, N, 4)
            inputs = {"state": _b64(state, np.float32)}
        job = {"kind": "logic", "N": N, "viewW": int(view[0]), "viewH": int(view[1]),
               "state": uniforms or {}, "time0": float(time0), "steps": int(steps),
               "returnEach": bool(return_each), "inputs": inputs, "draw": bool(draw)}
        if flow is not None:
            fh, fw = flow.shape[:2]
            if flow_shape is None:
                flow_shape = (fw, fh)
            assert (fw, fh) == tuple(flow_shape)
            job["inputs"]["flow"] = _b64(flow, np.float32)
        if flow_shape is not None:
            job["flowW"], job["flowH"] = int(flow_shape[0]), int(flow_shape[1])
        if targets is not None:
            job["inputs"]["targets"] = _b64(targets, np.float32)
        if view_size is not None:
            job["viewSize"] = [float(view_size[0]), float(view_size[1])]
        if rows is not None:
            job["rows"] = [[int(a), int(b)] for a, b in rows]
        res = self._run(job)
        if res.get("err"):
            raise RuntimeError("GL error %s" % res["err"])
        if rows is not None:
            outs = [[_f32(b, (r[1] - r[0], N, 4)) for b, r in zip(bands, rows)] for bands in res["out"]]
        else:
            outs = [_f32(o, (N, N, 4)) for o in res["out"]]
        if res.get("flowOut"):
            fw, fh = res["flowShape"]
            res["flow_out"] = _f32(res["flowOut"], (fh, fw, 4))
        return outs, res

    # -- reference Tendrils.draw(): flow deposit ------------------------------------
    def deposit(self, current, previous, flow=None, uniforms=None, time=0.0, view=(64, 64), view_size=None, want_view=False):
        """current/previous: [N,N,4] state textures; returns the flow FBO [H,W,4] after draw() (want_view: and, in
        res["view_out"], the view render [H,W,4] uint8 of the same draw() on a context without multisampling)."""
        N = current.shape[0]
        job = {"kind": "deposit", "N": N, "viewW": int(view[0]), "viewH": int(view[1]), "state": uniforms or {},
               "time": float(time), "view": bool(want_view),
               "inputs": {"current": _b64(current, np.float32), "previous": _b64(previous, np.float32)}}
        if flow is not None:
            job["inputs"]["flow"] = _b64(flow, np.float32)
        if view_size is not None:
            job["viewSize"] = [float(view_size[0]), float(view_size[1])]
        res = self._run(job)
        if res.get("err"):
            raise RuntimeError("GL error %s" % res["err"])
        w, h = res["flowShape"]
        if res.get("view"):
            res["view_out"] = np.frombuffer(base64.b64decode(res["view"]), dtype=np.uint8).reshape(int(view[1]), int(view[0]), 4).copy()
        return _f32(res["out"], (h, w, 4)), res

    def buffers(self, current, previous, ops, num_buffers=1, uniforms=None, time=0.0, view=(64, 64)):
        """A script on the reference's own Tendrils.buffers surface (src/index.js:172-184, 318-325, 359-391), on a context
        without multisampling: returns the RGBA8 images the script's 'read' ops took, in order, and the ring's length after
        every op."""
        job = {"kind": "buffers", "N": current.shape[0], "viewW": int(view[0]), "viewH": int(view[1]), "numBuffers": int(num_buffers),
               "state": uniforms or {}, "time": float(time), "ops": ops,
               "inputs": {"current": _b64(current, np.float32), "previous": _b64(previous, np.float32)}}
        res = self._run(job)
        if res.get("err"):
            raise RuntimeError("GL error %s" % res["err"])
        res["images"] = [np.frombuffer(base64.b64decode(b), dtype=np.uint8).reshape(int(view[1]), int(view[0]), 4).copy() for b in res["reads"]]
        return res

    # -- the reference's Player / Timeline classes, scripted (bundle="demo-modules") ---------------
    def animate(self, tracks, ops, outputs=None):
        return self._run({"kind": "animate", "tracks": tracks, "ops": ops, "outputs": outputs or {}})

    def scene(self, particles, ops, state0=None, colors0=None, time0=0.0, frames=24, view=(96, 54), grab=(), targets=None):
        """bundle="demo-modules": the reference's Player (tracks tendrils / baseColor / flowColor / fadeColor writing into
        tendrils.state, as src/demo.main.js:836-857) keyframed by `ops`, then `frames` x [timer.tick(); player.play(time);
        step(); draw()] of the reference's Tendrils.  Returns per frame: time, dt, the whole state object, the particle
        texture; flow [H,W,4] f32 and view [H,W,4] u8 for the frames in `grab`."""
        n = particles.shape[0]
        res = self._run({"kind": "scene", "N": n, "viewW": int(view[0]), "viewH": int(view[1]), "state0": state0 or {},
                         "colors0": colors0 or {}, "particles": _b64(particles, np.float32), "time0": float(time0),
                         "frames": int(frames), "ops": ops, "grab": [int(g) for g in grab],
                         **({"targets": _b64(targets, np.float32)} if targets is not None else {})})
        if res.get("err"):
            raise RuntimeError("GL error %s" % res["err"])
        fw, fh = res["flowShape"]
        res["particles"] = np.stack([_f32(p, (n, n, 4)) for p in res["particles"]])
        res["flows"] = {int(k): _f32(v, (fh, fw, 4)) for k, v in res["flows"].items()}
        res["views"] = {int(k): np.frombuffer(base64.b64decode(v), dtype=np.uint8).reshape(int(view[1]), int(view[0]), 4).copy()
                        for k, v in res["views"].items()}
        return res

    # -- reference Particles.spawn(map, pixels, offset) ------------------------------
    def spawn_map(self, n, coef, pixels=None, offset=None, view=(32, 32)):
        """Returns the ring buffers ([N,N,4] each) after particles.spawn(map, pixels, offset) with
        map(data,x,y) = (c0+c1*x+c2*y, c3+c4*x+c5*y, x, y)."""
        job = {"kind": "spawn_map", "N": int(n), "viewW": int(view[0]), "viewH": int(view[1]), "coef": [float(v) for v in coef]}
        if pixels is not None:
            job["pixels"] = [int(pixels[0]), int(pixels[1])]
            job["offset"] = [int(offset[0]), int(offset[1])] if offset is not None else [0, 0]
        res = self._run(job)
        if res.get("err"):
            raise RuntimeError("GL error %s" % res["err"])
        return [_f32(o, (n, n, 4)) for o in res["out"]], res

    # -- the reference's Timer class, scripted ---------------------------------------
    def timer(self, ops):
        """ops: list of ['new', now, since] | ['set', field, value] | ['tick', now|None] | ['seek', to] | ['scrub', by] |
        ['reset', now, since]; returns [len(ops), 6] float64: time, dt, offset, since, paused, now(..) after every op."""
        res = self._run({"kind": "timer", "ops": ops})
        return np.array(res["out"], dtype=np.float64)

    # -- compiled shader strings of demo.js --------------------------------------
    def demo_shaders(self):
        if self._demo_shaders is None:
            with open(os.path.join(REF, "docs/js/demo.js")) as f:
                s = f.read()
            lits = re.findall(r'"((?:[^"\\]|\\.)*?GLSLIFY(?:[^"\\]|\\.)*?)"', s)
            lits = [t.encode().decode("unicode_escape") for t in lits]
            out = {}
            for t in lits:
                if "attribute vec2 position" in t and "uv = position" in t and "viewSize" not in t:
                    out.setdefault("screen_vert", t)
                elif "uniform float radius;" in t and "randoms" in t:
                    out["spawn_ball"] = t
                elif "uniform sampler2D last;" in t and "gradMag" in t:
                    out["optical_flow"] = t
                elif "gl_Position = vec4(position*viewSize, 0.0, 1.0)" in t:
                    out["geometry_vert"] = t              # src/geom/vert/index.vert
                elif "gl_FragColor = color;" in t and "uniform vec4 color" in t and "varying" not in t:
                    out["geometry_frag"] = t              # src/geom/frag/index.frag
                elif "spawnData" in t and "luma" in t and "const float samples = 6.0" in t:
                    out["spawn_bright_sample"] = t        # bright-sample.frag (GeometrySpawner)
                elif "spawnData" in t and "rgb2hsv" in t and "const float samples = 6.0" in t:
                    out["spawn_best_sample"] = t          # best-sample.frag: colour apply, vignette
                elif "spawnData" in t and "rgb2hsv" in t and "samples" not in t:
                    out["spawn_direct"] = t               # index.frag (direct-main)
                elif "spawnData" in t and "const float samples = 5.0" in t and "flowDecay" in t:
                    out["spawn_flow_sample"] = t
                elif "spawnData" in t and "const float samples = 2.0" in t:
                    out["spawn_data_sample"] = t
                elif "const vec2 pos = vec2(inert)" in t or ("vec4(pos, vel)" in t and "inert" in t):
                    out["spawn_init"] = t
            self._demo_shaders = out
        return self._demo_shaders

    def shader(self, frag, out_shape, textures=None, uniforms=None, blend=False, dst=None, vert="screen_vert",
               positions=None):
        """One pass of compiled reference shader `frag` (name in demo_shaders()): the big triangle, or the
        triangles in `positions` (flat xy list) with vertex shader `vert`."""
        sh = self.demo_shaders()
        w, h = out_shape
        tex = {}
        for name, arr in (textures or {}).items():
            th, tw = arr.shape[:2]
            if arr.dtype == np.uint8:
                tex[name] = {"type": "u8", "w": tw, "h": th, "data": _b64(arr, np.uint8)}
            else:
                tex[name] = {"type": "f32", "w": tw, "h": th, "data": _b64(arr, np.float32)}
        job = {"kind": "shader", "vert": sh[vert], "frag": sh[frag],
               "outW": int(w), "outH": int(h), "textures": tex, "uniforms": uniforms or {},
               "blend": bool(blend)}
        if dst is not None:
            job["dst"] = _b64(dst, np.float32)
        if positions is not None:
            job["positions"] = [float(v) for v in positions]
        res = self._run(job)
        if res.get("err"):
            raise RuntimeError("GL error %s" % res["err"])
        return _f32(res["out"], (h, w, 4))
