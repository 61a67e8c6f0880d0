"""Host mirror of the reference's GPGPU core `Particles` (src/particles.js:43-196).

Same constructor options, fields and methods; the FBO ring, the full-screen logic
pass and the spawn upload are backed by the HIP library through the C ABI
(_capi.py).  `logic` is an opaque program object naming the kernel a pass runs
(the reference swaps shader objects: src/index.js:250,435,451).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import call


def defaults():
    """src/particles.js:30-41"""
    return dict(shape=[64, 64], geomShape=None,
                logic=None, logicVert=None, logicFrag=None,
                render=None, renderVert=None, renderFrag=None)


class Program:
    """Stand-in for a compiled gl-shader object: names a kernel family."""

    def __init__(self, kind, **fixed):
        self.kind = kind          # 'logic' | 'spawn-init' | 'spawn-ball' | 'spawn-sample' | 'spawn-direct'
        self.fixed = fixed        # compile-time constants of that shader (e.g. samples, apply)
        self.uniforms = {}

    def bind(self):
        return self


LOGIC = "logic"


class StateBuffer:
    """Stand-in for one gl-fbo of the ring: identity survives ring rotation."""

    def __init__(self, particles, ident):
        self._p = particles
        self.id = ident
        self.shape = list(particles.shape)

    @property
    def index(self):
        return self._p.buffers.index(self)

    def read(self):
        return self._p.read(self)

    def set_pixels(self, pixels, offset=(0, 0)):
        self._p._upload(self.index, pixels, offset)

    def source_index(self):          # as PixelSpawner.buffer / spawnData
        return self.index

    def dispose(self):
        pass


def _as_float(v):
    return float(v)


class Particles:
    def __init__(self, gl=None, options=None):
        params = {**defaults(), **(options or {})}
        self.gl = gl
        self.shape = list(params["shape"])
        self.geomShape = list(params["geomShape"] or self.shape)
        logic = params["logic"] or Program(LOGIC)
        self.logic = logic
        self.render = params["render"]
        self.buffers = []
        # src/particles.js:77-78: host staging, ndarray shape [w, h, 4]
        self.pixels = np.zeros((self.shape[0], self.shape[1], 4), np.float32)
        self._device = int(params.get("device", 0))
        self._mode = int(params.get("mode", _capi.TH_MODE_EXACT))
        self._row0 = int(params.get("row0", 0))
        self._global_height = int(params.get("globalHeight", 0))
        self._next_id = 0
        cfg = _capi.Config(device=self._device, width=self.shape[0], height=self.shape[1],
                           global_height=self._global_height, row0=self._row0, num_buffers=0,
                           mode=self._mode, state_format=int(params.get("stateFormat", 0)))
        self._ctx = C.c_void_p()
        call("th_create", C.byref(cfg), C.byref(self._ctx))

    # -- lifecycle -----------------------------------------------------------------
    def setup(self, numBuffers=1):                  # src/particles.js:81-92
        call("th_setup", self._ctx, int(numBuffers))
        while len(self.buffers) < numBuffers:
            self.buffers.append(StateBuffer(self, self._next_id))
            self._next_id += 1
        while len(self.buffers) > numBuffers:
            self.buffers.pop().dispose()

    def dispose(self):
        if self._ctx:
            call("th_destroy", self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    # -- spawn upload ------------------------------------------------------------------
    def spawn(self, map_fn, pixels=None, offset=(0, 0)):      # src/particles.js:94-117
        pixels = self.pixels if pixels is None else pixels
        w, h = pixels.shape[0], pixels.shape[1]
        data = np.zeros(4, np.float32)
        for x in range(w):
            for y in range(h):
                data[:] = 0
                map_fn(data, x, y)
                pixels[x, y, :] = data
        for k in range(len(self.buffers)):
            self._upload(k, pixels, offset)

    def _upload(self, index, pixels, offset=(0, 0)):
        """pixels: ndarray-convention [w, h, 4] (pixels[x, y]) as in the reference's setPixels."""
        px = np.ascontiguousarray(np.transpose(np.asarray(pixels, np.float32), (1, 0, 2)))
        h, w = px.shape[:2]
        call("th_upload_state", self._ctx, int(index), px.ctypes.data_as(_capi._fp),
             int(offset[0]), int(offset[1]), w, h)

    def upload_texels(self, texels, buffer=-1):
        """texels: [h, w, 4] row-major RGBA32F (readPixels order); buffer -1 = every ring buffer."""
        t = np.ascontiguousarray(texels, np.float32)
        assert t.shape == (self.shape[1], self.shape[0], 4)
        call("th_upload_state", self._ctx, int(buffer), t.ctypes.data_as(_capi._fp), 0, 0,
             self.shape[0], self.shape[1])

    def read(self, buffer=0):
        """readPixels(FLOAT) of a ring buffer -> [h, w, 4]."""
        index = buffer.index if isinstance(buffer, StateBuffer) else int(buffer)
        out = np.empty((self.shape[1], self.shape[0], 4), np.float32)
        call("th_download_state", self._ctx, index, out.ctypes.data_as(_capi._fp), 0, 0,
             self.shape[0], self.shape[1])
        return out

    # -- passes ------------------------------------------------------------------------
    def _target_index(self, buffer):
        if buffer is None:
            return _capi.TH_TARGET_RING
        if isinstance(buffer, StateBuffer):
            return buffer.index
        return buffer.target_index()        # Tendrils.targets texture object

    def step(self, update=None, buffer=None):        # src/particles.js:123-145
        target = self._target_index(buffer)
        uniforms = Particles.applyUpdate(
            dict(self.logic.uniforms, dataRes=self.shape, geomRes=self.geomShape), update)
        self.logic.uniforms = uniforms
        run_pass(self, self.logic, uniforms, target)
        if buffer is None:
            # utils.step(this.buffers): pop -> unshift  (src/utils/index.js:1-7); the C side did the same
            self.buffers.insert(0, self.buffers.pop())

    def step_n(self, update, time0, dt_ms, n):
        """n consecutive logic passes with a fixed-step timer (time_k = time0 + (k+1)*dt_ms, accumulated in
        double like src/timer.js:28-31), replayed from a captured hipGraph.  Extension: the reference
        issues these one draw call at a time."""
        uniforms = Particles.applyUpdate(
            dict(self.logic.uniforms, dataRes=self.shape, geomRes=self.geomShape), update)
        self.logic.uniforms = uniforms
        if self.logic.kind != LOGIC:
            raise ValueError("step_n runs the logic program only")
        s = logic_uniforms(uniforms)
        call("th_step_n", self._ctx, C.byref(s), C.c_double(time0), C.c_double(dt_ms), int(n))
        for _ in range(int(n) % max(len(self.buffers), 1)):
            self.buffers.insert(0, self.buffers.pop())

    def draw(self, update=None, mode=None):          # src/particles.js:147-158 - no display here
        return None

    def updateLogic(self, logicFrag):
        self.logic = logicFrag if isinstance(logicFrag, Program) else Program(LOGIC)

    def updateRender(self, *a):
        pass

    def sync(self):
        call("th_sync", self._ctx)

    def draw_pipeline(self, which):
        """Which of the library's two draw() pipelines runs (build-defined, the results are the same): "auto", "stream"
        (texel order, stream-ordered stable sort) or "bins" (any slot order, per-bin ordering)."""
        call("th_draw_pipeline", self._ctx, {"auto": -1, "stream": 0, "bins": 1}[which])

    OPTIONS = dict(bucket=0, resort_steps=1, rebucket_steps=2, fuse=3, graph=4, force_generic=5, draw_reuse=6, bins_pool=7,
                   inject_failure=8, bins_pages=9, async_sort=10, skip_unseen=11)

    def option(self, name, value=None):
        """A switch between equivalent paths of the library (th_option_set / _get; no switch changes a result): returns the
        current value, sets `value` first if given."""
        if value is not None:
            call("th_option_set", self._ctx, self.OPTIONS[name], int(value))
        out = C.c_int64(0)
        call("th_option_get", self._ctx, self.OPTIONS[name], C.byref(out))
        return out.value

    def deposit_flow(self, view_size, time, speed_limit):
        """The flow pass of Tendrils.draw(): (previous -> current) lines blended into the flow texture
        (src/index.js:295-303, src/particles.js:147-158).  Returns the number of fragments."""
        u = _capi.DepositUniforms(time=float(time), speedLimit=float(speed_limit))
        u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
        n = C.c_uint64(0)
        call("th_flow_deposit", self._ctx, C.byref(u), C.byref(n))
        return int(n.value)

    def export_lines(self, view_size, time, speed_limit):
        """The line list of draw() ([n, 12] float32: p0.xy, p1.xy, c0, c1 in stream order) for offline rendering."""
        u = _capi.DepositUniforms(time=float(time), speedLimit=float(speed_limit))
        u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
        n = C.c_uint64(0)
        call("th_export_lines", self._ctx, C.byref(u), None, 0, C.byref(n))
        out = np.empty((int(n.value), 12), np.float32)
        if n.value:
            call("th_export_lines", self._ctx, C.byref(u), out.ctypes.data_as(_capi._fp), n.value, C.byref(n))
        return out

    def stats(self, speed_limit):
        c = _capi.Counters()
        call("th_stats", self._ctx, C.c_float(speed_limit), C.byref(c))
        return {k: getattr(c, k) for k, _ in _capi.Counters._fields_}

    @staticmethod
    def generateLUT(shape):                          # src/particles.js:171-190
        w, h = max(shape[0], 2), max(shape[1], 2)
        inv_x, inv_y = 1 / (w - 1), 1 / (h - 1)
        data = np.zeros(shape[0] * shape[1] * 2, np.float32)
        k = 0
        for i in range(w):
            for j in range(h):
                if k + 1 < data.size:
                    data[k] = i * inv_x
                    data[k + 1] = j * inv_y
                k += 2
        return data

    @staticmethod
    def applyUpdate(state, update):                  # src/particles.js:192-195
        if callable(update):
            return update(state)
        state.update(update or {})
        return state


def logic_uniforms(u):
    """dict of reference uniform names -> C struct (double -> fp32 as gl.uniform1f does)."""
    s = _capi.LogicUniforms()
    vs = u.get("viewSize", (1.0, 1.0))
    s.viewSize[0], s.viewSize[1] = float(vs[0]), float(vs[1])
    for name, _ in _capi.LogicUniforms._fields_:
        if name == "viewSize":
            continue
        setattr(s, name, _as_float(u.get(name, 0.0)))
    return s


def run_pass(particles, program, uniforms, target):
    """One full-screen pass of `program` into `target` (screen.render(), src/particles.js:143)."""
    ctx = particles._ctx
    kind = program.kind
    if kind == LOGIC:
        s = logic_uniforms(uniforms)
        call("th_step", ctx, C.byref(s), target)
    elif kind == "spawn-init":
        call("th_spawn_init", ctx, target)
    elif kind == "spawn-ball":
        s = _capi.SpawnBallUniforms(radius=float(uniforms.get("radius", 1)), speed=float(uniforms.get("speed", 0)))
        call("th_spawn_ball", ctx, C.byref(s), target)
    elif kind in ("spawn-sample", "spawn-direct"):
        s = _capi.SpawnSampleUniforms()
        for name in ("spawnSize", "jitter"):
            v = uniforms.get(name, (1.0, 1.0))
            getattr(s, name)[0], getattr(s, name)[1] = float(v[0]), float(v[1])
        for name in ("time", "speed", "bias", "flowDecay"):
            setattr(s, name, float(uniforms.get(name, 0.0)))
        m = uniforms.get("spawnMatrix", (1, 0, 0, 0, 1, 0, 0, 0, 1))
        for k in range(9):
            s.spawnMatrix[k] = float(m[k])
        s.samples = int(program.fixed.get("samples", 0))
        s.apply = int(program.fixed.get("apply", 2))
        src = uniforms.get("spawnData")
        if hasattr(src, "bind_for"):                      # the spawner's own image buffer: uploaded on use
            src.bind_for(particles)
        source = src if isinstance(src, int) else src.source_index()
        if source >= 0 and target == _capi.TH_TARGET_RING:
            # the C side resolves ring indices AFTER utils.step() rotated the ring (the order the
            # pass sees); `source` was taken from the pre-rotation list
            source = (source + 1) % len(particles.buffers)
        call("th_spawn_sample" if kind == "spawn-sample" else "th_spawn_direct", ctx, C.byref(s), source, target)
    else:
        raise ValueError("unknown program kind %r" % (kind,))


default = Particles
