"""Host mirror of the reference's simulation facade `Tendrils` (src/index.js:84-457) for the
particle-update path: state uniforms, timer, flow/targets textures, step(), spawn(),
spawnShader(), resize(), and draw() = the flow pass + the view pass (the particles' lines with the
render shader's colours into an RGBA8 image of the drawing buffer: `view`, read with read_view()).
The multi-buffer screen passes (blur / copy shaders, src/screen) are outside this build.
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import call
from .particles import LOGIC, Particles, Program, run_pass
from .timer import Timer


def defaults():
    """src/index.js:28-75"""
    timer = Timer()
    timer.step = 1000 / 60
    return dict(
        state=dict(
            rootNum=2 ** 9,
            autoClearView=False, autoFade=True,
            damping=0.043, speedLimit=0.01,
            forceWeight=0.016, varyForce=-0.1,
            flowWeight=1, varyFlow=0.2,
            noiseWeight=0.002, varyNoise=0.3,
            flowDecay=0.005, flowWidth=5,
            noiseScale=2.125, varyNoiseScale=0.5,
            noiseSpeed=0.00025, varyNoiseSpeed=0.1,
            target=0, varyTarget=1,
            lineWidth=1, speedAlpha=0.000001, colorMapAlpha=0.4,
            baseColor=[1, 1, 1, 0.5], flowColor=[1, 1, 1, 0.04], fadeColor=[0.1333, 0.1333, 0.1333, 0]),
        timer=timer, numBuffers=0, logicShader=None, colorMap=None)


gl_settings = dict(preserveDrawingBuffer=True, antialias=True)     # src/index.js:77-80


def cover_aspect(size):
    """src/utils/aspect.js:4-11: scale(inverse(size), max(size))"""
    m = max(size[0], size[1])
    return [m / size[0], m / size[1]]


class View:
    """What the reference reads from its WebGL context: the drawing-buffer size."""

    def __init__(self, width, height):
        self.drawingBufferWidth = int(width)
        self.drawingBufferHeight = int(height)


class FlowTexture:
    """tendrils.flow (src/index.js:102): RGBA32F, NEAREST, CLAMP_TO_EDGE, resizable."""

    def __init__(self, owner):
        self._o = owner
        self._shape = [1, 1]

    @property
    def shape(self):
        return list(self._shape)

    @shape.setter
    def shape(self, wh):
        self._shape = [int(wh[0]), int(wh[1])]
        if self._o.particles is not None:
            call("th_flow_resize", self._o.particles._ctx, self._shape[0], self._shape[1])

    def set_pixels(self, texels):
        t = np.ascontiguousarray(texels, np.float32)
        assert t.shape == (self._shape[1], self._shape[0], 4), (t.shape, self._shape)
        call("th_flow_upload", self._o.particles._ctx, t.ctypes.data_as(_capi._fp))

    def read(self):
        out = np.empty((self._shape[1], self._shape[0], 4), np.float32)
        call("th_flow_download", self._o.particles._ctx, out.ctypes.data_as(_capi._fp))
        return out

    def clear(self):
        call("th_flow_clear", self._o.particles._ctx)

    def source_index(self):
        return _capi.TH_SOURCE_FLOW


class ColorMap:
    """tendrils.colorMap (src/index.js:94-96): RGBA32F, a 1x1 zero texture until set."""

    def __init__(self, owner):
        self._o = owner
        self.shape = [1, 1]
        self._pixels = None

    def set_pixels(self, texels):
        t = np.ascontiguousarray(texels, np.float32)
        assert t.ndim == 3 and t.shape[2] == 4
        self.shape = [t.shape[1], t.shape[0]]
        self._pixels = t
        self.bind()

    def bind(self):
        if self._pixels is not None and self._o.particles is not None:
            call("th_colormap_upload", self._o.particles._ctx, self._pixels.ctypes.data_as(_capi._fp),
                 self.shape[0], self.shape[1])


class TargetsTexture:
    """tendrils.targets (src/index.js:105,207): RGBA32F at the particle shape."""

    def __init__(self, owner):
        self._o = owner
        self.shape = [1, 1]

    def set_pixels(self, texels):
        t = np.ascontiguousarray(texels, np.float32)
        call("th_targets_upload", self._o.particles._ctx, t.ctypes.data_as(_capi._fp))

    def read(self):
        p = self._o.particles
        out = np.empty((p.shape[1], p.shape[0], 4), np.float32)
        call("th_targets_download", p._ctx, out.ctypes.data_as(_capi._fp))
        return out

    def clear(self):
        call("th_targets_clear", self._o.particles._ctx)

    def target_index(self):
        return _capi.TH_TARGET_TARGETS


def init_spawner(data, x=0, y=0):
    """src/spawn/init/cpu.js:3-8"""
    data[0] = data[1] = _capi.INERT
    data[2] = data[3] = 0
    return data


class ViewBuffer:
    """One of Tendrils.buffers (src/index.js:172-177: `FBO(gl, [1, 1])`, given viewRes by resize()): an off-screen RGBA8 view
    image on the device, addressed by its place in the owner's ring."""

    def __init__(self, owner):
        self._o = owner
        self.shape = [1, 1]

    def bind(self):
        self._o._bind_view(self)
        return self

    def read(self):
        """[viewRes.y, viewRes.x, 4] uint8"""
        return self._o.read_view(self)

    def dispose(self):
        self._o = None


class Tendrils:
    def __init__(self, gl=None, options=None):
        params = {**defaults(), **(options or {})}
        self.gl = gl if gl is not None else View(1, 1)
        self.state = params["state"]
        self.particles = None
        self.flow = FlowTexture(self)
        self.targets = TargetsTexture(self)
        self.colorMap = params.get("colorMap") or ColorMap(self)
        self.renderView = bool(params.get("renderView", True))     # draw() also runs the view pass (as the reference's does)
        self.buffers = []
        self.logicShader = None
        self.uniforms = dict(render={}, update={})
        self.viewRes = [0, 0]
        self.viewSize = [0, 0]
        self.timer = params["timer"]
        self._device = int(params.get("device", 0))
        self._mode = int(params.get("mode", _capi.TH_MODE_EXACT))
        self._state_format = int(params.get("stateFormat", _capi.TH_STATE_F32))
        self._band = (int(params.get("row0", 0)), params.get("rows"), int(params.get("globalHeight", 0)))
        self.dist = params.get("dist")           # torch.distributed module of a row-band-sharded job (draw() exchanges)
        # what gl.getParameter(gl.ALIASED_LINE_WIDTH_RANGE) reports here: [1, 1] like the GL the reference was captured on
        # (flowWidth: 5 then draws width-1 lines, as it does there); up to [1, 64] for the picture of a GL that honours widths
        self.lineWidthRange = tuple(params.get("lineWidthRange", (1, 1)))
        self._bound = None                       # the bound view image: None = the screen, else one of self.buffers
        self.setupBuffers(int(params.get("numBuffers", 0) or 0))      # src/index.js:109

    # -- setup ---------------------------------------------------------------------
    def setup(self, *rest):                                   # src/index.js:149-154
        self.setupParticles(*rest)
        self.reset()
        return self

    def reset(self):
        self.spawn()
        return self

    def dispose(self):
        if self.particles is not None:
            self.particles.dispose()
            self.particles = None
        return self

    # -- Tendrils.buffers: off-screen view images (src/index.js:172-184, 359-391) ----------------------------------------
    def setupBuffers(self, numBuffers=0):                      # src/index.js:172-184
        while len(self.buffers) < numBuffers:
            self.buffers.append(ViewBuffer(self))
        while len(self.buffers) > numBuffers:
            gone = self.buffers.pop()
            if self._bound is gone:
                self._bound = None                             # (the library leaves the screen bound as well)
            gone.dispose()
        if self.particles is not None:
            call("th_view_buffers", self.particles._ctx, len(self.buffers))
        return self

    def _bind_view(self, buffer=None):
        """gl.bindFramebuffer: None = the screen, else one of self.buffers"""
        self._bound = buffer
        if self.particles is not None:
            call("th_view_bind", self.particles._ctx, -1 if buffer is None else self.buffers.index(buffer))

    def drawBuffer(self, index=None):                          # src/index.js:359-367: a buffer's contents to the screen
        self._bind_view(None)
        if self.state["autoClearView"]:
            call("th_view_clear", self.particles._ctx)         # gl.clear of the bound framebuffer - the screen - alone
        return self.copyBuffer(0 if index is None else index).stepBuffers()      # (`undefined` takes copyBuffer's default)

    def copyBuffer(self, index=0):                             # src/index.js:370-383: into the current render target
        if 0 <= index < len(self.buffers):
            call("th_view_copy", self.particles._ctx, int(index))
        return self

    def stepBuffers(self):                                     # src/index.js:385-391
        if len(self.buffers) > 1:
            self.buffers.insert(0, self.buffers.pop())         # src/utils/index.js:1-7
            if self.particles is not None:
                call("th_view_step_buffers", self.particles._ctx)
        return self

    def viewport(self):                                        # src/index.js:410-419: gl.viewport(0, 0, ...viewRes) - every pass
        return self                                            # here covers its whole target already

    def setupParticles(self, rootNum=None, numBuffers=2):     # src/index.js:186-210
        rootNum = self.state["rootNum"] if rootNum is None else rootNum
        self.state["rootNum"] = rootNum
        row0, rows, gh = self._band
        shape = [rootNum, rows if rows else rootNum]
        if self.particles is not None:
            self.particles.dispose()
        self.particles = Particles(self.gl, dict(
            shape=shape, geomShape=[shape[0], shape[1] * 2], logic=Program(LOGIC),
            device=self._device, mode=self._mode, stateFormat=self._state_format, row0=row0,
            globalHeight=(gh if gh else (rootNum if rows else 0))))
        self.logicShader = self.particles.logic
        self.particles.setup(numBuffers)
        self.targets.shape = shape
        self.flow.shape = self.flow.shape          # (re)create on the new context
        if isinstance(self.colorMap, ColorMap):
            self.colorMap._o = self
            self.colorMap.bind()
        call("th_line_width_range", self.particles._ctx, float(self.lineWidthRange[0]), float(self.lineWidthRange[1]))
        self._bound = None                       # (a new context: its screen is bound, its ring is empty)
        if self.buffers:
            call("th_view_buffers", self.particles._ctx, len(self.buffers))
        return self

    def line_widths(self):
        """gl.lineWidth(Math.max(0, flowWidth)) before the flow pass, gl.lineWidth(Math.max(0, lineWidth)) before the view
        pass (src/index.js:302,336); a width of 0 is GL's INVALID_VALUE: the width of that pass stays what it was."""
        flow, view = max(0.0, float(self.state["flowWidth"])), max(0.0, float(self.state["lineWidth"]))
        if flow > 0:
            call("th_line_width", self.particles._ctx, _capi.TH_PASS_FLOW, flow)
        if view > 0:
            call("th_line_width", self.particles._ctx, _capi.TH_PASS_VIEW, view)

    # -- clears ---------------------------------------------------------------------
    def clear(self):
        self.clearView()
        self.clearFlow()
        return self

    def clearView(self):                                       # src/index.js:220-229: every buffer, then the screen -
        for b in self.buffers:                                 # which it leaves bound
            self._bind_view(b)
            call("th_view_clear", self.particles._ctx)
        self._bind_view(None)
        call("th_view_clear", self.particles._ctx)
        return self

    def drawFade(self):                                        # src/index.js:342-348
        if self.state["fadeColor"][3] > 0:
            self.drawFill(self.state["fadeColor"])
        return self

    def drawFill(self, color=None):                            # src/index.js:350-356
        col = (C.c_float * 4)(*[float(v) for v in (self.state["fadeColor"] if color is None else color)])
        call("th_view_fill", self.particles._ctx, col)
        return self

    def render_uniforms(self):
        """The view pass's uniforms (src/index.js:284-293 over `state`); sin(time*flowDecay) is evaluated here - as the
        shader would, in fp32 operands - because GLSL leaves its value to the implementation."""
        s = self.state
        u = _capi.RenderUniforms(time=float(self.timer.time), speedLimit=float(s["speedLimit"]), flowDecay=float(s["flowDecay"]),
                                 speedAlpha=float(s["speedAlpha"]), colorMapAlpha=float(s["colorMapAlpha"]))
        u.sinTerm = math.sin(float(np.float32(self.timer.time)) * float(np.float32(s["flowDecay"])))
        u.viewSize[0], u.viewSize[1] = float(self.viewSize[0]), float(self.viewSize[1])
        for k in range(4):
            u.baseColor[k], u.flowColor[k] = float(s["baseColor"][k]), float(s["flowColor"][k])
        return u

    def read_view(self, buffer=None):
        """The screen image - or one of self.buffers -: [viewRes.y, viewRes.x, 4] uint8 (readPixels order).  What is bound
        stays bound."""
        out = np.empty((self.viewRes[1], self.viewRes[0], 4), np.uint8)
        was = self._bound
        if was is not buffer:
            self._bind_view(buffer)
        call("th_view_download", self.particles._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        if was is not buffer:
            self._bind_view(was)
        return out

    def clearFlow(self):                                       # src/index.js:231-236
        self.flow.clear()
        return self

    def restart(self):
        self.clear()
        self.reset()
        return self

    # -- the hot path ------------------------------------------------------------------
    def step(self):                                            # src/index.js:248-272
        if not self.timer.paused:
            self.particles.logic = self.logicShader
            self.uniforms["update"].update(self.state)
            self.uniforms["update"].update(
                dt=self.timer.dt, time=self.timer.time, start=self.timer.since,
                flow=self.flow, targets=self.targets,
                viewSize=self.viewSize, viewRes=self.viewRes)
            self.particles.step(self.uniforms["update"])
        return self

    def step_n(self, n):
        """n x (timer.tick(); step()) for a fixed-step, unpaused timer, as one captured-graph replay."""
        tm = self.timer
        if tm.paused or tm.step < 0 or tm.end >= 0:
            for _ in range(n):
                tm.tick()
                self.step()
            return self
        self.particles.logic = self.logicShader
        dt = tm.step * tm.rate
        self.uniforms["update"].update(self.state)
        self.uniforms["update"].update(dt=dt, time=tm.time, start=tm.since, flow=self.flow, targets=self.targets,
                                       viewSize=self.viewSize, viewRes=self.viewRes)
        self.particles.step_n(self.uniforms["update"], tm.time, dt, n)
        for _ in range(n):
            tm.tick()
        return self

    def draw(self):                                            # src/index.js:278-340
        """The flow pass - particle lines into the flow texture, so that particles respond to each other's wake - and,
        with renderView, the view pass: the same lines into the RGBA8 view buffer (after the clear / fade the state
        asks for).  Both passes draw the same lines: one call rasterises and sorts them once (th_draw) when they draw them
        with the same width."""
        self.line_widths()
        if self.dist is not None:          # row-band shard of a torch.distributed job: emit / exchange / merge, pass by pass
            from .sharding import draw_sharded
            if self.renderView:            # (every rank holds the whole view buffer: the clear / fade are the same everywhere)
                self._bind_view(self.buffers[0] if self.buffers else None)
                if self.state["autoClearView"]:
                    self.clearView()
                if self.state["autoFade"]:
                    self.drawFade()
            self.fragments = draw_sharded(self.dist, self, view=self.renderView)
            self.view_fragments = self.fragments if self.renderView else 0
            return self
        row0, rows, gh = self._band
        if rows and rows != (gh or self.state["rootNum"]):
            # a row band without a torch.distributed host: the exchange is the library's, over the communicator the ranks
            # joined with sharding.comm_init() (th_draw_sharded; what the Node host runs) - every rank calls draw() together
            from .sharding import draw_sharded_native
            if self.renderView:
                self._bind_view(self.buffers[0] if self.buffers else None)
                if self.state["autoClearView"]:
                    self.clearView()
                if self.state["autoFade"]:
                    self.drawFade()
            self.fragments = draw_sharded_native(self, view=self.renderView)
            self.view_fragments = self.fragments if self.renderView else 0
            return self
        if not self.renderView:
            self.fragments = self.particles.deposit_flow(self.viewSize, self.timer.time, self.state["speedLimit"])
            return self
        # (the clear and the fade only touch the view buffer: it does not matter that the flow pass comes after them here)
        # The view goes to buffers[0] when there are buffers, else to the screen (src/index.js:318-325) - unless autoClearView
        # comes in between: clearView() leaves the SCREEN bound (src/index.js:226), there as here.
        self._bind_view(self.buffers[0] if self.buffers else None)
        if self.state["autoClearView"]:
            self.clearView()
        if self.state["autoFade"]:
            self.drawFade()
        d = _capi.DepositUniforms(time=float(self.timer.time), speedLimit=float(self.state["speedLimit"]))
        d.viewSize[0], d.viewSize[1] = float(self.viewSize[0]), float(self.viewSize[1])
        u, n = self.render_uniforms(), C.c_uint64(0)
        call("th_draw", self.particles._ctx, C.byref(d), C.byref(u), C.byref(n))
        self.fragments = self.view_fragments = int(n.value)
        return self

    def export_lines(self, view=False):
        """Trail export: the (previous -> current) line list this frame's draw() is made of (build-defined): [n, 12]
        float32 - p0.xy, p1.xy (clip space), then both vertices' flow varyings, or (view=True) their view colours."""
        if not view:
            return self.particles.export_lines(self.viewSize, self.timer.time, self.state["speedLimit"])
        u = self.render_uniforms()
        n = C.c_uint64(0)
        call("th_export_view_lines", self.particles._ctx, C.byref(u), None, 0, C.byref(n))
        out = np.empty((int(n.value), 12), np.float32)
        if n.value:
            call("th_export_view_lines", self.particles._ctx, C.byref(u), out.ctypes.data_as(_capi._fp), n.value, C.byref(n))
        return out

    def resize(self):                                          # src/index.js:393-408
        self.viewRes[0] = self.gl.drawingBufferWidth
        self.viewRes[1] = self.gl.drawingBufferHeight
        self.viewSize[:] = cover_aspect(self.viewRes)
        for b in self.buffers:                                 # src/index.js:404 (the images follow the flow texture's shape)
            b.shape = list(self.viewRes)
        self.flow.shape = self.viewRes
        return self

    # -- respawn --------------------------------------------------------------------------
    def spawn(self, spawner=init_spawner):                     # src/index.js:425-429
        if spawner is init_spawner:
            # Particles.spawn(initSpawner) fills every ring buffer with inert texels; do it on device
            for k in range(len(self.particles.buffers)):
                call("th_spawn_init", self.particles._ctx, k)
        else:
            self.particles.spawn(spawner)
        return self

    def spawnShader(self, shader, update=None, *rest):         # src/index.js:432-457
        self.timer.tick()                                       # every GPU spawn advances time
        self.particles.logic = shader
        base = dict(self.state, time=self.timer.time, viewSize=self.viewSize, viewRes=self.viewRes)
        self.particles.step(Particles.applyUpdate(base, update), *rest)
        self.particles.logic = self.logicShader
        return self


default = Tendrils
