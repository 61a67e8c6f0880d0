"""TEST INFRASTRUCTURE - ctypes binding of oracle/libtendrils_oracle.so.

CPU restatement of the reference's particle path (see tendrils_oracle.c).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package (tendrils_amd/) must never do so.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtendrils_oracle.so")

INERT = np.float32(-1000000.0)

# reference defaults: /root/reference/src/index.js:28-66 (state) and timer step 1000/60 (:67)
DEFAULT_STATE = dict(
    damping=0.043, speedLimit=0.01,
    forceWeight=0.016, varyForce=-0.1,
    flowWeight=1.0, varyFlow=0.2,
    noiseWeight=0.002, varyNoise=0.3,
    flowDecay=0.005,
    noiseScale=2.125, varyNoiseScale=0.5,
    noiseSpeed=0.00025, varyNoiseSpeed=0.1,
    target=0.0, varyTarget=1.0,
)


class LogicUniforms(C.Structure):
    _fields_ = [("data_w", C.c_int32), ("data_h", C.c_int32),
                ("viewSize", C.c_float * 2),
                ("time", C.c_float), ("dt", C.c_float),
                ("speedLimit", C.c_float), ("damping", C.c_float),
                ("forceWeight", C.c_float), ("flowWeight", C.c_float), ("noiseWeight", C.c_float),
                ("flowDecay", C.c_float),
                ("noiseSpeed", C.c_float), ("noiseScale", C.c_float),
                ("target", C.c_float),
                ("varyForce", C.c_float), ("varyFlow", C.c_float), ("varyNoise", C.c_float),
                ("varyNoiseScale", C.c_float), ("varyNoiseSpeed", C.c_float), ("varyTarget", C.c_float)]


class OpticalFlowUniforms(C.Structure):
    _fields_ = [("viewSize", C.c_float * 2), ("scaleUV", C.c_float * 2),
                ("offset", C.c_float), ("lambda_", C.c_float),
                ("time", C.c_float), ("speed", C.c_float), ("speedLimit", C.c_float)]


class SpawnBallUniforms(C.Structure):
    _fields_ = [("radius", C.c_float), ("speed", C.c_float)]


class SpawnSampleUniforms(C.Structure):
    _fields_ = [("data_w", C.c_int32), ("data_h", C.c_int32),
                ("spawnSize", C.c_float * 2), ("jitter", C.c_float * 2),
                ("time", C.c_float), ("speed", C.c_float), ("bias", C.c_float),
                ("flowDecay", C.c_float),
                ("spawnMatrix", C.c_float * 9),
                ("samples", C.c_int32), ("apply", C.c_int32)]


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("tendrils_oracle.c", "tendrils_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB_PATH) or \
            any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libtendrils_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


class DepositUniforms(C.Structure):
    _fields_ = [("data_w", C.c_int32), ("data_h", C.c_int32), ("viewSize", C.c_float * 2),
                ("time", C.c_float), ("speedLimit", C.c_float), ("lineWidth", C.c_float)]


class RenderUniforms(C.Structure):
    _fields_ = [("speedLimit", C.c_float), ("flowDecay", C.c_float), ("speedAlpha", C.c_float), ("colorMapAlpha", C.c_float),
                ("sinTerm", C.c_float), ("baseColor", C.c_float * 4), ("flowColor", C.c_float * 4)]


def render_uniforms(time, speedLimit=0.01, flowDecay=0.005, speedAlpha=0.000001, colorMapAlpha=0.4,
                    baseColor=(1, 1, 1, 0.5), flowColor=(1, 1, 1, 0.04), sin_term=None, **_):
    """defaults: src/index.js:58-65; sin(time*flowDecay) in double, rounded once (what the host mirrors pass down)"""
    import math
    r = RenderUniforms(speedLimit=speedLimit, flowDecay=flowDecay, speedAlpha=speedAlpha, colorMapAlpha=colorMapAlpha,
                       sinTerm=math.sin(float(np.float32(time)) * float(np.float32(flowDecay))) if sin_term is None else sin_term)
    for k in range(4):
        r.baseColor[k], r.flowColor[k] = float(baseColor[k]), float(flowColor[k])
    return r


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.to_snoise3.restype = C.c_float
        L.to_snoise3.argtypes = [C.c_float] * 3
        L.to_logic_step.restype = None
        L.to_logic_step.argtypes = [C.POINTER(LogicUniforms), fp, fp, C.c_int, C.c_int,
                                    fp, C.c_int, C.c_int, fp]
        L.to_spawn_init.restype = None
        L.to_spawn_init.argtypes = [fp, C.c_size_t]
        L.to_optical_flow.restype = None
        L.to_optical_flow.argtypes = [C.POINTER(OpticalFlowUniforms), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                                      C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int]
        L.to_random.restype = C.c_float
        L.to_random.argtypes = [C.c_float, C.c_float]
        L.to_spawn_ball.restype = None
        L.to_spawn_ball.argtypes = [C.POINTER(SpawnBallUniforms), fp, C.c_int, C.c_int, C.c_int]
        L.to_spawn_sample.restype = None
        L.to_spawn_sample.argtypes = [C.POINTER(SpawnSampleUniforms), fp, fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.to_spawn_direct.restype = None
        L.to_spawn_direct.argtypes = [C.POINTER(SpawnSampleUniforms), fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.to_triangles.restype = None
        L.to_triangles.argtypes = [fp, C.c_int, fp, fp, fp, C.c_int, C.c_int]
        L.to_export_lines.restype = C.c_long
        L.to_export_lines.argtypes = [C.POINTER(DepositUniforms), fp, fp, fp, C.c_long]
        L.to_flow_deposit.restype = C.c_long
        L.to_flow_deposit.argtypes = [C.POINTER(DepositUniforms), fp, fp, fp, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def logic_uniforms(data_w, data_h, time, dt, view_size=(1.0, 1.0), **state):
    s = dict(DEFAULT_STATE)
    for k, v in state.items():
        if k in s:
            s[k] = v
    u = LogicUniforms()
    u.data_w, u.data_h = int(data_w), int(data_h)
    u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
    u.time, u.dt = float(time), float(dt)     # double -> fp32 exactly as gl.uniform1f does
    for k, v in s.items():
        setattr(u, k, float(v))
    return u


def snoise3(x, y, z):
    return lib().to_snoise3(float(x), float(y), float(z))


def logic_step(u, state, flow, targets=None, y0=0, out=None):
    """state: [rows, W, 4] f32 band starting at global row y0; flow: [fh, fw, 4].  `out` may be given to reuse a
    buffer (timing runs: keeps page faults of a fresh 256 MiB array out of the measurement)."""
    state = np.ascontiguousarray(state, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    rows, W = state.shape[:2]
    assert W == u.data_w and state.shape[2] == 4
    if out is None:
        out = np.empty_like(state)
    assert out.shape == state.shape and out.dtype == np.float32 and out.flags.c_contiguous
    tp = None
    if targets is not None:
        targets = np.ascontiguousarray(targets, np.float32)
        assert targets.shape == state.shape
        tp = _fp(targets)
    fh, fw = flow.shape[:2]
    lib().to_logic_step(C.byref(u), _fp(state), _fp(out), int(y0), int(rows), _fp(flow), fw, fh, tp)
    return out


def spawn_init(shape):
    out = np.empty(tuple(shape) + (4,), np.float32)
    lib().to_spawn_init(_fp(out), out.size // 4)
    return out


def optical_flow_uniforms(time, view_size=(1.0, 1.0), scaleUV=(1.0, -1.0), offset=1.0, lambda_=0.001,
                          speed=1.0, speedLimit=1.0, **_):
    """defaults: /root/reference/src/optical-flow/index.js:21-29"""
    u = OpticalFlowUniforms()
    u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
    u.scaleUV[0], u.scaleUV[1] = float(scaleUV[0]), float(scaleUV[1])
    u.offset, u.lambda_, u.time, u.speed, u.speedLimit = float(offset), float(lambda_), float(time), float(speed), float(speedLimit)
    return u


def optical_flow(u, view, last, flow, blend=True):
    """view/last: [h, w, 4] uint8; flow: [H, W, 4] f32 (destination contents); returns the new flow."""
    view = np.ascontiguousarray(view, np.uint8)
    last = np.ascontiguousarray(last, np.uint8)
    out = np.array(flow, np.float32, order="C", copy=True)
    h, w = view.shape[:2]
    assert last.shape == view.shape
    lib().to_optical_flow(C.byref(u), view.ctypes.data_as(C.POINTER(C.c_uint8)),
                          last.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, _fp(out), out.shape[1], out.shape[0],
                          1 if blend else 0)
    return out


def spawn_ball(w, rows, radius=1.0, speed=0.0, y0=0):
    """src/spawn/ball/index.js:7-10 defaults: radius 1, speed 0."""
    u = SpawnBallUniforms(radius=float(radius), speed=float(speed))
    out = np.empty((rows, w, 4), np.float32)
    lib().to_spawn_ball(C.byref(u), _fp(out), int(w), int(y0), int(rows))
    return out


def spawn_sample_uniforms(data_w, data_h, time, samples, apply, spawnSize=(1.0, 1.0), jitter=(0.0, 0.0),
                          speed=1.0, bias=1.0, flowDecay=0.005, spawnMatrix=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
    u = SpawnSampleUniforms()
    u.data_w, u.data_h = int(data_w), int(data_h)
    u.spawnSize[0], u.spawnSize[1] = float(spawnSize[0]), float(spawnSize[1])
    u.jitter[0], u.jitter[1] = float(jitter[0]), float(jitter[1])
    u.time, u.speed, u.bias, u.flowDecay = float(time), float(speed), float(bias), float(flowDecay)
    for k in range(9):
        u.spawnMatrix[k] = float(spawnMatrix[k])
    u.samples, u.apply = int(samples), int(apply)
    return u


def spawn_sample(u, particles, spawn_data, y0=0):
    particles = np.ascontiguousarray(particles, np.float32)
    spawn_data = np.ascontiguousarray(spawn_data, np.float32)
    out = np.empty_like(particles)
    rows = particles.shape[0]
    sh, sw = spawn_data.shape[:2]
    lib().to_spawn_sample(C.byref(u), _fp(particles), _fp(out), int(y0), int(rows), _fp(spawn_data), sw, sh)
    return out


def flow_deposit(current, previous, flow, time, view_size=(1.0, 1.0), speedLimit=0.01, coverage=False, line_width=1.0):
    """Tendrils.draw()'s flow pass: blends the particle lines into a copy of `flow` [fh, fw, 4].
    Returns (flow_out, fragments[, per-texel fragment counts]).  line_width: the width the GL draws with (after its
    clamp to ALIASED_LINE_WIDTH_RANGE; 1 on the captured GL - wider lines are unpinned, see tendrils_oracle.c)."""
    current = np.ascontiguousarray(current, np.float32)
    previous = np.ascontiguousarray(previous, np.float32)
    out = np.array(flow, np.float32, copy=True, order="C")
    h, w = current.shape[:2]
    assert previous.shape == current.shape and current.shape[2] == 4 and out.shape[2] == 4
    fh, fw = out.shape[:2]
    u = DepositUniforms(data_w=w, data_h=h, time=float(time), speedLimit=float(speedLimit), lineWidth=float(line_width))
    u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
    cov = np.zeros((fh, fw), np.int32) if coverage else None
    n = lib().to_flow_deposit(C.byref(u), _fp(current), _fp(previous), _fp(out), fw, fh,
                              cov.ctypes.data_as(C.POINTER(C.c_int32)) if coverage else None)
    return (out, n, cov) if coverage else (out, n)


def spawn_direct(u, spawn_data, y0=0, rows=None):
    """index.frag (direct-main): particle (x, y) from its own texel of spawn_data [sh, sw, 4]."""
    spawn_data = np.ascontiguousarray(spawn_data, np.float32)
    rows = u.data_h if rows is None else rows
    out = np.empty((rows, u.data_w, 4), np.float32)
    sh, sw = spawn_data.shape[:2]
    lib().to_spawn_direct(C.byref(u), _fp(out), int(y0), int(rows), _fp(spawn_data), sw, sh)
    return out


def triangles(positions, shape, view_size=(1.0, 1.0), color=(1.0, 1.0, 1.0, 1.0), img=None):
    """GeometrySpawner's draw: triangles (flat xy list, 6 floats each) into a cleared [h, w, 4] float image."""
    w, h = shape
    pos = np.ascontiguousarray(positions, np.float32).ravel()
    out = np.zeros((h, w, 4), np.float32) if img is None else np.array(img, np.float32, copy=True, order="C")
    vs = np.asarray(view_size, np.float32)
    col = np.asarray(color, np.float32)
    lib().to_triangles(_fp(pos), len(pos) // 6, _fp(vs), _fp(col), _fp(out), int(w), int(h))
    return out


def view_render(current, previous, view, time, view_size=(1.0, 1.0), colormap=None, **uniforms):
    """Tendrils.draw()'s view pass: blends the particle lines into a copy of `view` [vh, vw, 4] uint8.
    Returns (view_out, fragments)."""
    current = np.ascontiguousarray(current, np.float32)
    previous = np.ascontiguousarray(previous, np.float32)
    out = np.array(view, np.uint8, copy=True, order="C")
    h, w = current.shape[:2]
    vh, vw = out.shape[:2]
    r = render_uniforms(time, **uniforms)
    u = DepositUniforms(data_w=w, data_h=h, time=float(time), speedLimit=float(r.speedLimit), lineWidth=float(uniforms.get("line_width", 1.0)))
    u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
    cm = None if colormap is None else np.ascontiguousarray(colormap, np.float32)
    L = lib()
    L.to_view_render.restype = C.c_long
    n = L.to_view_render(C.byref(u), C.byref(r), _fp(cm) if cm is not None else None, 0 if cm is None else cm.shape[1],
                         0 if cm is None else cm.shape[0], _fp(current), _fp(previous),
                         out.ctypes.data_as(C.POINTER(C.c_uint8)), vw, vh)
    return out, n


def view_fill(view, color):
    out = np.array(view, np.uint8, copy=True, order="C")
    col = np.asarray(color, np.float32)
    lib().to_view_fill(out.ctypes.data_as(C.POINTER(C.c_uint8)), out.shape[1], out.shape[0], _fp(col))
    return out


def export_view_lines(current, previous, time, view_size=(1.0, 1.0), colormap=None, **uniforms):
    """The line list of draw() with the view pass's vertex colours: [n, 12] float32 in stream order."""
    current = np.ascontiguousarray(current, np.float32)
    previous = np.ascontiguousarray(previous, np.float32)
    h, w = current.shape[:2]
    r = render_uniforms(time, **uniforms)
    u = DepositUniforms(data_w=w, data_h=h, time=float(time), speedLimit=float(r.speedLimit))
    u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
    cm = None if colormap is None else np.ascontiguousarray(colormap, np.float32)
    out = np.empty((w * h, 12), np.float32)
    L = lib()
    L.to_export_view_lines.restype = C.c_long
    n = L.to_export_view_lines(C.byref(u), C.byref(r), _fp(cm) if cm is not None else None, 0 if cm is None else cm.shape[1],
                               0 if cm is None else cm.shape[0], _fp(current), _fp(previous), _fp(out), w * h)
    return out[:n].copy()


def export_lines(current, previous, time, view_size=(1.0, 1.0), speedLimit=0.01):
    """The line list of draw(): [n, 12] float32 (p0.xy, p1.xy, c0, c1) in stream order."""
    current = np.ascontiguousarray(current, np.float32)
    previous = np.ascontiguousarray(previous, np.float32)
    h, w = current.shape[:2]
    u = DepositUniforms(data_w=w, data_h=h, time=float(time), speedLimit=float(speedLimit))
    u.viewSize[0], u.viewSize[1] = float(view_size[0]), float(view_size[1])
    out = np.empty((w * h, 12), np.float32)
    n = lib().to_export_lines(C.byref(u), _fp(current), _fp(previous), _fp(out), w * h)
    return out[:n].copy()
