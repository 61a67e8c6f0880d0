"""ctypes binding of lib/libtendrils_hip.so (the C ABI in include/tendrils_hip.h).

This is plumbing: every symbol the header declares is bound here, nothing is
computed in Python.  There is no CPU fallback - if the library is missing or no
gfx950 device is present, calls raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TH_LIB") or os.path.join(HERE, "lib", "libtendrils_hip.so")   # TH_LIB: diagnostic builds

TH_OK = 0
TH_MODE_EXACT, TH_MODE_FAST = 0, 1
TH_STATE_F32, TH_STATE_F16 = 0, 1
TH_PASS_FLOW, TH_PASS_VIEW = 0, 1
TH_MAX_LINE_WIDTH = 64.0
TH_TARGET_RING, TH_TARGET_TARGETS, TH_SOURCE_FLOW, TH_SOURCE_IMAGE = -1, -2, -3, -4
INERT = -1000000.0


class TendrilsHipError(RuntimeError):
    """A C-ABI call returned a non-zero status (the reference throws JS Errors from gl-fbo/gl-shader)."""

    def __init__(self, status, message):
        super().__init__("tendrils_hip status %d: %s" % (status, message))
        self.status = status


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("global_height", C.c_int32), ("row0", C.c_int32), ("num_buffers", C.c_int32),
                ("mode", C.c_int32), ("state_format", C.c_int32)]


class LogicUniforms(C.Structure):
    _fields_ = [("viewSize", C.c_float * 2),
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
    _fields_ = [("spawnSize", C.c_float * 2), ("jitter", C.c_float * 2),
                ("time", C.c_float), ("speed", C.c_float), ("bias", C.c_float),
                ("flowDecay", C.c_float),
                ("spawnMatrix", C.c_float * 9),
                ("samples", C.c_int32), ("apply", C.c_int32)]


class DepositUniforms(C.Structure):
    _fields_ = [("viewSize", C.c_float * 2), ("time", C.c_float), ("speedLimit", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("particles", C.c_uint64), ("live", C.c_uint64), ("nan", C.c_uint64),
                ("capped", C.c_uint64), ("respawned", C.c_uint64),
                ("sum_speed", C.c_double), ("max_speed", C.c_double)]


class RenderUniforms(C.Structure):
    _fields_ = [("viewSize", C.c_float * 2), ("time", C.c_float), ("speedLimit", C.c_float), ("flowDecay", C.c_float),
                ("speedAlpha", C.c_float), ("colorMapAlpha", C.c_float), ("sinTerm", C.c_float),
                ("baseColor", C.c_float * 4), ("flowColor", C.c_float * 4)]


class ShapesInfo(C.Structure):
    _fields_ = [("state_w", C.c_int32), ("state_h", C.c_int32), ("flow_w", C.c_int32), ("flow_h", C.c_int32),
                ("frames_w", C.c_int32), ("frames_h", C.c_int32)]


class SlotOrderInfo(C.Structure):
    _fields_ = [("sorted_buffers", C.c_int32), ("steps_since_sort", C.c_int32), ("window_misses", C.c_uint64),
                ("sorts", C.c_uint64)]


class DrawInfo(C.Structure):
    _fields_ = [("pipeline", C.c_int32), ("reserved", C.c_int32), ("fragments", C.c_uint64), ("crowded_fragments", C.c_uint64),
                ("sent_bytes", C.c_uint64), ("received_bytes", C.c_uint64)]


class CommInfo(C.Structure):
    _fields_ = [("active", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("rccl_version", C.c_int32)]


COMM_ID_BYTES = 128         # TH_COMM_ID_BYTES

_ctx = C.c_void_p
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/tendrils_hip.h one to one
PROTOTYPES = {
    "th_abi_version": (C.c_int32, []),
    "th_last_error": (C.c_char_p, []),
    "th_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "th_create": (C.c_int32, [C.POINTER(Config), C.POINTER(_ctx)]),
    "th_destroy": (C.c_int32, [_ctx]),
    "th_set_mode": (C.c_int32, [_ctx, C.c_int32]),
    "th_setup": (C.c_int32, [_ctx, C.c_int32]),
    "th_num_buffers": (C.c_int32, [_ctx, C.POINTER(C.c_int32)]),
    "th_upload_state": (C.c_int32, [_ctx, C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "th_download_state": (C.c_int32, [_ctx, C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "th_flow_resize": (C.c_int32, [_ctx, C.c_int32, C.c_int32]),
    "th_flow_upload": (C.c_int32, [_ctx, _fp]),
    "th_flow_download": (C.c_int32, [_ctx, _fp]),
    "th_flow_clear": (C.c_int32, [_ctx]),
    "th_targets_upload": (C.c_int32, [_ctx, _fp]),
    "th_targets_download": (C.c_int32, [_ctx, _fp]),
    "th_targets_clear": (C.c_int32, [_ctx]),
    "th_step": (C.c_int32, [_ctx, C.POINTER(LogicUniforms), C.c_int32]),
    "th_step_n": (C.c_int32, [_ctx, C.POINTER(LogicUniforms), C.c_double, C.c_double, C.c_int32]),
    "th_spawn_init": (C.c_int32, [_ctx, C.c_int32]),
    "th_spawn_ball": (C.c_int32, [_ctx, C.POINTER(SpawnBallUniforms), C.c_int32]),
    "th_spawn_sample": (C.c_int32, [_ctx, C.POINTER(SpawnSampleUniforms), C.c_int32, C.c_int32]),
    "th_spawn_direct": (C.c_int32, [_ctx, C.POINTER(SpawnSampleUniforms), C.c_int32, C.c_int32]),
    "th_spawn_image_upload": (C.c_int32, [_ctx, _fp, C.c_int32, C.c_int32]),
    "th_spawn_image_triangles": (C.c_int32, [_ctx, _fp, C.c_int32, _fp, _fp, C.c_int32, C.c_int32]),
    "th_spawn_image_download": (C.c_int32, [_ctx, _fp]),
    "th_frames_resize": (C.c_int32, [_ctx, C.c_int32, C.c_int32]),
    "th_frames_upload": (C.c_int32, [_ctx, C.POINTER(C.c_uint8)]),
    "th_frames_rotate": (C.c_int32, [_ctx]),
    "th_optical_flow": (C.c_int32, [_ctx, C.POINTER(OpticalFlowUniforms)]),
    "th_flow_deposit": (C.c_int32, [_ctx, C.POINTER(DepositUniforms), C.POINTER(C.c_uint64)]),
    "th_export_lines": (C.c_int32, [_ctx, C.POINTER(DepositUniforms), _fp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "th_deposit_emit": (C.c_int32, [_ctx, C.POINTER(DepositUniforms), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "th_deposit_set_halo": (C.c_int32, [_ctx, C.c_void_p, C.c_void_p]),
    "th_deposit_set_owners": (C.c_int32, [_ctx, C.c_int32]),
    "th_deposit_merge": (C.c_int32, [_ctx, C.c_void_p, C.c_void_p, C.c_uint64]),
    "th_flow_device_ptr": (C.c_int32, [_ctx, C.POINTER(C.c_void_p)]),
    "th_view_emit": (C.c_int32, [_ctx, C.POINTER(RenderUniforms), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "th_view_merge": (C.c_int32, [_ctx, C.c_void_p, C.c_void_p, C.c_uint64]),
    "th_draw_emit": (C.c_int32, [_ctx, C.POINTER(DepositUniforms), C.POINTER(RenderUniforms), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "th_draw_merge": (C.c_int32, [_ctx, C.c_void_p, C.c_void_p, C.c_uint64]),
    "th_draw_sharded": (C.c_int32, [_ctx, C.POINTER(DepositUniforms), C.POINTER(RenderUniforms), C.POINTER(C.c_uint64)]),
    "th_view_device_ptr": (C.c_int32, [_ctx, C.POINTER(C.c_void_p)]),
    "th_state_gather": (C.c_int32, [_ctx, C.c_int32]),
    "th_state_gather_ptr": (C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_void_p)]),
    "th_stats": (C.c_int32, [_ctx, C.c_float, C.POINTER(Counters)]),
    "th_stats_async": (C.c_int32, [_ctx, C.c_float, C.POINTER(C.c_void_p)]),
    "th_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "th_comm_loopback_id": (C.c_int32, [C.c_void_p]),
    "th_comm_init": (C.c_int32, [_ctx, C.c_void_p, C.c_int32, C.c_int32]),
    "th_comm_destroy": (C.c_int32, [_ctx]),
    "th_comm_query": (C.c_int32, [_ctx, C.POINTER(CommInfo)]),
    "th_stats_allreduce": (C.c_int32, [_ctx]),
    "th_stats_global": (C.c_int32, [_ctx, C.c_float, C.POINTER(Counters)]),
    "th_sync": (C.c_int32, [_ctx]),
    "th_stream": (C.c_int32, [_ctx, C.POINTER(C.c_void_p)]),
    "th_state_device_ptr": (C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_void_p)]),
    "th_timer_start": (C.c_int32, [_ctx]),
    "th_timer_stop": (C.c_int32, [_ctx, C.POINTER(C.c_float)]),
    "th_kernel_timing": (C.c_int32, [_ctx, C.c_int32]),
    "th_kernel_timing_read": (C.c_int32, [_ctx, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "th_slot_order": (C.c_int32, [_ctx, C.POINTER(SlotOrderInfo)]),
    "th_shapes": (C.c_int32, [_ctx, C.POINTER(ShapesInfo)]),
    "th_draw_pipeline": (C.c_int32, [_ctx, C.c_int32]),
    "th_draw_query": (C.c_int32, [_ctx, C.POINTER(DrawInfo)]),
    "th_option_set": (C.c_int32, [_ctx, C.c_int32, C.c_int64]),
    "th_option_get": (C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_int64)]),
    "th_line_width": (C.c_int32, [_ctx, C.c_int32, C.c_float]),
    "th_line_width_range": (C.c_int32, [_ctx, C.c_float, C.c_float]),
    "th_line_width_query": (C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "th_view_draw": (C.c_int32, [_ctx, C.POINTER(RenderUniforms), C.POINTER(C.c_uint64)]),
    "th_draw": (C.c_int32, [_ctx, C.POINTER(DepositUniforms), C.POINTER(RenderUniforms), C.POINTER(C.c_uint64)]),
    "th_view_fill": (C.c_int32, [_ctx, _fp]),
    "th_view_clear": (C.c_int32, [_ctx]),
    "th_view_download": (C.c_int32, [_ctx, C.POINTER(C.c_uint8)]),
    "th_view_buffers": (C.c_int32, [_ctx, C.c_int32]),
    "th_view_bind": (C.c_int32, [_ctx, C.c_int32]),
    "th_view_copy": (C.c_int32, [_ctx, C.c_int32]),
    "th_view_step_buffers": (C.c_int32, [_ctx]),
    "th_colormap_upload": (C.c_int32, [_ctx, _fp, C.c_int32, C.c_int32]),
    "th_export_view_lines": (C.c_int32, [_ctx, C.POINTER(RenderUniforms), _fp, C.c_uint64, C.POINTER(C.c_uint64)]),
}

_NO_STATUS = {"th_abi_version", "th_last_error"}
TEST_BUILD_ONLY = {"th_comm_loopback_id"}      # declared under TH_TESTING in include/tendrils_hip.h
_lib = None


def _mapped(name):
    """paths of the shared objects called `name`* that this process has mapped"""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if os.path.basename(path).startswith(name):
                    found.add(os.path.realpath(path))
    except OSError:
        pass
    return found


def _runtime_order():
    """PyTorch-ROCm wheels bundle their own ROCm runtime (libamdhip64 / libhsa-runtime64 under torch/lib); this library
    links the system one.  A process must end up with ONE runtime: the copy that is mapped first serves everybody
    (same SONAMEs), and that only works in one order - torch's first (the other way round torch finds no devices).  So
    when torch is installed it is imported before the library is loaded, whatever the caller's import order was;
    TH_SKIP_TORCH=1 skips that (hosts that never touch torch)."""
    import importlib.util
    import sys
    if os.environ.get("TH_SKIP_TORCH") == "1" or "torch" in sys.modules:
        return
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load():
    """Load the shared library (raises OSError if it was not built) and bind every symbol."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found - run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C tendrils_amd/csrc)" % LIB_PATH)
        _runtime_order()
        lib = C.CDLL(LIB_PATH)
        # (a profiler's preloaded tool library maps the system runtime before Python starts: not this loader's doing)
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        paths = _mapped("libamdhip64")
        import sys
        if len(paths) > 1 and not profiled and "torch" not in sys.modules:
            # (a host that never touches torch shares no streams with it: two copies on disk under different names -
            # a hashed wheel copy, another ROCm - are this host's own business)
            import warnings
            warnings.warn("two copies of the HIP runtime are mapped into this process: %s" % ", ".join(sorted(paths)))
        elif len(paths) > 1 and not profiled:
            raise OSError("two copies of the HIP runtime are mapped into this process (%s): streams and events cannot be "
                          "shared between them - import torch before anything loads the ROCm runtime, or set "
                          "TH_SKIP_TORCH=1 and keep torch out of the process" % ", ".join(sorted(paths)))
        for name, (res, args) in PROTOTYPES.items():
            if name in TEST_BUILD_ONLY and not hasattr(lib, name):
                continue                 # a release build of the library (make release): the test machinery is not in it
            fn = getattr(lib, name)      # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def call(name, *args):
    """Invoke a status-returning entry point; raise TendrilsHipError on failure."""
    lib = load()
    status = getattr(lib, name)(*args)
    if name not in _NO_STATUS and status != TH_OK:
        raise TendrilsHipError(status, lib.th_last_error().decode(errors="replace"))
    return status
