"""Host mirror of the reference's OpticalFlow (src/optical-flow/index.js:32-71): two RGBA8 frame
buffers (view = buffers[0], last = buffers[1]), uniforms, update()/step()/setPixels()/resize().
The full-screen draw that the demo issues after update() (src/demo.main.js:1107-1159) is `render()`,
which runs the HIP pass alpha-blended into tendrils.flow."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import call


def defaults():
    """src/optical-flow/index.js:15-30"""
    return dict(options=dict(shader=None, buffers=[[[1, 1]], [[1, 1]]]),
                uniforms=dict(viewSize=[1, 1], scaleUV=[1, -1], offset=1, speed=1, speedLimit=1, time=1,
                              **{"lambda": 0.001}))


class OpticalFlow:
    def __init__(self, tendrils, options=None, uniforms=None):
        base = defaults()
        self.tendrils = tendrils
        self.buffers = [0, 1]                     # identities of the two frame textures; [0] is `view`
        self.uniforms = dict(base["uniforms"], **(uniforms or {}))
        self.shape = [1, 1]
        self._bound = dict(self.uniforms)

    @property
    def _ctx(self):
        return self.tendrils.particles._ctx

    def update(self, uniforms=None):              # src/optical-flow/index.js:50-58
        self._bound = dict(self.uniforms, **(uniforms or {}))
        return self._bound

    def render(self):
        """screen.render() with the optical-flow shader bound, into tendrils.flow (blended)."""
        b = self._bound
        u = _capi.OpticalFlowUniforms()
        u.viewSize[0], u.viewSize[1] = float(b["viewSize"][0]), float(b["viewSize"][1])
        u.scaleUV[0], u.scaleUV[1] = float(b["scaleUV"][0]), float(b["scaleUV"][1])
        u.offset, u.lambda_ = float(b["offset"]), float(b["lambda"])
        u.time, u.speed, u.speedLimit = float(b["time"]), float(b["speed"]), float(b["speedLimit"])
        call("th_optical_flow", self._ctx, C.byref(u))

    def step(self):                               # :60-62 utils.step(this.buffers)
        self.buffers.insert(0, self.buffers.pop())
        call("th_frames_rotate", self._ctx)

    def set_pixels(self, pixels):                 # :64-66 setPixels -> buffers[0]
        px = np.ascontiguousarray(pixels, np.uint8)
        assert px.shape == (self.shape[1], self.shape[0], 4), (px.shape, self.shape)
        call("th_frames_upload", self._ctx, px.ctypes.data_as(C.POINTER(C.c_uint8)))

    setPixels = set_pixels

    def resize(self, size):                       # :68-70
        self.shape = [int(size[0]), int(size[1])]
        call("th_frames_resize", self._ctx, self.shape[0], self.shape[1])


default = OpticalFlow
