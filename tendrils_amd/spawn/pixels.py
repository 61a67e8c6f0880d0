"""src/spawn/pixels/index.js:15-67 - PixelSpawner: samples a texture (`buffer`: tendrils.flow or a
particle state buffer) to respawn particles.  Programs built here: flow-sample.frag (5 taps,
apply/flow.glsl), data-sample.frag (2 taps, identity after the vignette pass), best-sample.frag (6 taps, colour
apply after the vignette pass) and index.frag (direct: every particle from its own texel, colour apply)."""
import ctypes as C

import numpy as np

from .. import _capi
from ..particles import Program


def flow_sample_frag():
    """src/spawn/pixels/flow-sample.frag:1-10"""
    return Program("spawn-sample", samples=5, apply=0)


def data_sample_frag():
    """src/spawn/pixels/data-sample.frag:1-12"""
    return Program("spawn-sample", samples=2, apply=1)


def best_sample_frag():
    """src/spawn/pixels/best-sample.frag:1-25 (image spawner `sample`, src/demo.main.js:457)"""
    return Program("spawn-sample", samples=6, apply=2)


def pixels_frag():
    """src/spawn/pixels/index.frag:1-16 (image spawner `direct`, src/demo.main.js:456)"""
    return Program("spawn-direct", apply=2)


class ImageBuffer:
    """The spawner's own buffer: FBO(gl, [1, 1], {float: true}) (src/spawn/pixels/index.js:17,34-36) holding an
    RGBA image as float texels; the pixels travel to the device when a pass first uses them."""

    def __init__(self, shape=(1, 1)):
        self._shape = [int(shape[0]), int(shape[1])]
        self._pixels = np.zeros((self._shape[1], self._shape[0], 4), np.float32)
        self._dirty = True
        self.color = [self]

    @property
    def shape(self):
        return list(self._shape)

    @shape.setter
    def shape(self, wh):
        self._shape = [int(wh[0]), int(wh[1])]
        self._pixels = np.zeros((self._shape[1], self._shape[0], 4), np.float32)
        self._dirty = True

    def setPixels(self, pixels):
        """pixels: [h, w, 4] float (0..1) or uint8 (converted as WebGL does for a float texture: c/255)."""
        a = np.asarray(pixels)
        if a.dtype == np.uint8:
            a = a.astype(np.float32) / np.float32(255.0)
        a = np.ascontiguousarray(a, np.float32)
        assert a.ndim == 3 and a.shape[2] == 4
        self._shape = [a.shape[1], a.shape[0]]
        self._pixels = a
        self._dirty = True
        return self

    def bind_for(self, particles):
        if self._dirty or getattr(self, "_bound_to", None) is not particles:
            _capi.call("th_spawn_image_upload", particles._ctx, self._pixels.ctypes.data_as(_capi._fp),
                       self._shape[0], self._shape[1])
            self._dirty, self._bound_to = False, particles

    def source_index(self):
        return _capi.TH_SOURCE_IMAGE


def defaults():
    return dict(shader=None, buffer=None, spawnSize=[1, 1], jitterRad=2, speed=1, bias=1)


def aspect(size, scale):
    """src/utils/aspect.js:4-5: scale(inverse(size), scale)"""
    return [scale / size[0], scale / size[1]]


class PixelSpawner:
    def __init__(self, gl=None, options=None):
        params = {**defaults(), **(options or {})}
        self.gl = gl
        self.shader = params["shader"] or flow_sample_frag()
        self.buffer = params["buffer"] if params["buffer"] is not None else ImageBuffer()
        self.speed = params["speed"]
        self.bias = params["bias"]
        self.jitterRad = params["jitterRad"]
        self.jitter = [0.0, 0.0]
        self.spawnSize = list(params["spawnSize"])
        self.spawnMatrix = [1, 0, 0, 0, 1, 0, 0, 0, 1]       # mat3.create()

    def update(self, uniforms):                              # src/spawn/pixels/index.js:47-56
        self.jitter = aspect(uniforms["viewRes"], self.jitterRad)
        uniforms.update(spawnData=self.buffer, spawnSize=self.spawnSize, spawnMatrix=self.spawnMatrix,
                        speed=self.speed, jitter=self.jitter, bias=self.bias)
        return uniforms

    def spawn(self, tendrils, update=None, *rest):           # :58-60
        return tendrils.spawnShader(self.shader, update or self.update, *rest)

    def setPixels(self, pixels):                             # :62-64
        return self.buffer.color[0].setPixels(pixels)


default = PixelSpawner
