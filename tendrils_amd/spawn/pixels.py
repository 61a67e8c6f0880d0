"""src/spawn/pixels/index.js:15-67 - PixelSpawner: samples a texture (`buffer`: tendrils.flow or a
particle state buffer) to respawn particles.  Programs built here: flow-sample.frag (5 taps,
apply/flow.glsl) and data-sample.frag (2 taps, identity after the vignette pass)."""
from ..particles import Program


def flow_sample_frag():
    """src/spawn/pixels/flow-sample.frag:1-10"""
    return Program("spawn-sample", samples=5, apply=0)


def data_sample_frag():
    """src/spawn/pixels/data-sample.frag:1-12"""
    return Program("spawn-sample", samples=2, apply=1)


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
        self.buffer = params["buffer"]
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


default = PixelSpawner
