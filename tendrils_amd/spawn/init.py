"""src/spawn/init/index.js:6-28 - generic spawner factory; default program = all-inert fill
(src/spawn/init/index.frag:5-10).  CPU variant: src/spawn/init/cpu.js:3-8."""
from .._capi import INERT
from ..particles import Program


def frag():
    return Program("spawn-init")


def defaults():
    return dict(shader=frag(), uniforms=None)


class Spawner:
    def __init__(self, gl, params):
        self.gl = gl
        self.uniforms = params["uniforms"]
        self.shader = params["shader"]

    def spawn(self, tendrils, *rest):                 # src/spawn/init/index.js:22-24
        tendrils.spawnShader(self.shader, self.uniforms, *rest)


def spawner(gl=None, options=None):
    return Spawner(gl, {**defaults(), **(options or {})})


def cpu(data, x=0, y=0):
    data[0] = data[1] = INERT
    data[2] = data[3] = 0
    return data


default = spawner
