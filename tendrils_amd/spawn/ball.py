"""src/spawn/ball/index.js:5-16 - hash-random disc spawner (program: src/spawn/ball/index.frag).
CPU variant (Math.random based): src/spawn/ball/cpu.js:1-19."""
import math
import random as _random

from ..particles import Program
from . import init


def frag():
    return Program("spawn-ball")


def defaults():
    return dict(shader=frag(), uniforms=dict(radius=1, speed=0))


def spawnBall(gl=None, options=None):
    base = defaults()
    base.update(options or {})
    return init.spawner(gl, base)


def cpu(data, radius=1.0, speed=0.01):
    angle = _random.random() * math.pi * 2
    scaled = _random.random() * radius
    data[0] = math.cos(angle) * scaled
    data[1] = math.sin(angle) * scaled
    angle = _random.random() * math.pi * 2
    scaled = _random.random() * speed
    data[2] = math.cos(angle) * scaled
    data[3] = math.sin(angle) * scaled
    return data


default = spawnBall
