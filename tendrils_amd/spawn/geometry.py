"""src/spawn/geometry/index.js:22-118 - GeometrySpawner: draws shuffled triangles ("simple Platonic forms") into
the spawner's buffer and respawns particles from it with bright-sample.frag (apply/brightest.glsl, 6 samples)."""
import math
import random

import numpy as np

from .. import _capi
from ..particles import Program
from .pixels import PixelSpawner


def bright_sample_frag():
    """src/spawn/pixels/bright-sample.frag:1-8"""
    return Program("spawn-sample", samples=6, apply=3)


def defaults():                                              # src/spawn/geometry/index.js:22-32
    return dict(shader=None, color=[1, 1, 1, 1], positions=[0.0] * (2 * 3 * 1),
                shuffles=dict(size=2, count=3, radii=[0.25, 1.3], arcs=[1e-2, 3e-2], obtuse=dict(rate=0.5, pad=0.25)))


class GeometryBuffer:
    """The spawner's buffer when it is drawn into rather than uploaded: lives on the device only."""

    def __init__(self):
        self._shape = [1, 1]
        self.color = [self]
        self._p = None

    @property
    def shape(self):
        return list(self._shape)

    @shape.setter
    def shape(self, wh):
        self._shape = [int(wh[0]), int(wh[1])]

    def draw(self, particles, positions, view_size, color):
        pos = np.ascontiguousarray(positions, np.float32)
        vs = np.asarray(view_size, np.float32)
        col = np.asarray(color, np.float32)
        _capi.call("th_spawn_image_triangles", particles._ctx, pos.ctypes.data_as(_capi._fp), len(pos) // 6,
                   vs.ctypes.data_as(_capi._fp), col.ctypes.data_as(_capi._fp), self._shape[0], self._shape[1])
        self._p = particles

    def read(self):
        out = np.empty((self._shape[1], self._shape[0], 4), np.float32)
        _capi.call("th_spawn_image_download", self._p._ctx, out.ctypes.data_as(_capi._fp))
        return out

    def source_index(self):
        return _capi.TH_SOURCE_IMAGE


class GeometrySpawner(PixelSpawner):
    def __init__(self, gl=None, options=None):
        to = defaults()
        options = dict(options or {})
        shuffles = {**to["shuffles"], **options.pop("shuffles", {})}
        to.update(options)
        to["shuffles"] = shuffles
        super().__init__(gl, dict(shader=to["shader"] or bright_sample_frag(), buffer=GeometryBuffer(),
                                  **{k: to[k] for k in ("spawnSize", "jitterRad", "speed", "bias") if k in to}))
        self.color = list(to["color"])
        self.positions = list(to["positions"])
        self.shuffles = shuffles
        self.random = random.random                          # Math.random: replaceable for reproducible runs

    def shuffle(self):
        """One random fan blade per triangle: vertex 0 stays at the origin, vertices 1 and 2 sit on either side of a
        random direction, `arc` apart from it, each at its own random radius (src/spawn/geometry/index.js:53-95).
        Triangles are visited last to first and draw (direction, arc width, obtuse?, radius, radius) from
        `self.random` in that order - the reference's consumption order."""
        cfg = self.shuffles
        per_triangle = cfg["size"] * cfg["count"]
        full_turn = 2 * math.pi
        draw = self.random

        def rim_point(direction):
            reach = cfg["radii"][0] + draw() * cfg["radii"][1]
            return math.cos(direction) * reach, math.sin(direction) * reach

        for last in range(len(self.positions) - 1, -1, -per_triangle):
            heading = full_turn * draw()
            spread = cfg["arcs"][0] + draw() * cfg["arcs"][1]
            if draw() < cfg["obtuse"]["rate"]:
                spread += cfg["obtuse"]["pad"]
            spread *= full_turn
            self.positions[last - 3], self.positions[last - 2] = rim_point(heading - spread)
            self.positions[last - 1], self.positions[last] = rim_point(heading + spread)
        return self

    def spawn(self, tendrils, *rest):                        # :97-117
        self.buffer.shape = [tendrils.viewRes[0] * 0.2, tendrils.viewRes[1] * 0.2]      # vec2.scale(shape, viewRes, 0.2)
        self.buffer.draw(tendrils.particles, self.positions, tendrils.viewSize, self.color)
        return super().spawn(tendrils, *rest)


default = GeometrySpawner
