'use strict';
// src/spawn/geometry/index.js:22-118 - GeometrySpawner: draws shuffled triangles ("simple Platonic forms") into the
// spawner's buffer and respawns particles from it with bright-sample.frag (apply/brightest.glsl, 6 samples).
const native = require('../native');
const { Program } = require('../particles');
const { PixelSpawner } = require('./pixels');

const brightSampleFrag = () => new Program('spawn-sample', { samples: 6, apply: 3 });   // bright-sample.frag:1-8

const defaults = () => ({                          // src/spawn/geometry/index.js:22-32
  shader: null, color: [1, 1, 1, 1], positions: Array(2 * 3 * 1).fill(0),
  shuffles: { size: 2, count: 3, radii: [0.25, 1.3], arcs: [1e-2, 3e-2], obtuse: { rate: 0.5, pad: 0.25 } }
});

// The spawner's buffer when it is drawn into rather than uploaded: lives on the device only.
class GeometryBuffer {
  constructor() { this._shape = [1, 1]; this.color = [this]; }
  get shape() { return this._shape.slice(); }
  set shape(wh) { this._shape = [wh[0] | 0, wh[1] | 0]; }
  draw(particles, positions, viewSize, color) {
    native.spawnImageTriangles(particles.handle, Float32Array.from(positions),
      new Float32Array([viewSize[0], viewSize[1], color[0], color[1], color[2], color[3]]), this._shape[0], this._shape[1]);
  }
  sourceIndex() { return native.SOURCE_IMAGE; }
}

class GeometrySpawner extends PixelSpawner {
  constructor(gl, options = {}) {
    const to = defaults();
    const shuffles = Object.assign(to.shuffles, options.shuffles);
    Object.assign(to, options).shuffles = shuffles;
    to.shader = (to.shader || brightSampleFrag());
    to.buffer = new GeometryBuffer();
    super(gl, to);
    this.color = to.color;
    this.positions = to.positions;
    this.shuffles = shuffles;
    this.random = Math.random;                     // replaceable for reproducible runs
  }

  // One random fan blade per triangle: vertex 0 stays at the origin, vertices 1 and 2 sit on either side of a
  // random direction, `arc` apart from it, each at its own random radius (src/spawn/geometry/index.js:53-95).
  // Triangles are visited last to first and draw (direction, arc width, obtuse?, radius, radius) from `random`
  // in that order - the reference's consumption order, so a seeded generator reproduces its forms.
  shuffle() {
    const cfg = this.shuffles;
    const perTriangle = cfg.size * cfg.count;
    const fullTurn = 2 * Math.PI;
    const draw = this.random;
    const rimPoint = (direction) => {
      const reach = cfg.radii[0] + draw() * cfg.radii[1];
      return [Math.cos(direction) * reach, Math.sin(direction) * reach];
    };
    for (let last = this.positions.length - 1; last >= 0; last -= perTriangle) {
      const heading = fullTurn * draw();
      let spread = cfg.arcs[0] + draw() * cfg.arcs[1];
      if (draw() < cfg.obtuse.rate) spread += cfg.obtuse.pad;
      spread *= fullTurn;
      const [ax, ay] = rimPoint(heading - spread);
      const [bx, by] = rimPoint(heading + spread);
      this.positions[last - 3] = ax; this.positions[last - 2] = ay;
      this.positions[last - 1] = bx; this.positions[last] = by;
    }
    return this;
  }

  spawn(tendrils, ...rest) {                       // :97-117
    this.buffer.shape = [tendrils.viewRes[0] * 0.2, tendrils.viewRes[1] * 0.2];   // vec2.scale(shape, viewRes, 0.2)
    this.buffer.draw(tendrils.particles, this.positions, tendrils.viewSize, this.color);
    return super.spawn(tendrils, ...rest);
  }
}

module.exports = { defaults, GeometrySpawner, brightSampleFrag, default: GeometrySpawner };
