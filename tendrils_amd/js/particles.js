'use strict';
// Mirror of the reference's GPGPU core `Particles` (src/particles.js:43-196): same options,
// fields and methods; the FBO ring, the full-screen logic pass and the spawn upload go to the
// HIP library through the N-API shim instead of WebGL.  `logic` is an opaque program object
// naming the kernel family a pass runs (the reference swaps gl-shader objects:
// src/index.js:250,435,451).
const native = require('./native');
const { step } = require('./utils');

const defaults = () => ({
  shape: [64, 64],
  geomShape: null,
  logic: null, logicVert: null, logicFrag: null,
  render: null, renderVert: null, renderFrag: null
});

// Stand-in for a compiled gl-shader: { kind, fixed, uniforms }.
class Program {
  constructor(kind, fixed = {}) {
    this.kind = kind;       // 'logic' | 'spawn-init' | 'spawn-ball' | 'spawn-sample'
    this.fixed = fixed;     // compile-time constants of that shader (samples, apply)
    this.uniforms = {};
  }
  bind() { return this; }
}

// Stand-in for one gl-fbo of the ring; identity survives ring rotation.
class StateBuffer {
  constructor(particles, id) {
    this.particles = particles;
    this.id = id;
    this.shape = [...particles.shape];
  }
  get index() { return this.particles.buffers.indexOf(this); }
  sourceIndex() { return this.index; }
  read(out) { return this.particles.read(this, out); }
  dispose() {}
}

const LOGIC_FIELDS = ['time', 'dt', 'speedLimit', 'damping', 'forceWeight', 'flowWeight', 'noiseWeight',
  'flowDecay', 'noiseSpeed', 'noiseScale', 'target', 'varyForce', 'varyFlow', 'varyNoise',
  'varyNoiseScale', 'varyNoiseSpeed', 'varyTarget'];

// uniform object -> Float32Array in th_logic_uniforms layout (Float32Array stores do the
// double -> fp32 rounding gl.uniform1f does)
function packLogic(u) {
  const f = new Float32Array(19);
  const vs = u.viewSize || [1, 1];
  f[0] = vs[0]; f[1] = vs[1];
  LOGIC_FIELDS.forEach((name, i) => { f[2 + i] = +(u[name] || 0); });
  return f;
}

function runPass(particles, program, uniforms, target) {
  const h = particles.handle;
  switch (program.kind) {
    case 'logic':
      native.step(h, packLogic(uniforms), target);
      break;
    case 'spawn-init':
      native.spawnInit(h, target);
      break;
    case 'spawn-ball':
      native.spawnBall(h, new Float32Array([+(uniforms.radius === undefined ? 1 : uniforms.radius),
        +(uniforms.speed || 0)]), target);
      break;
    case 'spawn-direct':
    case 'spawn-sample': {
      const f = new Float32Array(17);
      const size = uniforms.spawnSize || [1, 1];
      const jit = uniforms.jitter || [0, 0];
      const m = uniforms.spawnMatrix || [1, 0, 0, 0, 1, 0, 0, 0, 1];
      f[0] = size[0]; f[1] = size[1]; f[2] = jit[0]; f[3] = jit[1];
      f[4] = +(uniforms.time || 0); f[5] = +(uniforms.speed === undefined ? 1 : uniforms.speed);
      f[6] = +(uniforms.bias === undefined ? 1 : uniforms.bias); f[7] = +(uniforms.flowDecay || 0);
      for (let k = 0; k < 9; ++k) f[8 + k] = m[k];
      if (uniforms.spawnData.bindFor) uniforms.spawnData.bindFor(particles);   // the spawner's own image buffer
      let source = uniforms.spawnData.sourceIndex();
      if (source >= 0 && target === native.TARGET_RING) {
        // ring indices are resolved by the library after utils.step() rotated the ring
        source = (source + 1) % particles.buffers.length;
      }
      if (program.kind === 'spawn-direct') native.spawnDirect(h, f, source, target);
      else native.spawnSample(h, f, program.fixed.samples, program.fixed.apply, source, target);
      break;
    }
    default:
      throw new Error(`unknown program kind ${program.kind}`);
  }
}

class Particles {
  constructor(gl, options) {
    const params = { ...defaults(), ...options };

    this.gl = gl;
    this.shape = params.shape;
    this.geomShape = (params.geomShape || [...this.shape]);
    this.logic = (params.logic || new Program('logic'));
    this.render = params.render;
    this.buffers = [];
    // src/particles.js:77-78: host staging, ndarray-like {data, shape:[w,h,4]} with pixels[x][y]
    this.pixels = {
      data: new Float32Array(this.shape[0] * this.shape[1] * 4),
      shape: [this.shape[0], this.shape[1], 4]
    };
    this.nextId = 0;
    this.handle = native.create(params.device | 0, this.shape[0], this.shape[1],
      params.globalHeight | 0, params.row0 | 0, 0, params.mode | 0, params.stateFormat | 0);
  }

  setup(numBuffers = 1) {                       // src/particles.js:81-92
    native.setup(this.handle, numBuffers);
    while (this.buffers.length < numBuffers) this.buffers.push(new StateBuffer(this, this.nextId++));
    while (this.buffers.length > numBuffers) this.buffers.pop().dispose();
  }

  spawn(map, pixels = this.pixels, offset = [0, 0]) {   // src/particles.js:94-117
    const data = new Float32Array(4);
    const [w, h] = pixels.shape;
    const px = pixels.data;
    let i = 0;

    for (let x = 0; x < w; ++x) {
      for (let y = 0; y < h; ++y) {
        data[0] = data[1] = data[2] = data[3] = 0;
        map(data, x, y);
        px[i++] = data[0]; px[i++] = data[1]; px[i++] = data[2]; px[i++] = data[3];
      }
    }

    // setPixels of an [w,h,4] ndarray: texel (x,y) <- pixels[x][y]; the library takes row-major texels
    const texels = new Float32Array(w * h * 4);
    for (let x = 0; x < w; ++x) {
      for (let y = 0; y < h; ++y) {
        const s = (x * h + y) * 4, d = (y * w + x) * 4;
        texels[d] = px[s]; texels[d + 1] = px[s + 1]; texels[d + 2] = px[s + 2]; texels[d + 3] = px[s + 3];
      }
    }
    native.uploadState(this.handle, -1, texels, offset[0], offset[1], w, h);
  }

  // readPixels(FLOAT) order: row-major texels
  uploadTexels(texels, buffer = -1) {
    native.uploadState(this.handle, buffer, texels, 0, 0, this.shape[0], this.shape[1]);
  }

  read(buffer = 0, out) {
    const index = ((buffer instanceof StateBuffer) ? buffer.index : buffer);
    const px = (out || new Float32Array(this.shape[0] * this.shape[1] * 4));
    native.downloadState(this.handle, index, px, 0, 0, this.shape[0], this.shape[1]);
    return px;
  }

  step(update, buffer) {                         // src/particles.js:123-145
    const target = ((!buffer) ? native.TARGET_RING
      : ((buffer instanceof StateBuffer) ? buffer.index : buffer.targetIndex()));

    const uniforms = Particles.applyUpdate(Object.assign(this.logic.uniforms, {
      dataRes: this.shape,
      geomRes: this.geomShape
    }), update);

    runPass(this, this.logic, uniforms, target);

    if (!buffer) step(this.buffers);             // the library rotated its ring the same way
  }

  // n consecutive logic passes with a fixed-step timer (time_k = time0 + (k+1)*dtMs, accumulated in
  // double like src/timer.js:28-31), replayed from a captured hipGraph.  Extension: the reference
  // issues these one draw call at a time.
  stepN(update, time0, dtMs, n) {
    if (this.logic.kind !== 'logic') throw new Error('stepN runs the logic program only');
    const uniforms = Particles.applyUpdate(Object.assign(this.logic.uniforms, {
      dataRes: this.shape,
      geomRes: this.geomShape
    }), update);
    native.stepN(this.handle, packLogic(uniforms), time0, dtMs, n);
    for (let k = 0; k < (n % Math.max(this.buffers.length, 1)); ++k) step(this.buffers);
  }

  draw() {}                                       // no display on this path (src/particles.js:147-158)

  updateLogic(logic) { this.logic = ((logic instanceof Program) ? logic : new Program('logic')); }
  updateRender() {}

  sync() { native.sync(this.handle); }
  // a switch between equivalent paths of the library (th_option_set / _get; no switch changes a result):
  // option('bucket') reads, option('bucket', 1) sets and returns the value
  option(name, value) {
    const key = native['OPT_' + name.replace(/[A-Z]/g, (ch) => '_' + ch).toUpperCase()];
    if (key === undefined) throw new Error('unknown option ' + name);
    return (value === undefined) ? native.option(this.handle, key) : native.option(this.handle, key, +value);
  }
  stats(speedLimit) { return native.stats(this.handle, speedLimit); }

  // -- one Node process per GPU (row-band shards; build-defined: the reference is one WebGL context) --------------
  // Rank 0 makes the communicator id (Particles.commUniqueId()), the application hands its 128 bytes to the other
  // ranks (a file, a socket, an environment variable), every rank joins with commInit(id, rank, world); from then on
  // statsGlobal() is the job's counter block - the library's RCCL all-reduce on the context's stream.
  static commUniqueId() { return native.commUniqueId(); }
  // ... or the id of an in-process world (contexts of this process as ranks; a test transport: see include/tendrils_hip.h)
  static commLoopbackId() { return native.commLoopbackId(); }
  commInit(id, rank, world) { native.commInit(this.handle, id, rank | 0, world | 0); return this; }
  commDestroy() { native.commDestroy(this.handle); return this; }
  commQuery() { return native.commQuery(this.handle); }
  statsGlobal(speedLimit) { return native.statsGlobal(this.handle, speedLimit); }

  dispose() {
    if (this.handle) { native.destroy(this.handle); this.handle = null; }
  }

  static generateLUT(shape) {                    // src/particles.js:171-190
    const data = new Float32Array(shape[0] * shape[1] * 2);
    let k = 0;
    const w = Math.max(shape[0], 2);
    const h = Math.max(shape[1], 2);
    const invX = 1 / (w - 1);
    const invY = 1 / (h - 1);

    for (let i = 0; i < w; ++i) {
      for (let j = 0; j < h; ++j) {
        data[k++] = i * invX;
        data[k++] = j * invY;
      }
    }
    return data;
  }

  static applyUpdate(state, update) {            // src/particles.js:192-195
    return ((typeof update === 'function') ? update(state) : Object.assign(state, update));
  }
}

module.exports = { defaults, Particles, Program, StateBuffer, runPass, default: Particles };
