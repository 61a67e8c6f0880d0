'use strict';
// src/spawn/pixels/index.js:15-67 - PixelSpawner: samples a texture (`buffer`: tendrils.flow or a
// particle state buffer) to respawn particles.  Programs: flow-sample.frag (5 taps, apply/flow.glsl)
// data-sample.frag (2 taps, identity after the vignette pass), best-sample.frag (6 taps, colour apply after the
// vignette pass) and index.frag (direct: every particle from its own texel, colour apply).
const native = require('../native');
const { Program } = require('../particles');
const { aspect } = require('../utils');

const flowSampleFrag = () => new Program('spawn-sample', { samples: 5, apply: 0 });
const dataSampleFrag = () => new Program('spawn-sample', { samples: 2, apply: 1 });

const bestSampleFrag = () => new Program('spawn-sample', { samples: 6, apply: 2 });   // src/demo.main.js:457
const pixelsFrag = () => new Program('spawn-direct', { apply: 2 });                  // src/demo.main.js:456

// The spawner's own buffer: FBO(gl, [1, 1], {float: true}) (src/spawn/pixels/index.js:17,34-36) holding an RGBA
// image as float texels; the pixels travel to the device when a pass first uses them.
class ImageBuffer {
  constructor(shape = [1, 1]) {
    this._shape = [shape[0] | 0, shape[1] | 0];
    this._pixels = new Float32Array(this._shape[0] * this._shape[1] * 4);
    this._dirty = true;
    this.color = [this];
  }

  get shape() { return this._shape.slice(); }

  set shape(wh) {
    this._shape = [wh[0] | 0, wh[1] | 0];
    this._pixels = new Float32Array(this._shape[0] * this._shape[1] * 4);
    this._dirty = true;
  }

  // pixels: Float32Array (0..1) or Uint8Array/Uint8ClampedArray (converted as WebGL does for a float texture)
  setPixels(pixels, shape = this._shape) {
    this._shape = [shape[0] | 0, shape[1] | 0];
    const n = this._shape[0] * this._shape[1] * 4;
    if (pixels instanceof Float32Array) this._pixels = Float32Array.from(pixels.subarray(0, n));
    else {
      this._pixels = new Float32Array(n);
      const inv = Math.fround(255);
      for (let k = 0; k < n; ++k) this._pixels[k] = Math.fround(Math.fround(pixels[k]) / inv);
    }
    this._dirty = true;
    return this;
  }

  bindFor(particles) {
    if (this._dirty || this._boundTo !== particles) {
      native.spawnImageUpload(particles.handle, this._pixels, this._shape[0], this._shape[1]);
      this._dirty = false; this._boundTo = particles;
    }
  }

  sourceIndex() { return native.SOURCE_IMAGE; }
}

const defaults = () => ({ shader: null, buffer: null, spawnSize: [1, 1], jitterRad: 2, speed: 1, bias: 1 });

class PixelSpawner {
  constructor(gl, options) {
    const params = Object.assign(defaults(), options);
    this.gl = gl;
    this.shader = (params.shader || flowSampleFrag());
    this.buffer = (params.buffer || new ImageBuffer());
    this.speed = params.speed;
    this.bias = params.bias;
    this.jitterRad = params.jitterRad;
    this.jitter = [0, 0];
    this.spawnSize = params.spawnSize;
    this.spawnMatrix = [1, 0, 0, 0, 1, 0, 0, 0, 1];
  }

  update(uniforms) {                               // src/spawn/pixels/index.js:47-56
    return Object.assign(uniforms, {
      spawnData: this.buffer,
      spawnSize: this.spawnSize,
      spawnMatrix: this.spawnMatrix,
      speed: this.speed,
      jitter: aspect(this.jitter, uniforms.viewRes, this.jitterRad),
      bias: this.bias
    });
  }

  spawn(tendrils, update = this.update.bind(this), ...rest) {
    return tendrils.spawnShader(this.shader, update, ...rest);
  }

  setPixels(pixels, shape) { return this.buffer.color[0].setPixels(pixels, shape); }   // :62-64
}

module.exports = { defaults, PixelSpawner, ImageBuffer, flowSampleFrag, dataSampleFrag, bestSampleFrag, pixelsFrag,
                   default: PixelSpawner };
