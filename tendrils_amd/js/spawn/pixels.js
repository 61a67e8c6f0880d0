'use strict';
// src/spawn/pixels/index.js:15-67 - PixelSpawner: samples a texture (`buffer`: tendrils.flow or a
// particle state buffer) to respawn particles.  Programs: flow-sample.frag (5 taps, apply/flow.glsl)
// and data-sample.frag (2 taps, identity after the vignette pass).
const { Program } = require('../particles');
const { aspect } = require('../utils');

const flowSampleFrag = () => new Program('spawn-sample', { samples: 5, apply: 0 });
const dataSampleFrag = () => new Program('spawn-sample', { samples: 2, apply: 1 });

const defaults = () => ({ shader: null, buffer: null, spawnSize: [1, 1], jitterRad: 2, speed: 1, bias: 1 });

class PixelSpawner {
  constructor(gl, options) {
    const params = Object.assign(defaults(), options);
    this.gl = gl;
    this.shader = (params.shader || flowSampleFrag());
    this.buffer = params.buffer;
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
}

module.exports = { defaults, PixelSpawner, flowSampleFrag, dataSampleFrag, default: PixelSpawner };
