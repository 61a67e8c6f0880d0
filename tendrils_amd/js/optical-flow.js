'use strict';
// Mirror of the reference's OpticalFlow (src/optical-flow/index.js:32-71): two RGBA8 frame
// buffers (view = buffers[0], last = buffers[1]), uniforms, update()/step()/setPixels()/resize().
// `render()` stands for the full-screen draw the demo issues after update()
// (src/demo.main.js:1107-1159): the HIP pass, alpha-blended into tendrils.flow.
const native = require('./native');
const { step } = require('./utils');

const defaults = () => ({
  options: { shader: null, buffers: [[[1, 1]], [[1, 1]]] },
  uniforms: { viewSize: [1, 1], scaleUV: [1, -1], offset: 1, lambda: 0.001, speed: 1, speedLimit: 1, time: 1 }
});

class OpticalFlow {
  constructor(tendrils, options, uniforms) {
    const base = defaults();
    this.tendrils = tendrils;
    this.buffers = [{ id: 0 }, { id: 1 }];
    this.uniforms = Object.assign(base.uniforms, uniforms);
    this.bound = { ...this.uniforms };
    this.shape = [1, 1];
  }

  get handle() { return this.tendrils.particles.handle; }

  update(uniforms) {                               // src/optical-flow/index.js:50-58
    this.bound = Object.assign({}, this.uniforms, uniforms);
    return this.bound;
  }

  render() {
    const b = this.bound;
    native.opticalFlow(this.handle, new Float32Array([b.viewSize[0], b.viewSize[1], b.scaleUV[0], b.scaleUV[1],
      b.offset, b.lambda, b.time, b.speed, b.speedLimit]));
  }

  step() { step(this.buffers); native.framesRotate(this.handle); }     // :60-62

  setPixels(pixels) {                              // :64-66, RGBA8 rows in texture order
    native.framesUpload(this.handle, pixels);
  }

  resize(size) {                                   // :68-70
    this.shape = [size[0] | 0, size[1] | 0];
    native.framesResize(this.handle, this.shape[0], this.shape[1]);
  }
}

module.exports = { defaults, OpticalFlow, default: OpticalFlow };
