'use strict';
// Mirror of the reference's facade `Tendrils` (src/index.js:84-457) for the particle-update
// path: state uniforms, timer, flow/targets textures, step(), spawn(), spawnShader(), resize().
// draw() runs the flow pass (particle lines into the flow field) and the view pass (the same lines into the RGBA8 view
// buffer) in one native call.
const native = require('./native');
const { Particles, Program } = require('./particles');
const { Timer } = require('./timer');
const { coverAspect, inert, step } = require('./utils');

const defaults = () => ({                         // src/index.js:28-75
  state: {
    rootNum: Math.pow(2, 9),
    autoClearView: false, autoFade: true,
    damping: 0.043, speedLimit: 0.01,
    forceWeight: 0.016, varyForce: -0.1,
    flowWeight: 1, varyFlow: 0.2,
    noiseWeight: 0.002, varyNoise: 0.3,
    flowDecay: 0.005, flowWidth: 5,
    noiseScale: 2.125, varyNoiseScale: 0.5,
    noiseSpeed: 0.00025, varyNoiseSpeed: 0.1,
    target: 0, varyTarget: 1,
    lineWidth: 1, speedAlpha: 0.000001, colorMapAlpha: 0.4,
    baseColor: [1, 1, 1, 0.5], flowColor: [1, 1, 1, 0.04], fadeColor: [0.1333, 0.1333, 0.1333, 0]
  },
  timer: Object.assign(new Timer(), { step: 1000 / 60 }),
  numBuffers: 0,
  logicShader: null,
  colorMap: null
});

const glSettings = { preserveDrawingBuffer: true, antialias: true };   // src/index.js:77-80

// tendrils.flow (src/index.js:102): RGBA32F, NEAREST, CLAMP_TO_EDGE, resizable
class FlowTexture {
  constructor(owner) { this.owner = owner; this._shape = [1, 1]; }
  get shape() { return [...this._shape]; }
  set shape(wh) {
    this._shape = [wh[0] | 0, wh[1] | 0];
    if (this.owner.particles) native.flowResize(this.owner.particles.handle, this._shape[0], this._shape[1]);
  }
  setPixels(texels) { native.flowUpload(this.owner.particles.handle, texels); }
  read(out) {
    const px = (out || new Float32Array(this._shape[0] * this._shape[1] * 4));
    native.flowDownload(this.owner.particles.handle, px);
    return px;
  }
  clear() { native.flowClear(this.owner.particles.handle); }
  sourceIndex() { return native.SOURCE_FLOW; }
}

// tendrils.targets (src/index.js:105,207)
class TargetsTexture {
  constructor(owner) { this.owner = owner; this.shape = [1, 1]; }
  setPixels(texels) { native.targetsUpload(this.owner.particles.handle, texels); }
  read(out) {
    const px = (out || new Float32Array(this.shape[0] * this.shape[1] * 4));
    native.targetsDownload(this.owner.particles.handle, px);
    return px;
  }
  clear() { native.targetsClear(this.owner.particles.handle); }
  targetIndex() { return native.TARGET_TARGETS; }
}

const initSpawner = (data) => {                   // src/spawn/init/cpu.js:3-8
  data[0] = data[1] = inert;
  data[2] = data[3] = 0;
  return data;
};

// One of Tendrils.buffers (src/index.js:172-177: `FBO(gl, [1, 1])`, given viewRes by resize()): an off-screen RGBA8 view
// image on the device, addressed by its place in the owner's ring
class ViewBuffer {
  constructor(owner) { this.owner = owner; this.shape = [1, 1]; }
  bind() { this.owner.bindView(this); return this; }
  read() { return this.owner.readView(this); }     // Uint8Array, viewRes[0] x viewRes[1] RGBA8
  dispose() { this.owner = null; }
}

class Tendrils {
  constructor(gl, options) {
    const params = { ...defaults(), ...options };

    // `gl` only needs drawingBufferWidth/Height here (what resize() reads)
    this.gl = (gl || { drawingBufferWidth: 1, drawingBufferHeight: 1 });
    this.state = params.state;
    this.flow = new FlowTexture(this);
    this.targets = new TargetsTexture(this);
    this.buffers = [];
    this.logicShader = null;
    this.uniforms = { render: {}, update: {} };
    this.particles = null;
    this.viewRes = [0, 0];
    this.viewSize = [0, 0];
    this.timer = params.timer;
    this.device = params.device | 0;
    this.mode = params.mode | 0;
    this.stateFormat = params.stateFormat | 0;
    // a row band of a larger texture (one process per GPU: DESIGN.md 6): rows [row0, row0 + rows) of globalHeight
    this.band = { row0: params.row0 | 0, rows: params.rows | 0, globalHeight: params.globalHeight | 0 };
    this.colorMap = (params.colorMap || null);     // { shape: [w, h], data: Float32Array } (null = the 1x1 zero texture)
    this.renderView = (params.renderView !== false);   // draw() also runs the view pass, as the reference's does
    // what gl.getParameter(gl.ALIASED_LINE_WIDTH_RANGE) reports here: [1, 1] like the GL the reference was captured on
    // (flowWidth: 5 then draws width-1 lines, as it does there); up to [1, 64] for the picture of a GL that honours widths
    this.lineWidthRange = (params.lineWidthRange || [1, 1]);
    this.bound = null;                             // the bound view image: null = the screen, else one of this.buffers
    this.setupBuffers(params.numBuffers);          // src/index.js:109
  }

  setup(...rest) { this.setupParticles(...rest); this.reset(); return this; }
  reset() { this.spawn(); return this; }

  dispose() {
    if (this.particles) { this.particles.dispose(); this.particles = null; }
    return this;
  }

  // ---- Tendrils.buffers: off-screen view images (src/index.js:172-184, 359-391)
  setupBuffers(numBuffers = 0) {                   // src/index.js:172-184
    while (this.buffers.length < numBuffers) this.buffers.push(new ViewBuffer(this));
    while (this.buffers.length > numBuffers) {
      const gone = this.buffers.pop();
      if (this.bound === gone) this.bound = null;  // (the library leaves the screen bound as well)
      gone.dispose();
    }
    if (this.particles) native.viewBuffers(this.particles.handle, this.buffers.length);
    return this;
  }

  bindView(buffer = null) {                        // gl.bindFramebuffer: null = the screen, else one of this.buffers
    this.bound = buffer;
    if (this.particles) native.viewBind(this.particles.handle, buffer ? this.buffers.indexOf(buffer) : -1);
  }

  drawBuffer(index) {                              // src/index.js:359-367: a buffer's contents to the screen
    this.bindView(null);
    if (this.state.autoClearView) native.viewClear(this.particles.handle);   // gl.clear of the bound framebuffer - the screen - alone
    return this.copyBuffer(index).stepBuffers();
  }

  copyBuffer(index = 0) {                          // src/index.js:370-383: into the current render target
    if (index < this.buffers.length && index >= 0) native.viewCopy(this.particles.handle, index | 0);
    return this;
  }

  stepBuffers() {                                  // src/index.js:385-391
    if (this.buffers.length > 1) {
      step(this.buffers);
      if (this.particles) native.viewStepBuffers(this.particles.handle);
    }
    return this;
  }

  viewport() { return this; }                      // src/index.js:410-419: gl.viewport(0, 0, ...viewRes) - every pass here covers its whole target

  setupParticles(rootNum = this.state.rootNum, numBuffers = 2) {   // src/index.js:186-210
    this.state.rootNum = rootNum;
    const shape = [rootNum, this.band.rows || rootNum];

    if (this.particles) this.particles.dispose();
    this.particles = new Particles(this.gl, {
      shape,
      geomShape: [shape[0], shape[1] * 2],
      logic: new Program('logic'),
      device: this.device,
      mode: this.mode,
      stateFormat: this.stateFormat,
      row0: this.band.row0,
      globalHeight: this.band.globalHeight || (this.band.rows ? rootNum : 0)
    });
    this.logicShader = this.particles.logic;
    this.particles.setup(numBuffers);
    this.targets.shape = shape;
    this.flow.shape = this.flow.shape;            // (re)create on the new context
    if (this.colorMap) native.colormapUpload(this.particles.handle, this.colorMap.data, this.colorMap.shape[0], this.colorMap.shape[1]);
    native.lineWidthRange(this.particles.handle, this.lineWidthRange[0], this.lineWidthRange[1]);
    this.bound = null;                             // (a new context: its screen is bound, its ring is empty)
    if (this.buffers.length) native.viewBuffers(this.particles.handle, this.buffers.length);
    return this;
  }

  // gl.lineWidth(Math.max(0, flowWidth)) before the flow pass, gl.lineWidth(Math.max(0, lineWidth)) before the view pass
  // (src/index.js:302,336); a width of 0 is GL's INVALID_VALUE: the width of that pass stays what it was
  lineWidths() {
    const flow = Math.max(0, this.state.flowWidth), view = Math.max(0, this.state.lineWidth);
    if (flow > 0) native.lineWidth(this.particles.handle, 0, flow);
    if (view > 0) native.lineWidth(this.particles.handle, 1, view);
  }

  clear() { this.clearView(); this.clearFlow(); return this; }
  clearView() {                                    // src/index.js:220-229: every buffer, then the screen - which it leaves bound
    this.buffers.forEach((buffer) => { this.bindView(buffer); native.viewClear(this.particles.handle); });
    this.bindView(null);
    native.viewClear(this.particles.handle);
    return this;
  }

  drawFade() {                                     // src/index.js:342-348
    if (this.state.fadeColor[3] > 0) this.drawFill(this.state.fadeColor);
    return this;
  }

  drawFill(color = this.state.fadeColor) {         // src/index.js:350-356
    native.viewFill(this.particles.handle, new Float32Array(color));
    return this;
  }

  // th_render_uniforms: viewSize, time, speedLimit, flowDecay, speedAlpha, colorMapAlpha, sin(time*flowDecay) - evaluated
  // here, on fp32 operands as the shader would: GLSL leaves its value to the implementation -, baseColor, flowColor
  renderUniforms() {
    const s = this.state;
    return new Float32Array([this.viewSize[0], this.viewSize[1], this.timer.time, s.speedLimit, s.flowDecay, s.speedAlpha,
      s.colorMapAlpha, Math.sin(Math.fround(this.timer.time) * Math.fround(s.flowDecay)), ...s.baseColor, ...s.flowColor]);
  }

  // the view buffer: Uint8Array, viewRes[0] x viewRes[1] RGBA8 in readPixels order
  // (the screen image, or one of this.buffers; what is bound stays bound)
  readView(buffer = null) {
    const was = this.bound;
    if (was !== buffer) this.bindView(buffer);
    const pixels = native.viewDownload(this.particles.handle);
    if (was !== buffer) this.bindView(was);
    return pixels;
  }

  setColorMap(texels, shape) {                     // tendrils.colorMap (src/index.js:94-96): RGBA32F texels, [w, h]
    this.colorMap = { shape: [shape[0], shape[1]], data: texels };
    if (this.particles) native.colormapUpload(this.particles.handle, texels, shape[0], shape[1]);
    return this;
  }
  clearFlow() { this.flow.clear(); return this; }   // src/index.js:231-236
  restart() { this.clear(); this.reset(); return this; }

  step() {                                         // src/index.js:248-272
    if (!this.timer.paused) {
      this.particles.logic = this.logicShader;

      Object.assign(this.uniforms.update, this.state, {
        dt: this.timer.dt,
        time: this.timer.time,
        start: this.timer.since,
        flow: this.flow,
        targets: this.targets,
        viewSize: this.viewSize,
        viewRes: this.viewRes
      });

      this.particles.step(this.uniforms.update);
    }
    return this;
  }

  // n x (timer.tick(); step()) for a fixed-step, unpaused timer, as one captured-graph replay
  stepN(n) {
    const tm = this.timer;
    if (tm.paused || tm.step < 0 || tm.end >= 0) {
      for (let k = 0; k < n; ++k) { tm.tick(); this.step(); }
      return this;
    }
    this.particles.logic = this.logicShader;
    const dt = tm.step * tm.rate;
    Object.assign(this.uniforms.update, this.state, {
      dt, time: tm.time, start: tm.since, flow: this.flow, targets: this.targets,
      viewSize: this.viewSize, viewRes: this.viewRes
    });
    this.particles.stepN(this.uniforms.update, tm.time, dt, n);
    for (let k = 0; k < n; ++k) tm.tick();
    return this;
  }

  // Trail export (build-defined): the (previous -> current) line list this frame's draw() is made of,
  // 12 floats per line: p0.xy, p1.xy (clip space), then both vertices' (vel.x, vel.y, time, alpha).
  exportLines(view = false) {                      // view: the view pass's vertex colours instead of the flow varyings
    if (view) return native.exportViewLines(this.particles.handle, this.renderUniforms());
    return native.exportLines(this.particles.handle,
      new Float32Array([this.viewSize[0], this.viewSize[1], this.timer.time, this.state.speedLimit]));
  }

  draw() {                                         // src/index.js:278-340: the flow pass, then the view pass
    const deposit = new Float32Array([this.viewSize[0], this.viewSize[1], this.timer.time, this.state.speedLimit]);
    this.lineWidths();
    if (this.band.rows && this.band.rows !== (this.band.globalHeight || this.state.rootNum)) {
      // a row band of a larger texture (one process per GPU): the passes' exchange is the library's, over the communicator
      // the ranks joined with particles.commInit() - every rank calls draw() together
      if (this.renderView) {
        this.bindView(this.buffers.length ? this.buffers[0] : null);
        if (this.state.autoClearView) this.clearView();
        if (this.state.autoFade) this.drawFade();
      }
      this.fragments = native.drawSharded(this.particles.handle, deposit, this.renderView ? this.renderUniforms() : null);
      this.viewFragments = this.renderView ? this.fragments : 0;
      return this;
    }
    if (!this.renderView) {
      this.fragments = native.flowDeposit(this.particles.handle, deposit);
      return this;
    }
    // (the clear and the fade only touch the view buffer; both passes draw the same lines: rasterised and sorted once
    // when they draw them with the same width).  The view goes to buffers[0] when there are buffers, else to the screen
    // (src/index.js:318-325) - unless autoClearView comes in between: clearView() leaves the SCREEN bound (src/index.js:226).
    this.bindView(this.buffers.length ? this.buffers[0] : null);
    if (this.state.autoClearView) this.clearView();
    if (this.state.autoFade) this.drawFade();
    this.fragments = this.viewFragments = native.draw(this.particles.handle, deposit, this.renderUniforms());
    return this;
  }

  resize() {                                       // src/index.js:393-408
    this.viewRes[0] = this.gl.drawingBufferWidth;
    this.viewRes[1] = this.gl.drawingBufferHeight;
    coverAspect(this.viewSize, this.viewRes);
    this.buffers.forEach((buffer) => { buffer.shape = [...this.viewRes]; });   // src/index.js:404 (the images follow the flow texture's shape)
    this.flow.shape = this.viewRes;
    return this;
  }

  spawn(spawner = initSpawner) {                   // src/index.js:425-429
    if (spawner === initSpawner) {
      // Particles.spawn(initSpawner) leaves every ring buffer inert: fill them on the device
      this.particles.buffers.forEach((b, k) => native.spawnInit(this.particles.handle, k));
    } else {
      this.particles.spawn(spawner);
    }
    return this;
  }

  spawnShader(shader, update, ...rest) {           // src/index.js:432-457
    this.timer.tick();                             // every GPU spawn advances time
    this.particles.logic = shader;

    this.particles.step(Particles.applyUpdate({
      ...this.state,
      time: this.timer.time,
      viewSize: this.viewSize,
      viewRes: this.viewRes
    }, update), ...rest);

    this.particles.logic = this.logicShader;
    return this;
  }
}

module.exports = { defaults, glSettings, Tendrils, Particles, Program, Timer, default: Tendrils };
