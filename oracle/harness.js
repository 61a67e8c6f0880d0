/*
 * TEST INFRASTRUCTURE - oracle harness (runs ONLY in the build container).
 *
 * Appended (at fixture-generation time, in /tmp) after the reference's own
 * prebuilt UMD bundle /root/reference/docs/js/index.js and loaded by kaleido's
 * headless Chromium (SwiftShader software WebGL 1) in place of plotly.js.
 * It drives the reference's *own* Tendrils/Particles objects and compiled
 * shaders on caller-supplied textures and returns readPixels(FLOAT) output.
 * Nothing in here restates the reference's arithmetic: it is pure plumbing.
 *
 * Job kinds (fig.layout.job):
 *   {kind:'probe'}                       -> renderer / extension info
 *   {kind:'logic', ...}                  -> reference Tendrils.step() K times
 *   {kind:'deposit', ...}                -> reference Tendrils.draw(): the particle
 *                                           lines (previous -> current) blended
 *                                           into the flow FBO
 *   {kind:'spawn_map', ...}              -> reference Particles.spawn(map, pixels, offset) with a
 *                                           position-dependent map on a non-square staging
 *                                           array, every ring buffer read back
 *   {kind:'timer', ...}                  -> a scripted sequence of operations on the reference's
 *                                           own Timer class (tendrils.timer.constructor)
 *   {kind:'buffers', ...}                -> a script on the reference's Tendrils.buffers surface (setupBuffers, draw,
 *                                           copyBuffer, drawBuffer, stepBuffers ...), the images it reads on the way
 *   {kind:'shader', ...}                 -> one full-screen pass of a compiled
 *                                           reference shader string (spawners,
 *                                           optical flow) handed in by python
 */
(function () {
  function b64ToBytes(s) {
    var bin = atob(s), n = bin.length, out = new Uint8Array(n);
    for (var i = 0; i < n; ++i) out[i] = bin.charCodeAt(i);
    return out;
  }
  function bytesToB64(u8) {
    var parts = [], CH = 0x8000;
    for (var i = 0; i < u8.length; i += CH)
      parts.push(String.fromCharCode.apply(null, u8.subarray(i, Math.min(i + CH, u8.length))));
    return btoa(parts.join(''));
  }
  function f32FromB64(s) { return new Float32Array(b64ToBytes(s).buffer); }
  function f32ToB64(f) { return bytesToB64(new Uint8Array(f.buffer, f.byteOffset, f.byteLength)); }

  function getGL(w, h, T, settings) {
    var c = document.createElement('canvas');
    c.width = w; c.height = h;
    var gl = c.getContext('webgl', settings || (T ? T.glSettings : {preserveDrawingBuffer: true}));
    if (!gl) throw new Error('no webgl');
    if (!gl.getExtension('OES_texture_float')) throw new Error('no OES_texture_float');
    gl.getExtension('WEBGL_color_buffer_float');
    gl.getExtension('EXT_float_blend');
    return gl;
  }

  function uploadF32(gl, handle, w, h, f32) {
    gl.bindTexture(gl.TEXTURE_2D, handle);
    gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA, w, h, 0, gl.RGBA, gl.FLOAT, f32);
  }

  function readFBO(gl, w, h) {
    var px = new Float32Array(4 * w * h);
    gl.readPixels(0, 0, w, h, gl.RGBA, gl.FLOAT, px);
    return px;
  }

  // ---- reference Tendrils.step() driven on supplied textures ---------------
  function runLogic(job) {
    var T = window.Tendrils, N = job.N;
    var gl = getGL(job.viewW, job.viewH, T);
    var t = new T.Tendrils(gl, {});
    t.resize();              // viewRes <- canvas; viewSize <- coverAspect; flow.shape <- viewRes
    t.setup(N);              // reference default: 2 state buffers, inert spawn
    if (job.flowW) t.flow.shape = [job.flowW, job.flowH];
    if (job.viewSize) { t.viewSize[0] = job.viewSize[0]; t.viewSize[1] = job.viewSize[1]; }
    var k;
    for (k in (job.state || {})) t.state[k] = job.state[k];

    // state either supplied, or generated here from an integer hash (large N: too big to ship as base64);
    // the same generator lives in tests/helpers.py:hashed_state
    var st;
    if (job.inputs.stateGen) {
      var seed = job.inputs.stateGen.seed >>> 0, inertMod = job.inputs.stateGen.inertMod >>> 0;
      st = new Float32Array(4 * N * N);
      var hash = function (v) {           // 32-bit integer mix (Math.imul keeps it exact)
        v = Math.imul(v ^ (v >>> 16), 0x7feb352d) >>> 0;
        v = Math.imul(v ^ (v >>> 15), 0x846ca68b) >>> 0;
        return (v ^ (v >>> 16)) >>> 0;
      };
      for (var q = 0; q < N * N; ++q) {
        var b = hash((q * 4 + seed) >>> 0);
        if (inertMod && (b % inertMod) === 0) { st[4 * q] = -1000000; st[4 * q + 1] = -1000000; st[4 * q + 2] = 0; st[4 * q + 3] = 0; continue; }
        // 24-bit mantissas so that every value is exactly representable in fp32
        st[4 * q] = ((hash((q * 4 + seed) >>> 0) >>> 8) - 8388608) / 8388608;                 // pos.x in [-1, 1)
        st[4 * q + 1] = ((hash((q * 4 + 1 + seed) >>> 0) >>> 8) - 8388608) / 8388608;         // pos.y
        st[4 * q + 2] = ((hash((q * 4 + 2 + seed) >>> 0) >>> 8) - 8388608) / 838860800;       // vel.x in [-.01, .01)
        st[4 * q + 3] = ((hash((q * 4 + 3 + seed) >>> 0) >>> 8) - 8388608) / 838860800;       // vel.y
      }
    } else {
      st = f32FromB64(job.inputs.state);
    }
    for (var b = 0; b < t.particles.buffers.length; ++b)
      uploadF32(gl, t.particles.buffers[b].color[0].handle, N, N, st);
    if (job.inputs.flow)
      uploadF32(gl, t.flow.color[0].handle, t.flow.shape[0], t.flow.shape[1], f32FromB64(job.inputs.flow));
    if (job.inputs.targets)
      uploadF32(gl, t.targets.color[0].handle, N, N, f32FromB64(job.inputs.targets));

    t.timer.time = job.time0;
    if (job.rate != null) t.timer.rate = job.rate;
    var outs = [], times = [], dts = [];
    var rows = job.rows || null;   // optional [y0, y1) bands to return instead of everything
    var t0 = performance.now();
    for (var s = 0; s < job.steps; ++s) {
      t.timer.tick();
      t.step();
      if (job.draw) t.draw();             // the reference's frame: step() then draw() (flow deposit + view render)
      times.push(t.timer.time); dts.push(t.timer.dt);
      if (job.returnEach || s === job.steps - 1) {
        t.particles.buffers[0].bind();
        if (rows) {
          var bands = [];
          for (var r = 0; r < rows.length; ++r) {
            var px = new Float32Array(4 * N * (rows[r][1] - rows[r][0]));
            gl.readPixels(0, rows[r][0], N, rows[r][1] - rows[r][0], gl.RGBA, gl.FLOAT, px);
            bands.push(f32ToB64(px));
          }
          outs.push(bands);
        } else {
          outs.push(f32ToB64(readFBO(gl, N, N)));
        }
      }
    }
    var ms = performance.now() - t0;
    var flowOut = null;
    if (job.draw) { t.flow.bind(); flowOut = f32ToB64(readFBO(gl, t.flow.shape[0], t.flow.shape[1])); }
    return {out: outs, times: times, dts: dts, ms: ms, flowOut: flowOut,
            viewSize: [t.viewSize[0], t.viewSize[1]], viewRes: [t.viewRes[0], t.viewRes[1]],
            flowShape: [t.flow.shape[0], t.flow.shape[1]],
            state: t.state, err: gl.getError()};
  }

  // ---- reference Tendrils.draw(): flow deposit ---------------------------------
  function runDeposit(job) {
    var T = window.Tendrils, N = job.N;
    // job.view: also return the view render of draw() (the default framebuffer, RGBA8) - on a context without
    // multisampling, so that the view's lines are rasterised by the same rules as the flow pass's
    var gl = getGL(job.viewW, job.viewH, T, job.view ? {preserveDrawingBuffer: true, antialias: false, alpha: true, premultipliedAlpha: false} : null);
    var t = new T.Tendrils(gl, {});
    t.resize();
    t.setup(N);
    if (job.viewSize) { t.viewSize[0] = job.viewSize[0]; t.viewSize[1] = job.viewSize[1]; }
    var k;
    for (k in (job.state || {})) t.state[k] = job.state[k];
    uploadF32(gl, t.particles.buffers[0].color[0].handle, N, N, f32FromB64(job.inputs.current));
    uploadF32(gl, t.particles.buffers[1].color[0].handle, N, N, f32FromB64(job.inputs.previous));
    if (job.inputs.flow)
      uploadF32(gl, t.flow.color[0].handle, t.flow.shape[0], t.flow.shape[1], f32FromB64(job.inputs.flow));
    t.timer.time = job.time;
    // GL state as Tendrils.step() leaves it (src/index.js:267-268) - draw() follows step() in the frame loop
    gl.enable(gl.BLEND);
    gl.blendFunc(gl.SRC_ALPHA, gl.ONE_MINUS_SRC_ALPHA);
    t.draw();
    var view = null;
    if (job.view) {
      gl.bindFramebuffer(gl.FRAMEBUFFER, null);
      var px = new Uint8Array(4 * job.viewW * job.viewH);
      gl.readPixels(0, 0, job.viewW, job.viewH, gl.RGBA, gl.UNSIGNED_BYTE, px);
      view = bytesToB64(px);
    }
    t.flow.bind();
    var out = f32ToB64(readFBO(gl, t.flow.shape[0], t.flow.shape[1]));
    return {out: out, view: view, samples: gl.getParameter(gl.SAMPLES), lineWidthRange: Array.prototype.slice.call(gl.getParameter(gl.ALIASED_LINE_WIDTH_RANGE)),
            lineWidth: gl.getParameter(gl.LINE_WIDTH),
            viewSize: [t.viewSize[0], t.viewSize[1]], viewRes: [t.viewRes[0], t.viewRes[1]],
            flowShape: [t.flow.shape[0], t.flow.shape[1]], state: t.state, err: gl.getError()};
  }

  // ---- one full-screen pass of a compiled reference shader -----------------
  function compile(gl, type, src) {
    var sh = gl.createShader(type);
    gl.shaderSource(sh, src); gl.compileShader(sh);
    if (!gl.getShaderParameter(sh, gl.COMPILE_STATUS)) throw new Error(gl.getShaderInfoLog(sh));
    return sh;
  }
  function makeTex(gl, spec) {
    var tex = gl.createTexture();
    gl.activeTexture(gl.TEXTURE0 + 7);      // scratch unit: creation must not rebind a sampler's unit
    gl.bindTexture(gl.TEXTURE_2D, tex);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.NEAREST);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.NEAREST);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_S, gl.CLAMP_TO_EDGE);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_T, gl.CLAMP_TO_EDGE);
    if (spec.type === 'u8') {
      gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA, spec.w, spec.h, 0, gl.RGBA, gl.UNSIGNED_BYTE,
                    spec.data ? b64ToBytes(spec.data) : null);
    } else {
      gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA, spec.w, spec.h, 0, gl.RGBA, gl.FLOAT,
                    spec.data ? f32FromB64(spec.data) : null);
    }
    return tex;
  }
  function runShader(job) {
    var gl = getGL(job.outW, job.outH, null);
    var prog = gl.createProgram();
    gl.attachShader(prog, compile(gl, gl.VERTEX_SHADER, job.vert));
    gl.attachShader(prog, compile(gl, gl.FRAGMENT_SHADER, job.frag));
    gl.bindAttribLocation(prog, 0, 'position');
    gl.linkProgram(prog);
    if (!gl.getProgramParameter(prog, gl.LINK_STATUS)) throw new Error(gl.getProgramInfoLog(prog));
    gl.useProgram(prog);

    // render target: RGBA32F, optionally pre-filled (for the blended optical-flow pass)
    var dst = makeTex(gl, {type: 'f32', w: job.outW, h: job.outH, data: job.dst || null});
    var fbo = gl.createFramebuffer();
    gl.bindFramebuffer(gl.FRAMEBUFFER, fbo);
    gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0, gl.TEXTURE_2D, dst, 0);
    if (gl.checkFramebufferStatus(gl.FRAMEBUFFER) !== gl.FRAMEBUFFER_COMPLETE) throw new Error('fbo incomplete');
    if (!job.dst) { gl.clearColor(0, 0, 0, 0); gl.clear(gl.COLOR_BUFFER_BIT); }

    var unit = 0, name;
    for (name in (job.textures || {})) {
      var tex = makeTex(gl, job.textures[name]);
      gl.activeTexture(gl.TEXTURE0 + unit);
      gl.bindTexture(gl.TEXTURE_2D, tex);
      gl.uniform1i(gl.getUniformLocation(prog, name), unit);
      ++unit;
    }
    for (name in (job.uniforms || {})) {
      var v = job.uniforms[name], loc = gl.getUniformLocation(prog, name);
      if (loc === null) continue;
      if (typeof v === 'number') gl.uniform1f(loc, v);
      else if (v.length === 2) gl.uniform2f(loc, v[0], v[1]);
      else if (v.length === 3) gl.uniform3f(loc, v[0], v[1], v[2]);
      else if (v.length === 4) gl.uniform4f(loc, v[0], v[1], v[2], v[3]);
      else if (v.length === 9) gl.uniformMatrix3fv(loc, false, new Float32Array(v));
    }

    if (job.blend) { gl.enable(gl.BLEND); gl.blendFunc(gl.SRC_ALPHA, gl.ONE_MINUS_SRC_ALPHA); }
    else gl.disable(gl.BLEND);

    var vb = gl.createBuffer();
    gl.bindBuffer(gl.ARRAY_BUFFER, vb);
    // gl-big-triangle geometry, or caller-supplied triangles (GeometrySpawner's draw, src/spawn/geometry/index.js:103-115)
    var verts = job.positions ? new Float32Array(job.positions) : new Float32Array([-1, -1, -1, 4, 4, -1]);
    gl.bufferData(gl.ARRAY_BUFFER, verts, gl.STATIC_DRAW);
    gl.enableVertexAttribArray(0);
    gl.vertexAttribPointer(0, 2, gl.FLOAT, false, 0, 0);
    gl.viewport(0, 0, job.outW, job.outH);
    gl.drawArrays(gl.TRIANGLES, 0, verts.length / 2);
    return {out: f32ToB64(readFBO(gl, job.outW, job.outH)), err: gl.getError()};
  }

  // ---- reference Particles.spawn(map, pixels, offset) --------------------------
  // map(data, x, y) = (a0 + ax*x + ay*y, b0 + bx*x + by*y, x, y): affine in the loop indices, with coefficients that
  // are exact in fp32 for the sizes used - every staging cell, and so every texel, gets a value that names its (x, y)
  function runSpawnMap(job) {
    var T = window.Tendrils, N = job.N;
    var gl = getGL(job.viewW, job.viewH, T);
    var t = new T.Tendrils(gl, {});
    t.resize();
    t.setup(N);                                   // 2 ring buffers, all inert
    var P = t.particles, c = job.coef;
    var map = function (data, x, y) {
      data[0] = c[0] + c[1] * x + c[2] * y;
      data[1] = c[3] + c[4] * x + c[5] * y;
      data[2] = x;
      data[3] = y;
    };
    if (job.pixels) {
      var w = job.pixels[0], h = job.pixels[1];
      // an ndarray like particles.pixels (shape [w, h, 4], row-major strides) made by the bundle's own ndarray class
      var px = new P.pixels.constructor(new Float32Array(w * h * 4), w, h, 4, h * 4, 4, 1, 0);
      P.spawn(map, px, job.offset || [0, 0]);
    } else P.spawn(map);
    var out = [];
    for (var b = 0; b < P.buffers.length; ++b) {
      P.buffers[b].bind();
      out.push(f32ToB64(readFBO(gl, N, N)));
    }
    return {out: out, shape: [P.pixels.shape[0], P.pixels.shape[1], P.pixels.shape[2]], err: gl.getError()};
  }

  // ---- the reference's Timer, scripted ------------------------------------------
  function runTimer(job) {
    var T = window.Tendrils;
    var gl = getGL(4, 4, T);
    var Timer = new T.Tendrils(gl, {}).timer.constructor;
    var tm = null, out = [];
    for (var i = 0; i < job.ops.length; ++i) {
      var op = job.ops[i];
      if (op[0] === 'new') tm = new Timer(op[1], op[2]);
      else if (op[0] === 'set') tm[op[1]] = op[2];
      else if (op[0] === 'tick') tm.tick(op[1]);
      else if (op[0] === 'seek') tm.seek(op[1]);
      else if (op[0] === 'scrub') tm.scrub(op[1]);
      else if (op[0] === 'reset') tm.reset(op[1], op[2]);
      else throw new Error('unknown timer op ' + op[0]);
      out.push([tm.time, tm.dt, tm.offset, tm.since, tm.paused ? 1 : 0, tm.now(op[0] === 'tick' ? op[1] : 12345)]);
    }
    return {out: out};
  }

  // ---- Tendrils.buffers: the reference's own setupBuffers / draw / copyBuffer / drawBuffer / stepBuffers, scripted -------
  // job.ops: ['draw'] | ['stepBuffers'] | ['drawFade'] | ['drawFill', rgba] | ['clearView'] | ['copyBuffer', i] |
  //          ['drawBuffer', i | null] | ['bind', i | -1] | ['setupBuffers', n] | ['set', key, value] | ['tickStep'] |
  //          ['read', i | -1]  (-1 = the screen; a read binds what it reads and leaves it bound - the script says so)
  function runBuffers(job) {
    var T = window.Tendrils, N = job.N;
    var gl = getGL(job.viewW, job.viewH, T, {preserveDrawingBuffer: true, antialias: false, alpha: true, premultipliedAlpha: false});
    var t = new T.Tendrils(gl, {numBuffers: job.numBuffers});
    t.resize();
    t.setup(N);
    var k;
    for (k in (job.state || {})) t.state[k] = job.state[k];
    uploadF32(gl, t.particles.buffers[0].color[0].handle, N, N, f32FromB64(job.inputs.current));
    uploadF32(gl, t.particles.buffers[1].color[0].handle, N, N, f32FromB64(job.inputs.previous));
    t.timer.time = job.time;
    gl.enable(gl.BLEND);
    gl.blendFunc(gl.SRC_ALPHA, gl.ONE_MINUS_SRC_ALPHA);
    var reads = [], lengths = [];
    job.ops.forEach(function (op) {
      var what = op[0];
      if (what === 'draw') t.draw();
      else if (what === 'tickStep') { t.timer.tick(); t.step(); }
      else if (what === 'stepBuffers') t.stepBuffers();
      else if (what === 'drawFade') t.drawFade();
      else if (what === 'drawFill') t.drawFill(op[1]);
      else if (what === 'clearView') t.clearView();
      else if (what === 'copyBuffer') t.copyBuffer(op[1]);
      else if (what === 'drawBuffer') { if (op[1] === null) t.drawBuffer(); else t.drawBuffer(op[1]); }
      else if (what === 'setupBuffers') { t.setupBuffers(op[1]); t.resize(); }
      else if (what === 'set') t.state[op[1]] = op[2];
      else if (what === 'viewport') t.viewport();
      else if (what === 'bind') { if (op[1] < 0) gl.bindFramebuffer(gl.FRAMEBUFFER, null); else t.buffers[op[1]].bind(); }
      else if (what === 'read') {
        if (op[1] < 0) gl.bindFramebuffer(gl.FRAMEBUFFER, null); else t.buffers[op[1]].bind();
        var px = new Uint8Array(4 * job.viewW * job.viewH);
        gl.readPixels(0, 0, job.viewW, job.viewH, gl.RGBA, gl.UNSIGNED_BYTE, px);
        reads.push(bytesToB64(px));
      } else throw new Error('unknown op ' + what);
      lengths.push(t.buffers.length);
    });
    return {reads: reads, lengths: lengths, samples: gl.getParameter(gl.SAMPLES), viewSize: [t.viewSize[0], t.viewSize[1]],
            state: t.state, time: t.timer.time, err: gl.getError()};
  }

  window.Plotly = {
    version: '2.0.0',
    toImage: function (fig) {
      var res;
      try {
        var job = fig.layout.job;
        if (job.kind === 'probe') {
          var gl = getGL(4, 4, window.Tendrils);
          var dbg = gl.getExtension('WEBGL_debug_renderer_info');
          var pf = gl.getShaderPrecisionFormat(gl.FRAGMENT_SHADER, gl.HIGH_FLOAT);
          res = {keys: Object.keys(window.Tendrils),
                 renderer: dbg ? gl.getParameter(dbg.UNMASKED_RENDERER_WEBGL) : gl.getParameter(gl.RENDERER),
                 highp: [pf.rangeMin, pf.rangeMax, pf.precision],
                 maxTex: gl.getParameter(gl.MAX_TEXTURE_SIZE),
                 threads: navigator.hardwareConcurrency};
        } else if (job.kind === 'logic') res = runLogic(job);
        else if (job.kind === 'deposit') res = runDeposit(job);
        else if (job.kind === 'spawn_map') res = runSpawnMap(job);
        else if (job.kind === 'timer') res = runTimer(job);
        else if (job.kind === 'shader') res = runShader(job);
        else if (job.kind === 'buffers') res = runBuffers(job);
        else res = {error: 'unknown job kind'};
      } catch (e) {
        res = {error: String(e), stack: e && e.stack};
      }
      return Promise.resolve(JSON.stringify(res));
    }
  };
})();
