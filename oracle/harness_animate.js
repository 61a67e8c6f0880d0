/*
 * TEST INFRASTRUCTURE - oracle harness for the animation classes (runs ONLY in the build container).
 *
 * Appended (at fixture-generation time, in /tmp) after the reference's prebuilt bundle /root/reference/docs/js/demo.js,
 * whose bootstrap has been told to hand out its module loader instead of starting the app (ref_runner.py replaces the
 * final `t(0)` of the webpack bootstrap by `t`): window.Tendrils is then the bundle's `require`.  The harness looks up
 * the compiled src/animate/index.js (Player) and src/animate/timeline.js (Timeline) modules by their text and drives the
 * reference's OWN classes through a scripted sequence.  Nothing in here restates their arithmetic.
 *
 * job: {kind:'animate', tracks:{name:[]...}, ops:[...]}
 *   ['track', name, method, ...args]   a Timeline method of the reference (to, smoothTo, flipTo, over, easeOver, ...);
 *                                      frames may carry call: ['label', ...] - turned into functions that log the label
 *   ['play', time] | ['seek', time] | ['playFrom', time, start]          on the Player
 *   ['query', name, method, ...args]   a Timeline method whose return value is recorded (gapAt, indexOf, minFrame, ...)
 *      {kind:'scene', ...}   see runScene below
 * returns, after every player op: the outputs of all tracks, each track's playhead (time, gap) and the call log.
 */
(function () {
  function findModule(req, needles) {
    var ids = Object.keys(req.m), hit = [];
    for (var i = 0; i < ids.length; ++i) {
      var src = Function.prototype.toString.call(req.m[ids[i]]), ok = true;
      for (var k = 0; k < needles.length; ++k) if (src.indexOf(needles[k]) < 0) { ok = false; break; }
      if (ok) hit.push(ids[i]);
    }
    if (hit.length !== 1) throw new Error('module lookup ' + JSON.stringify(needles) + ' matched ' + hit.length);
    return req(+hit[0]);
  }

  function runAnimate(job) {
    var req = window.Tendrils;
    if (typeof req !== 'function' || !req.m) throw new Error('the bundle did not hand out its module loader');
    var Player = findModule(req, ['playFrom', 'outputs', 'tracks']).default;
    var log = [];
    var fix = function (frame) {          // call labels -> functions
      if (frame && frame.call) {
        frame.call = frame.call.map(function (label) { return function () { log.push(label); }; });
      }
      return frame;
    };
    var outputs = {}, tracks = {}, k;
    for (k in job.tracks) { tracks[k] = job.tracks[k].map(fix); outputs[k] = (job.outputs && job.outputs[k]) || {}; }
    var player = new Player(tracks, outputs);
    var out = [], queries = [];
    for (var i = 0; i < job.ops.length; ++i) {
      var op = job.ops[i];
      if (op[0] === 'track') {
        var tl = player.tracks[op[1]];
        tl[op[2]].apply(tl, op.slice(3).map(function (a) { return (a && typeof a === 'object' && !Array.isArray(a)) ? fix(a) : a; }));
        continue;
      }
      if (op[0] === 'query') {            // a Timeline method whose RETURN VALUE is recorded (infinities spelled out)
        var qt = player.tracks[op[1]];
        var val = qt[op[2]].apply(qt, op.slice(3));
        queries.push(JSON.parse(JSON.stringify(val === undefined ? null : val, function (key, v) {
          return (v === Infinity ? 'inf' : (v === -Infinity ? '-inf' : (typeof v === 'number' && v !== v ? 'nan' : v)));
        })));
        continue;
      }
      if (op[0] === 'play') player.play(op[1]);
      else if (op[0] === 'seek') player.seek(op[1]);
      else if (op[0] === 'playFrom') player.playFrom(op[1], op[2]);
      else throw new Error('unknown op ' + op[0]);
      var heads = {};
      for (k in player.tracks) heads[k] = [player.tracks[k].time, player.tracks[k].gap, player.tracks[k].frames.length];
      out.push({outputs: JSON.parse(JSON.stringify(player.outputs)), heads: heads, calls: log.slice()});
    }
    var frames = {};
    for (k in player.tracks) frames[k] = player.tracks[k].frames.map(function (f) {
      return {time: (f.time === Infinity ? 'inf' : (f.time === -Infinity ? '-inf' : f.time)), ease: f.ease || null, to: (f.to === undefined ? null : f.to)};
    });
    var num = function (v) { return (v === Infinity ? 'inf' : (v === -Infinity ? '-inf' : (v !== v ? 'nan' : v))); };
    return {out: out, queries: queries, frames: frames, start: num(player.start()), end: num(player.end()), duration: num(player.duration())};
  }

  // ---- a keyframed scene: the reference's Player driving the reference's Tendrils ------------------------------------
  // job: {kind:'scene', N, viewW, viewH, state0:{...scalars}, colors0:{baseColor:[..],...}, particles: b64 f32 [N*N*4],
  //       time0, frames, ops:[['track', name, method, ...args]...], grab:[frame indices whose flow + view are returned],
  //       targets: (optional) b64 f32 [N*N*4] for tendrils.targets}
  // Every frame is the demo's loop body (src/demo.main.js:1027-1031, :1082): timer.tick(); player.play(time); step(); draw().
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
  function f32ToB64(f) { return bytesToB64(new Uint8Array(f.buffer, f.byteOffset, f.byteLength)); }

  function runScene(job) {
    var req = window.Tendrils;
    var Player = findModule(req, ['playFrom', 'outputs', 'tracks']).default;
    var T = findModule(req, ['spawnShader', 'flowDecay']);
    var N = job.N, c = document.createElement('canvas');
    c.width = job.viewW; c.height = job.viewH;
    // no multisampling: the view's lines are then rasterised by the same rules as the flow pass's (as in harness.js:runDeposit)
    var gl = c.getContext('webgl', {preserveDrawingBuffer: true, antialias: false, alpha: true, premultipliedAlpha: false});
    if (!gl || !gl.getExtension('OES_texture_float')) throw new Error('no float webgl');
    gl.getExtension('WEBGL_color_buffer_float');
    gl.getExtension('EXT_float_blend');
    var t = new T.Tendrils(gl, {});
    t.resize();
    t.setup(N);
    var k;
    for (k in (job.state0 || {})) t.state[k] = job.state0[k];
    for (k in (job.colors0 || {})) for (var q = 0; q < 4; ++q) t.state[k][q] = job.colors0[k][q];
    var st = new Float32Array(b64ToBytes(job.particles).buffer);
    for (var b = 0; b < t.particles.buffers.length; ++b) {
      gl.bindTexture(gl.TEXTURE_2D, t.particles.buffers[b].color[0].handle);
      gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA, N, N, 0, gl.RGBA, gl.FLOAT, st);
    }
    if (job.targets) {                 // tendrils.targets (src/index.js:105,207): what `target` > 0 pulls the particles towards
      gl.bindTexture(gl.TEXTURE_2D, t.targets.color[0].handle);
      gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA, N, N, 0, gl.RGBA, gl.FLOAT, new Float32Array(b64ToBytes(job.targets).buffer));
    }
    // the demo's track table (src/demo.main.js:836-857), restricted to what the particle path reads
    var outputs = {tendrils: t.state, baseColor: t.state.baseColor, flowColor: t.state.flowColor, fadeColor: t.state.fadeColor};
    var player = new Player({tendrils: [], baseColor: [], flowColor: [], fadeColor: []}, outputs);
    for (var i = 0; i < job.ops.length; ++i) {
      var op = job.ops[i], tl = player.tracks[op[1]];
      tl[op[2]].apply(tl, op.slice(3));
    }
    t.timer.time = job.time0;
    var snap = function () {
      var o = {}, s = t.state;
      for (var key in s) o[key] = (s[key] && s[key].length != null && typeof s[key] !== 'string') ? Array.prototype.slice.call(s[key]) : s[key];
      return o;
    };
    var states = [], times = [], dts = [], parts = [], flows = {}, views = {};
    for (var f = 0; f < job.frames; ++f) {
      t.timer.tick();
      player.play(t.timer.time);
      t.step();
      t.draw();
      times.push(t.timer.time); dts.push(t.timer.dt); states.push(snap());
      t.particles.buffers[0].bind();
      var px = new Float32Array(4 * N * N);
      gl.readPixels(0, 0, N, N, gl.RGBA, gl.FLOAT, px);
      parts.push(f32ToB64(px));
      if (job.grab.indexOf(f) >= 0) {
        t.flow.bind();
        var fw = t.flow.shape[0], fh = t.flow.shape[1], fpx = new Float32Array(4 * fw * fh);
        gl.readPixels(0, 0, fw, fh, gl.RGBA, gl.FLOAT, fpx);
        flows[f] = f32ToB64(fpx);
        gl.bindFramebuffer(gl.FRAMEBUFFER, null);
        var v8 = new Uint8Array(4 * job.viewW * job.viewH);
        gl.readPixels(0, 0, job.viewW, job.viewH, gl.RGBA, gl.UNSIGNED_BYTE, v8);
        views[f] = bytesToB64(v8);
      }
    }
    return {states: states, times: times, dts: dts, particles: parts, flows: flows, views: views,
            viewSize: [t.viewSize[0], t.viewSize[1]], flowShape: [t.flow.shape[0], t.flow.shape[1]],
            samples: gl.getParameter(gl.SAMPLES), err: gl.getError()};
  }

  window.Plotly = {
    version: '2.0.0',
    toImage: function (fig) {
      var res;
      try {
        var job = fig.layout.job;
        if (job.kind === 'animate') res = runAnimate(job);
        else if (job.kind === 'scene') res = runScene(job);
        else res = {error: 'unknown job kind'};
      } catch (e) {
        res = {error: String(e), stack: e && e.stack};
      }
      return Promise.resolve(JSON.stringify(res));
    }
  };
})();
