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
    var out = [];
    for (var i = 0; i < job.ops.length; ++i) {
      var op = job.ops[i];
      if (op[0] === 'track') {
        var tl = player.tracks[op[1]];
        tl[op[2]].apply(tl, op.slice(3).map(function (a) { return (a && typeof a === 'object' && !Array.isArray(a)) ? fix(a) : a; }));
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
    return {out: out, frames: frames, start: player.start(), end: player.end()};
  }

  window.Plotly = {
    version: '2.0.0',
    toImage: function (fig) {
      var res;
      try {
        var job = fig.layout.job;
        if (job.kind === 'animate') res = runAnimate(job);
        else res = {error: 'unknown job kind'};
      } catch (e) {
        res = {error: String(e), stack: e && e.stack};
      }
      return Promise.resolve(JSON.stringify(res));
    }
  };
})();
