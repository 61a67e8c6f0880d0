'use strict';
// Headless scene replay (SURVEY.md 8f-4) for the Node host: presets and keyframed tracks driving `tendrils.state` while
// the particle path runs - the demo's colour proxy (src/demo.main.js:1335-1354), its track table (:836-857) and its loop
// body (:1027-1031, :1082) as a batch job.  See tendrils_amd/scenes.py; pinned by tests/golden/scene_*.npz.
const { Player } = require('./animate');

const COLOURS = ['base', 'flow', 'fade'];

// Colours as 0..255 rgb + an alpha apart; a preset assigns some of the six entries, the rest stays as the last preset
// left it; a state colour is [r / 255, g / 255, b / 255, alpha].
class ColourProxy {
  constructor(state) {
    this.rgb = {};
    this.alpha = {};
    for (const c of COLOURS) {
      this.rgb[c] = Array.from(state[c + 'Color']).slice(0, 3).map((v) => v * 255);
      this.alpha[c] = state[c + 'Color'][3];
    }
  }

  // Assign what `preset` names; returns the state colours it touched.
  take(preset) {
    const named = preset.colorProxy || {}, touched = {};
    for (const c of COLOURS) {
      const rgb = (c + 'Color') in named, alpha = (c + 'Alpha') in named;
      if (rgb) this.rgb[c] = Array.from(named[c + 'Color']);
      if (alpha) this.alpha[c] = named[c + 'Alpha'];
      if (rgb || alpha) touched[c + 'Color'] = this.colour(c);
    }
    return touched;
  }

  colour(c) { return this.rgb[c].map((v) => v / 255).concat([this.alpha[c]]); }
}

// Set a preset at once (what clicking it does in the demo).
function applyPreset(tendrils, preset, proxy = new ColourProxy(tendrils.state)) {
  proxy.take(preset);
  Object.assign(tendrils.state, preset.state || {});
  for (const c of COLOURS) proxy.colour(c).forEach((v, i) => { tendrils.state[c + 'Color'][i] = v; });
  return tendrils;
}

class Scene {
  constructor(tendrils) {
    const s = tendrils.state;
    this.t = tendrils;
    this.proxy = new ColourProxy(s);
    this.player = new Player({ tendrils: [], baseColor: [], flowColor: [], fadeColor: [] },
      { tendrils: s, baseColor: s.baseColor, flowColor: s.flowColor, fadeColor: s.fadeColor });
  }

  preset(preset) {
    applyPreset(this.t, preset, this.proxy);
    return this;
  }

  // Reach `preset` at `time` (ms): eased over the `duration` ms before it, or - duration 0 - from the key before.
  keyframe(preset, time, duration = 0, ease = null) {
    const targets = Object.assign({ tendrils: Object.assign({}, preset.state || {}) }, this.proxy.take(preset));
    for (const name of Object.keys(targets)) {
      const frame = { to: targets[name], time, ease: (ease ? Array.from(ease) : null) };
      if (duration) this.player.tracks[name].smoothOver(duration, frame);
      else this.player.tracks[name].smoothTo(frame);
    }
    return this;
  }

  // One pass of the demo's loop body.
  frame() {
    const t = this.t;
    t.timer.tick();
    this.player.play(t.timer.time);
    t.step().draw();
    return this;
  }

  run(frames, each = null, spawner = null) {
    if (spawner) spawner.spawn(this.t);
    for (let k = 0; k < frames; ++k) {
      this.frame();
      if (each) each(k, this.t);
    }
    return this;
  }
}

module.exports = { Scene, ColourProxy, applyPreset };
