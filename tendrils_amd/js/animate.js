'use strict';
// Keyframe timelines and the player that drives `tendrils.state` from them: the Node host's mirror of the reference's
// animation utilities (src/animate/timeline.js:49-398, src/animate/index.js:13-129, src/animate/tween.js:10-48,
// src/animate/join-curve.js:6-9, with the `lerp` / `bezier` / `clamp` packages they use).  Same class and method names,
// argument order and arithmetic (pinned value for value against the reference's own compiled classes by
// tests/test_animate_reference.py); see tendrils_amd/animate.py for the description of the pieces.

const isNumber = (v) => (typeof v === 'number');
const clamp = (v, lo, hi) => ((lo < hi) ? (v < lo ? lo : (v > hi ? hi : v)) : (v < hi ? hi : (v > lo ? lo : v)));
const lerp = (a, b, t) => a * (1 - t) + b * t;

// 1-D Bezier curve through control values: closed forms up to 4 points, de Casteljau beyond
function bezier(points, t) {
  const n = points.length;
  if (!n) throw new Error('Cannot create a interpolator with no elements');
  if (n === 1) return points[0];
  if (n === 2) return points[0] + (points[1] - points[0]) * t;
  const ut = 1 - t;
  if (n === 3) return (points[0] * ut + points[1] * t) * ut + (points[1] * ut + points[2] * t) * t;
  if (n === 4) {
    const a1 = points[1] * ut + points[2] * t;
    return ((points[0] * ut + points[1] * t) * ut + a1 * t) * ut + (a1 * ut + (points[2] * ut + points[3] * t) * t) * t;
  }
  let p = [];
  for (let r = 0; r < n - 1; ++r) p.push(points[r] * ut + points[r + 1] * t);
  while (p.length > 1) {
    const q = [];
    for (let r = 0; r < p.length - 1; ++r) q.push(p[r] * ut + p[r + 1] * t);
    p = q;
  }
  return p[0];
}

const joinCurve = (curve, align = 1) =>
  ((!curve || curve.length === 0) ? 0 : ((curve.length === 1) ? curve[0] : (curve[curve.length - 1] - curve[curve.length - 2]) * align));

// ---- tween ---------------------------------------------------------------------------------------------------
const tweenValue = (a, b, t, ease) => ((a === b || !isNumber(a)) ? b : lerp(a, b, (ease ? bezier(ease, t) : t)));
const tweenable = (k, values, defaults) => {
  const v = (values && values[k]);
  return (isNumber(v) ? v : (defaults && defaults[k]));
};
function tweenProps(a, b, t, ease, out = {}) {
  if (!b) return out;
  for (const k of Object.keys(b)) {
    const va = tweenable(k, a, out), vb = tweenable(k, b, out);
    out[k] = ((isNumber(va) && isNumber(vb)) ? tweenValue(va, vb, t, ease) : ((t < 1) ? va : vb));
  }
  return out;
}
const tween = (span, out) => tweenProps(span.a, span.b, span.t, span.ease, out);

// ---- timeline --------------------------------------------------------------------------------------------------
function makeFrame(to, time, ease, call) { return ((arguments.length > 1) ? { to, time, ease, call } : to); }
const after = (a, b) => (a.time > b.time);
function offset(a, b, time) {
  const lo = Math.min(a.time, b.time);
  return clamp(((time - lo) / (Math.max(a.time, b.time) - lo) || 0), 0, 1);
}
const within = (a, b, time) => (Math.min(a.time, b.time) < time && time <= Math.max(a.time, b.time));
function accumulate(frame, out) {
  out.apply = Object.assign((out.apply || {}), frame.to);
  if (frame.call && frame.call.length) (out.call || (out.call = [])).push(...frame.call);
  return out;
}

class Timeline {
  constructor(frames, infinite = true, rewind = false, symmetric = true) {
    this.frames = this.setup(frames, infinite);
    this.time = 0;
    this.gap = -1;
    this.span = undefined;
    this.symmetric = symmetric;
    this.infinite = infinite;
    this.rewind = rewind;
  }

  setup(frames = [], infinite = true) {
    const all = (infinite ? [{ time: -Infinity }, ...frames, { time: Infinity }] : [...frames]);
    const out = [];                                // ordered insertion: later frames of equal time stay behind
    for (const f of all) {
      let k = out.length;
      while (k > 0 && after(out[k - 1], f)) --k;
      out.splice(k, 0, f);
    }
    return (this.frames = out);
  }

  merge(frames) { frames.forEach((f) => this.add(f)); return frames; }
  insertFrame(f, frame) { this.frames.splice(f, 0, frame); return this; }

  add(...frame) {
    const adding = makeFrame(...frame);
    const f = this.indexOf(adding);
    this.insertFrame(f, adding);
    return f;
  }

  addSpan(duration, ...frame) {
    const f = this.add(...frame);
    const t0 = this.frames[f].time - duration;
    const past = this.frames[f - 1];
    if (duration && (!past || past.time < t0)) this.add(null, t0);
    return f;
  }

  seek(time) {
    if (this.valid() && within(this.span.past, this.span.next, time)) this.span.t = offset(this.span.past, this.span.next, time);
    else this.setTime(time);
    return this.span;
  }

  play(time) {
    const gap0 = Math.max(this.gap, 0.5);
    let span = this.seek(time);
    if (this.valid()) {
      const accumulated = {};
      const passed = this.gap - gap0;
      const skipped = Math.abs(passed);
      const dir = Math.sign(passed);
      const onwards = ((this.reverse ? -dir : dir) > 0);
      if (skipped > 0 && onwards) {
        const side = ((dir < 0) ? Math.floor : Math.ceil);
        for (let f = 0; f < skipped; ++f) accumulate(this.frames[side(gap0 + (f * dir))], accumulated);
      }
      span = { ...span, ...accumulated };
    }
    return span;
  }

  playFrom(time = this.time, start = 0) { this.seek(start); return this.play(time); }

  setTime(time) {
    const gap = this.gapAt(time);
    this.span = this.spanGapAt(time, gap, this.span);
    this.gap = gap;
    this.time = time;
    return this;
  }

  indexOf(frame) {
    const next = this.frames.findIndex((other) => after(other, frame));
    return ((next < 0) ? this.frames.length : next);
  }

  gapAt(time) {
    if (this.frames.length < 2) return -1;
    const next = this.frames.findIndex((frame) => frame.time >= time);
    return ((next < 0) ? this.frames.length - 1 : Math.max(next, 1)) - 0.5;
  }

  spanGapAt(time, gap = this.gapAt(time), out = {}) {
    if (gap < 0) return undefined;
    let past = this.frames[Math.floor(gap)], next = this.frames[Math.ceil(gap)];
    let ease = next.ease;
    if (this.rewind) {
      if (!this.symmetric) ease = past.ease;
      [past, next] = [next, past];
    }
    out.past = past; out.next = next;
    out.a = past.to; out.b = next.to;
    out.t = offset(past, next, time);
    out.ease = ease;
    return out;
  }

  to(...frame) { this.add(...frame); return this; }
  easeTo(align, ...frame) { this.easeJoin(this.add(...frame), align); return this; }
  smoothTo(...frame) { return this.easeTo(1, ...frame); }
  flipTo(...frame) { return this.easeTo(-1, ...frame); }
  over(duration, ...frame) { this.addSpan(duration, ...frame); return this; }
  easeOver(duration, align, ...frame) { this.easeJoin(this.addSpan(duration, ...frame), align); return this; }
  smoothOver(duration, ...frame) { return this.easeOver(duration, 1, ...frame); }
  flipOver(duration, ...frame) { return this.easeOver(duration, -1, ...frame); }

  easeJoin(f, align) {
    let ease = null;
    if (f > 0) {
      const frame = this.frames[f];
      ease = ((frame.ease && frame.ease.length) ? frame.ease : [0, 1]);
      ease.splice(1, 0, joinCurve(this.frames[f - 1].ease, align));
      frame.ease = ease;
    }
    return ease;
  }

  valid(gap = this.gap, span = this.span) { return (gap > 0 && span); }
  start() { return (this.frames.length ? this.frames[0].time : null); }
  end() { return (this.frames.length ? this.frames[this.frames.length - 1].time : null); }
  duration() { return (this.end() || 0) - (this.start() || 0); }
}

function apply(span, out = {}) {                   // src/animate/index.js:13-22
  if (span) {
    Object.assign(out, span.apply);
    tween(span, out);
    (span.call || []).forEach((f) => f(out, span));
  }
  return out;
}

class Player {
  constructor(tracks, outputs = {}) {
    this.tracks = tracks;
    this.outputs = outputs;
    this.add(this.tracks);
  }

  add(tracks) {
    for (const key of Object.keys(tracks)) {
      const track = tracks[key];
      this.tracks[key] = ((track instanceof Timeline) ? track : new Timeline(track));
    }
    return this;
  }

  each(f) { for (const key of Object.keys(this.tracks)) f(this.tracks[key], key); return this; }

  apply(f, out = this.outputs) {
    return this.each((track, key) => {
      const trackOut = (out[key] || (out[key] = {}));
      return apply(f(track, key, trackOut), trackOut);
    });
  }

  seek(time, out) { return this.apply((track) => track.seek(time), out); }
  play(time, out) { return this.apply((track) => track.play(time), out); }
  playFrom(time, start, out) { return this.apply((track) => track.playFrom(time, start), out); }

  start() { return Object.keys(this.tracks).reduce((acc, k) => Math.min(this.tracks[k].start(), acc), null); }
  end() { return Object.keys(this.tracks).reduce((acc, k) => Math.min(this.tracks[k].end(), acc), null); }
  duration() { return (this.end() || 0) - (this.start() || 0); }
}

module.exports = { Timeline, Player, apply, tween, tweenValue, tweenProps, bezier, lerp, clamp, joinCurve, makeFrame, default: Player };
