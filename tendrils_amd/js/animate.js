'use strict';
// Keyframe tracks for the headless scene replay (SURVEY.md 8f-4): what a host needs to drive `tendrils.state` the way
// the reference's demo does with its `Player` / `Timeline` / `tween` (src/animate/index.js, timeline.js, tween.js;
// used at src/demo.main.js:836-865, :928-953, :1027-1031, :1267-1274).  The public names, argument order and every
// observable number are the reference's - pinned by tests/golden/animate_script.json, a script run on the reference's
// own compiled classes (values after every play / seek / playFrom, the playheads, edits made while playing, the return
// values of the queries) and by tests/golden/scene_*.npz (the reference's Player driving the reference's Tendrils).
//
// How it is built here:
//   * a track keeps its keys in time order in two arrays in lock-step - `stamps` (plain numbers, what gets bisected)
//     and `keys` (the records: the caller's own frame objects where it passed objects).  There are no end records: an
//     open-ended track answers for "before the first key" / "after the last key" with two shared constants, and the
//     reference's frame numbering (its -Infinity frame is number 0) is an offset applied at the surface (`base`);
//   * the playhead is a cached interval - the two stamps it lies between, the records at either end, the curve - plus
//     the reference's half-integer `gap`.  A seek that stays inside the cached interval only moves `t`; nothing is looked
//     up again until the time leaves it.  Edits do not touch the playhead, which is also what the reference's does:
//     a key added inside the cached interval is noticed when the time next leaves that interval;
//   * easing is one closed-form blend per number (`blend`), the curve evaluated by de Casteljau in the a*(1-u) + b*u
//     form of the `bezier` package (its two-point special case included), the mix as a*(1-w) + b*w (`lerp` package).

const BEFORE = Object.freeze({ time: -Infinity });
const AFTER = Object.freeze({ time: Infinity });

const numeric = (v) => (typeof v === 'number' || v instanceof Number);
const names = (any) => (any ? Object.keys(any) : []);

// ---- curves ------------------------------------------------------------------------------------------------------

// Value at `u` of the 1-D Bezier curve with the given control values.
function curveAt(points, u) {
  const n = points.length;
  if (n < 2) return (n ? points[0] : u);
  if (n === 2) return points[0] + (points[1] - points[0]) * u;
  const w = Array.from(points), v = 1 - u;
  for (let top = n - 1; top > 0; --top) {
    for (let r = 0; r < top; ++r) w[r] = w[r] * v + w[r + 1] * u;
  }
  return w[0];
}

// The control value that continues `curve` into the next one: its last leg, mirrored (align 1) or flipped (align -1).
function joinCurve(curve, align = 1) {
  const n = (curve ? curve.length : 0);
  return (n === 0 ? 0 : (n === 1 ? curve[0] : (curve[n - 1] - curve[n - 2]) * align));
}

// ---- blending ----------------------------------------------------------------------------------------------------

const weight = (u, curve) => (curve ? curveAt(curve, u) : u);

function tweenValue(a, b, t, ease) {
  if (a === b || !numeric(a)) return b;
  const w = weight(t, ease);
  return a * (1 - w) + b * w;
}

// Every entry `b` names, written into `out`: numbers are blended from `a` (or, where `a` has no number of that name, from
// what `out` holds), anything else switches over when t reaches 1.
function tweenProps(a, b, t, ease, out = {}) {
  for (const k of names(b)) {
    const from = (a && numeric(a[k])) ? a[k] : out[k];
    const to = numeric(b[k]) ? b[k] : out[k];
    out[k] = (numeric(from) && numeric(to)) ? tweenValue(from, to, t, ease) : (t < 1 ? from : to);
  }
  return out;
}

// tween(a, b, t, ease[, out]) for two numbers or two collections, or tween(span[, out]) with span = {a, b, t, ease}.
function tween(a, b, ...rest) {
  if (!rest.length) return tween(a.a, a.b, a.t, a.ease, b);
  return (numeric(b) ? tweenValue : tweenProps)(a, b, ...rest);
}

// What a move of the playhead does to its output: keys that were jumped over land whole, the interval the head stands
// in is blended, then the jumped-over keys' calls run.
function apply(span, out = {}) {
  if (span) {
    Object.assign(out, span.apply);
    tween(span, out);
    names(span.call).forEach((k) => span.call[k](out, span));
  }
  return out;
}

// ---- a track -----------------------------------------------------------------------------------------------------

function makeFrame(to, time, ease, call) { return (arguments.length > 1 ? { to, time, ease, call } : to); }

// first index whose stamp is >= time (strict: > time); stamps.length when there is none
function bisect(stamps, time, strict) {
  let lo = 0, hi = stamps.length;
  while (lo < hi) {
    const mid = (lo + hi) >>> 1;
    if (strict ? stamps[mid] > time : stamps[mid] >= time) hi = mid; else lo = mid + 1;
  }
  return lo;
}

const stretch = (timed) => (timed.end() || 0) - (timed.start() || 0);

function fraction(lo, hi, time) {
  const f = (time - lo) / (hi - lo) || 0;           // NaN (an infinite or empty interval) counts as its start
  return (f < 0 ? 0 : (f > 1 ? 1 : f));
}

class Timeline {
  constructor(frames, infinite = true, rewind = false, symmetric = true) {
    this.infinite = infinite;
    this.rewind = rewind;           // the interval is handed out back to front (next <-> past)
    this.symmetric = symmetric;     // rewinding keeps the later key's curve; if not, the curve of the key being approached
    this.reverse = false;           // read by play() and spliceAt(): the direction that counts as "onwards"
    this.time = 0;
    this.gap = -1;
    this.held = null;               // the cached interval
    this.setup(frames, infinite);
  }

  // -- numbering: frame number = key index + base (an open-ended track counts its "before" end as frame 0)

  get base() { return (this.infinite ? 1 : 0); }

  get size() { return this.keys.length + 2 * this.base; }

  at(number) {
    const i = number - this.base;
    return (i < 0 ? (this.infinite && i === -1 ? BEFORE : undefined)
      : (i < this.keys.length ? this.keys[i] : (this.infinite && i === this.keys.length ? AFTER : undefined)));
  }

  // The reference's array, rebuilt on request (a view: edit through the methods).
  get frames() { return (this.infinite ? [BEFORE, ...this.keys, AFTER] : this.keys.slice()); }

  get span() {
    const h = this.held;
    return (h ? { past: h.past, next: h.next, a: h.past.to, b: h.next.to, t: h.t, ease: h.ease } : undefined);
  }

  // -- keys

  // (frames of equal time keep their given order: the reference's comparator answers -1 for a tie and so leaves them to the
  // engine's sort algorithm - src/animate/timeline.js `order`; ties met by add() follow the reference's rule)
  setup(frames = [], infinite = this.infinite) {
    this.infinite = infinite;
    const sorted = frames.map((frame, i) => [frame, i]).sort((p, q) => (p[0].time - q[0].time) || (p[1] - q[1]));
    this.keys = sorted.map((p) => p[0]);
    this.stamps = this.keys.map((frame) => frame.time);
    return this.frames;
  }

  merge(frames) {
    names(frames).forEach((k) => this.add(frames[k]));
    return frames;
  }

  insertFrame(number, frame) {
    const i = Math.min(Math.max(number - this.base, 0), this.keys.length);
    this.keys.splice(i, 0, frame);
    this.stamps.splice(i, 0, frame.time);
    return this;
  }

  indexOf(frame) { return bisect(this.stamps, frame.time, true) + this.base; }

  add(...frame) {
    const record = makeFrame(...frame);
    const number = this.indexOf(record);
    this.insertFrame(number, record);
    return number;
  }

  // The key, and before it a key without values `duration` earlier - where its transition starts - unless another key
  // already stands inside that stretch.  Returns the number the key had when it went in.
  addSpan(duration, ...frame) {
    const number = this.add(...frame);
    const from = this.at(number).time - duration;
    const before = this.at(number - 1);
    if (duration && (!before || before.time < from)) this.add(null, from);
    return number;
  }

  // -- playhead

  gapAt(time) {
    if (this.size < 2) return -1;
    const i = bisect(this.stamps, time, false);
    const number = (this.infinite || i < this.keys.length) ? Math.max(i + this.base, 1) : this.size - 1;
    return number - 0.5;
  }

  spanGapAt(time, gap = this.gapAt(time), out = {}) {
    const h = this.interval(time, gap);
    return (h ? Object.assign(out, { past: h.past, next: h.next, a: h.past.to, b: h.next.to, t: h.t, ease: h.ease }) : undefined);
  }

  interval(time, gap) {
    if (!(gap >= 0)) return null;
    const early = this.at(gap - 0.5), late = this.at(gap + 0.5);
    const lo = Math.min(early.time, late.time), hi = Math.max(early.time, late.time);
    const turned = !!this.rewind;
    return { lo, hi, past: (turned ? late : early), next: (turned ? early : late),
             ease: ((turned && !this.symmetric) ? early.ease : late.ease), t: fraction(lo, hi, time) };
  }

  valid(gap = this.gap, span = this.span) { return (gap > 0 && span); }

  setTime(time) {
    this.gap = this.gapAt(time);
    this.held = this.interval(time, this.gap);
    this.time = time;
    return this;
  }

  seek(time) {
    const h = this.held;
    if (this.gap > 0 && h && h.lo < time && time <= h.hi) h.t = fraction(h.lo, h.hi, time);
    else this.setTime(time);
    return this.span;
  }

  // seek(), and what the head jumped over on its way (going onwards only) comes along as `apply` / `call`.
  play(time) {
    const from = Math.max(this.gap, 0.5);
    const span = this.seek(time);
    if (!this.valid()) return span;
    const steps = Math.abs(this.gap - from), down = (this.gap < from);
    if (steps > 0 && (down === !!this.reverse)) {
      span.apply = {};
      for (let j = 0; j < steps; ++j) {
        const frame = this.at(down ? from - 0.5 - j : from + 0.5 + j);
        Object.assign(span.apply, frame.to);
        if (frame.call && frame.call.length) span.call = (span.call || []).concat(frame.call);
      }
    }
    return span;
  }

  playFrom(time = this.time, start = 0) { return (this.seek(start), this.play(time)); }

  // -- taking keys out.  Frame numbers as everywhere; an open-ended track keeps its two ends and - like the reference,
  //    whose count stops one short - its last key.  Returns the records taken out; `adding` goes in their place unsorted.

  splice(index = 0, num = 0, ...adding) {
    let number = index, count = num;
    if (this.infinite) {
      const n = this.keys.length, asked = (index < 0 ? n + index : index);
      number = Math.min(n, Math.max(1, asked));
      count = Math.min(num - Math.max(number - asked, 0), n - number);
    } else if (number < 0) number += this.size;
    const i = Math.max(Math.min(number, this.size) - this.base, 0), gone = Math.max(count, 0);
    this.stamps.splice(i, gone, ...adding.map((frame) => frame.time));
    return this.keys.splice(i, gone, ...adding);
  }

  spliceIndex(index, ...adding) {
    const [gone] = this.splice(index, 1, ...adding);
    return gone;
  }

  // the key next to `time`: adjacent -1 the one before it, 1 the one after (as seen in the playing direction)
  spliceAt(time, adjacent = -1, ...adding) {
    const towards = (this.reverse ? -adjacent : adjacent);
    return this.spliceIndex(Math[towards > 0 ? 'ceil' : 'floor'](this.gapAt(time)), ...adding);
  }

  // the keys between `start` and `start + duration`
  spliceSpan(duration, start = 0, ...adding) {
    const [low, high] = [this.gapAt(start), this.gapAt(start + duration)].sort((p, q) => p - q);
    return this.splice(Math.ceil(low), Math.floor(high - low), ...adding);
  }

  // Gives frame `number` a curve that leaves the previous frame's curve without a kink: [first, joined, ...rest].
  easeJoin(number, align) {
    if (!(number > 0)) return null;
    const frame = this.at(number), own = (frame.ease && frame.ease.length ? frame.ease : [0, 1]);
    frame.ease = [own[0], joinCurve(this.at(number - 1).ease, align), ...Array.prototype.slice.call(own, 1)];
    return frame.ease;
  }

  // The frame as add() would take it, its `to` cut down to the entries that are not found - same name, same value - on
  // the record standing before its place and on the one standing after the next (the comparison the reference makes).
  minFrame(...frame) {
    const full = makeFrame(...frame), number = this.indexOf(full);
    const unlike = (record) => {
      if (!(record && record.to) || record === full.to) return null;
      if (!names(record).length || !names(full.to).length) return full.to;
      let kept = null;
      for (const k of names(full.to)) if (full.to[k] !== record[k]) (kept || (kept = {}))[k] = full.to[k];
      return kept;
    };
    const early = unlike(this.at(number - 1)), late = unlike(this.at(number + 1));
    return Object.assign({}, full, { to: ((names(early).length || names(late).length) ? Object.assign({}, early, late) : early) });
  }

  // -- extent

  start() { return (this.size ? this.at(0).time : null); }

  end() { return (this.size ? this.at(this.size - 1).time : null); }

  duration() { return stretch(this); }
}

// The chainable ways of adding a key: [takes a duration first (the key gets a start key, addSpan), the alignment of the
// joined curve - null: no join, undefined: the caller passes it].  With a duration the joined curve goes to the frame that
// now carries the number the key went in with: the transition's start key when addSpan() made one (the key itself keeps
// its curve as given) - as the reference's timelines come out.
const ADDING = {
  to: [false, null], easeTo: [false, undefined], smoothTo: [false, 1], flipTo: [false, -1],
  over: [true, null], easeOver: [true, undefined], smoothOver: [true, 1], flipOver: [true, -1]
};
Object.keys(ADDING).forEach((name) => {
  const [spanned, fixed] = ADDING[name];
  Timeline.prototype[name] = function (...args) {
    const duration = (spanned ? args.shift() : 0);
    const align = (fixed === undefined ? args.shift() : fixed);
    const number = (spanned ? this.addSpan(duration, ...args) : this.add(...args));
    if (align !== null) this.easeJoin(number, align);
    return this;
  };
});

// ---- tracks side by side -----------------------------------------------------------------------------------------

class Player {
  // `tracks`: any keyed collection (object or array) of key lists or Timelines - key lists are replaced, in place, by
  // Timelines; `outputs`: the collection, keyed alike, whose members the tracks write into.
  constructor(tracks, outputs = {}) {
    this.tracks = tracks;
    this.outputs = outputs;
    this.add(tracks);
  }

  add(tracks) {
    for (const k of names(tracks)) this.tracks[k] = (tracks[k] instanceof Timeline ? tracks[k] : new Timeline(tracks[k]));
    return this;
  }

  // Takes over the tracks of other players under their keys.  (The reference's `import` feeds each timeline to add() as
  // if it were a collection of tracks and cannot work; nothing calls it there.  This is what its name promises.)
  import(players) {
    for (const p of names(players)) this.add(players[p].tracks);
    return this;
  }

  each(f) {
    names(this.tracks).forEach((k, i, keys) => f(this.tracks[k], k, this.tracks, i, keys));
    return this;
  }

  // f(track, key, tracks, i, keys, out[key]) returns what to apply to out[key] (a span, {apply}, {call} ... or nothing).
  apply(f, out = this.outputs) {
    return this.each((track, key, ...more) => {
      if (!out[key]) out[key] = {};
      apply(f(track, key, ...more, out[key]), out[key]);
    });
  }

  frames(out = []) {
    this.each((track, key) => { out[key] = track.frames; });
    return out;
  }

  // Both ends are the smallest of zero and the tracks' own (the reference folds with a minimum from a null start, for
  // the end as well - a player of open-ended tracks ends at 0); null without tracks.
  start() { return this.extent('start'); }

  end() { return this.extent('end'); }

  extent(which) {
    const keys = names(this.tracks);
    return (keys.length ? Math.min(0, ...keys.map((k) => this.tracks[k][which]())) : null);
  }

  duration() { return stretch(this); }
}

// seek(time[, out]), play(time[, out]), playFrom(time, start[, out]): the same move on every track, each result applied
// to the track's output (in `out` if given, else in `outputs`).
[['seek', 1], ['play', 1], ['playFrom', 2]].forEach(([move, taken]) => {
  Player.prototype[move] = function (...args) {
    const out = args[taken];
    return this.apply((track) => track[move](...args.slice(0, taken)), out);
  };
});

module.exports = { Timeline, Player, apply, tween, tweenValue, tweenProps, curveAt, joinCurve, makeFrame, default: Player };
