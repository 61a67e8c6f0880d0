'use strict';
// Simulation clock with the public surface of the reference's Timer (src/timer.js:1-80): fields time, since,
// offset, rate, step, dt, paused, end, loop; chainable now / tick / seek / scrub / reset.
//
// Progress comes from one of two sources: a fixed step (step >= 0: every tick moves the clock by step*rate, the
// mode the headless host uses) or the wall clock ((now - offset)*rate).  While paused the would-be progress is
// absorbed into `offset`, so un-pausing continues where the clock stopped.  With an `end` the clock either wraps
// (loop) or stops at the end and pauses itself.

const INITIAL = Object.freeze({ time: 0, since: 0, offset: 0, rate: 1, step: -1, dt: 0, paused: false, end: -1, loop: false });

// where the clock would go on this tick, and by how much
function proposal(timer, wallNow) {
  if (timer.step >= 0) {
    const delta = timer.step * timer.rate;
    return { next: timer.time + delta, delta };
  }
  const next = timer.now(wallNow);
  return { next, delta: next - timer.time };
}

// the end / loop rules applied to a proposed clock value; returns [time, reachedTheEnd]
function bounded(timer, next) {
  if (timer.end < 0) return [next, false];
  if (timer.loop) return [next % timer.end, false];
  const stop = (timer.rate > 0 ? Math.min(next, timer.end) : Math.max(next, timer.end));
  return [stop, stop !== next];
}

class Timer {
  constructor(now, since) {
    Object.assign(this, INITIAL);
    this.reset(now, since);
  }

  now(wallNow = Date.now()) { return (wallNow - this.offset) * this.rate; }

  tick(wallNow) {
    const { next, delta } = proposal(this, wallNow);
    if (this.paused) {
      this.offset += delta;
      this.dt = 0;
      return this;
    }
    const [time, finished] = bounded(this, next);
    this.time = time;
    if (finished) this.paused = true;
    this.dt = delta;
    return this;
  }

  seek(to) { this.offset = -to; return this; }

  scrub(by) { this.offset -= by; return this; }

  reset(wallNow = Date.now(), since = wallNow) {
    this.offset = since;
    this.since = since;
    this.time = this.now(wallNow);
    return this;
  }
}

module.exports = { Timer, default: Timer };
