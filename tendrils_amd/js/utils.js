'use strict';
// Host helpers mirrored from the reference: src/utils/index.js:1-7 (`step`: rotate a ring,
// pop -> unshift), src/utils/aspect.js:4-11 (aspect / containAspect / coverAspect).

function step(array) {
  const next = Array.prototype.pop.call(array);
  Array.prototype.unshift.call(array, next);
  return next;
}

function aspect(out, size, scale) {
  out[0] = scale / size[0];
  out[1] = scale / size[1];
  return out;
}

const containAspect = (out, size) => aspect(out, size, Math.min(size[0], size[1]));
const coverAspect = (out, size) => aspect(out, size, Math.max(size[0], size[1]));

const inert = -1000000;   // src/const/inert.js:2

module.exports = { step, aspect, containAspect, coverAspect, inert };
