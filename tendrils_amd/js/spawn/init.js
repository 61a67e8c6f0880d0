'use strict';
// Generic spawner factory with the surface of src/spawn/init/index.js:6-28: spawner(gl, {shader, uniforms}) ->
// {gl, uniforms, shader, spawn(tendrils, ...rest)}.  Without options the program is the all-inert fill
// (src/spawn/init/index.frag:5-10).
const { Program } = require('../particles');

function Spawner(gl, program, uniforms) {
  this.gl = gl;
  this.shader = program;
  this.uniforms = uniforms;
}

// one respawn pass of this spawner's program through Tendrils.spawnShader (which ticks the timer)
Spawner.prototype.spawn = function spawn(tendrils, ...rest) {
  tendrils.spawnShader(this.shader, this.uniforms, ...rest);
};

const defaults = () => ({ shader: new Program('spawn-init'), uniforms: null });

function spawner(gl, options = {}) {
  const chosen = { ...defaults(), ...options };
  return new Spawner(gl, chosen.shader, chosen.uniforms);
}

module.exports = { defaults, spawner, default: spawner };
