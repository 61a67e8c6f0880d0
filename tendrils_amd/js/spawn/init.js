'use strict';
// src/spawn/init/index.js:6-28 - generic spawner factory; default program = all-inert fill.
const { Program } = require('../particles');

const defaults = () => ({ shader: new Program('spawn-init'), uniforms: null });

const spawner = (gl, options) => {
  const params = Object.assign(defaults(), options);
  return {
    gl,
    uniforms: params.uniforms,
    shader: params.shader,
    spawn(tendrils, ...rest) { tendrils.spawnShader(this.shader, this.uniforms, ...rest); }
  };
};

module.exports = { defaults, spawner, default: spawner };
