'use strict';
// src/spawn/ball/index.js:5-16 - hash-random disc spawner (program: src/spawn/ball/index.frag).
const init = require('./init');
const { Program } = require('../particles');

const defaults = () => ({ shader: new Program('spawn-ball'), uniforms: { radius: 1, speed: 0 } });

const spawnBall = (gl, options) => init.spawner(gl, Object.assign(defaults(), options));

// src/spawn/ball/cpu.js:1-19
const cpu = (data, radius = 1.0, speed = 0.01) => {
  let angle = Math.random() * Math.PI * 2;
  let scaled = Math.random() * radius;
  data[0] = Math.cos(angle) * scaled;
  data[1] = Math.sin(angle) * scaled;
  angle = Math.random() * Math.PI * 2;
  scaled = Math.random() * speed;
  data[2] = Math.cos(angle) * scaled;
  data[3] = Math.sin(angle) * scaled;
  return data;
};

module.exports = { defaults, spawnBall, cpu, default: spawnBall };
