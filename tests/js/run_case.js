'use strict';
// Drives the Node host (tendrils_amd/js: the reference's Tendrils/Particles/OpticalFlow/spawner
// API over the N-API shim) on a case described by a JSON file; raw little-endian binaries in/out.
// Usage: node run_case.js <case.json>
const fs = require('fs');
const path = require('path');

const root = path.join(__dirname, '..', '..');
const T = require(path.join(root, 'tendrils_amd', 'js'));
const { OpticalFlow } = require(path.join(root, 'tendrils_amd', 'js', 'optical-flow'));
const { spawnBall } = require(path.join(root, 'tendrils_amd', 'js', 'spawn', 'ball'));
const { PixelSpawner, flowSampleFrag, dataSampleFrag, bestSampleFrag, pixelsFrag } = require(path.join(root, 'tendrils_amd', 'js', 'spawn', 'pixels'));

const { GeometrySpawner } = require(path.join(root, 'tendrils_amd', 'js', 'spawn', 'geometry'));

const spec = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const dir = path.dirname(process.argv[2]);
const f32 = (name) => { const b = fs.readFileSync(path.join(dir, name)); return new Float32Array(b.buffer, b.byteOffset, b.length / 4); };
const u8 = (name) => { const b = fs.readFileSync(path.join(dir, name)); return new Uint8Array(b.buffer, b.byteOffset, b.length); };
const save = (name, arr) => fs.writeFileSync(path.join(dir, name), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));

const gl = { drawingBufferWidth: spec.viewRes[0], drawingBufferHeight: spec.viewRes[1] };
const t = new T.Tendrils(gl, {});
t.resize();
t.setup(spec.N);
if (spec.flowShape) t.flow.shape = spec.flowShape;
if (spec.viewSize) { t.viewSize[0] = spec.viewSize[0]; t.viewSize[1] = spec.viewSize[1]; }
Object.assign(t.state, spec.state || {});
if (spec.inputs.state) t.particles.uploadTexels(f32(spec.inputs.state));
if (spec.inputs.flow) t.flow.setPixels(f32(spec.inputs.flow));
if (spec.inputs.targets) t.targets.setPixels(f32(spec.inputs.targets));

if (spec.kind === 'logic') {
  t.timer.time = spec.time0;
  if (spec.stepN) {
    t.stepN(spec.steps);
    save(`out_${spec.steps - 1}.bin`, t.particles.read(0));
  } else for (let k = 0; k < spec.steps; ++k) {
    if (k && spec.follow) t.particles.uploadTexels(f32(spec.follow[k - 1]));
    if (spec.times) t.timer.time = spec.times[k] - spec.dts[k];
    t.timer.tick();
    t.step();
    save(`out_${k}.bin`, t.particles.read(0));
  }
  fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ time: t.timer.time, dt: t.timer.dt,
    order: t.particles.buffers.map((b) => b.id) }));
} else if (spec.kind === 'frames') {            // the reference's frame loop: step() then draw() (flow deposit)
  t.timer.time = spec.time0;
  const fragments = [];
  for (let k = 0; k < spec.frames; ++k) {
    t.timer.tick();
    t.step().draw();
    fragments.push(t.fragments);
  }
  save('state.out.bin', t.particles.read(0));
  save('flow.out.bin', t.flow.read());
  save('lines.out.bin', t.exportLines());
  fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ time: t.timer.time, fragments }));
} else if (spec.kind === 'optical_flow') {
  const of = new OpticalFlow(t);
  of.resize(spec.frame);
  of.setPixels(u8(spec.inputs.last));
  of.step();
  of.setPixels(u8(spec.inputs.view));
  of.update(spec.uniforms);
  of.render();
  save('out_0.bin', t.flow.read());
} else if (spec.kind === 'spawn_ball') {
  spawnBall(null, { uniforms: spec.uniforms }).spawn(t);
  save('out_0.bin', t.particles.read(0));
  fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ time: t.timer.time }));
} else if (spec.kind === 'spawn_sample') {
  const flowSrc = spec.apply === 0;
  const sp = new PixelSpawner(null, {
    shader: flowSrc ? flowSampleFrag() : dataSampleFrag(),
    buffer: flowSrc ? t.flow : t.particles.buffers[0],
    spawnSize: spec.uniforms.spawnSize, speed: spec.uniforms.speed, bias: spec.uniforms.bias
  });
  t.timer.time = spec.uniforms.time - t.timer.step;
  sp.spawn(t);
  save('out_0.bin', t.particles.read(0));
  fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ time: t.timer.time, jitter: sp.jitter }));
} else if (spec.kind === 'spawn_image') {         // image spawners (src/demo.main.js:455-515)
  const sp = new PixelSpawner(null, {
    shader: spec.direct ? pixelsFrag() : bestSampleFrag(),
    spawnSize: spec.uniforms.spawnSize, speed: spec.uniforms.speed, bias: spec.uniforms.bias,
    jitterRad: spec.jitterRad
  });
  sp.spawnMatrix = spec.uniforms.spawnMatrix;
  sp.setPixels(f32(spec.inputs.image), spec.imageShape);
  t.timer.time = spec.uniforms.time - t.timer.step;
  sp.spawn(t);
  save('out_0.bin', t.particles.read(0));
  fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ time: t.timer.time, jitter: sp.jitter }));
} else if (spec.kind === 'geometry') {            // GeometrySpawner (src/spawn/geometry/index.js)
  const sp = new GeometrySpawner(null, { speed: spec.speed, bias: spec.bias, positions: Array(spec.positions.length).fill(0) });
  sp.shuffle();                                   // exercises Math.random-driven shuffle; then pin the triangles
  const moved = sp.positions.some((v) => v !== 0);
  sp.positions = spec.positions;
  t.timer.time = spec.time0;
  sp.spawn(t);
  save('out_0.bin', t.particles.read(0));
  fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ time: t.timer.time, jitter: sp.jitter, moved }));
} else if (spec.kind === 'scene') {               // presets keyframed by js/scenes.js while step() + draw() run
  const { Scene } = require(path.join(root, 'tendrils_amd', 'js', 'scenes'));
  const scene = new Scene(t).preset(spec.table[spec.first]);
  for (const k of spec.script) scene.keyframe(spec.table[k.preset], k.time, k.duration, k.ease);
  t.timer.time = spec.time0;
  const states = [], times = [];
  scene.run(spec.frames, (k, tn) => {
    times.push(tn.timer.time);
    states.push(JSON.parse(JSON.stringify(tn.state)));
    save(`state_${k}.bin`, tn.particles.read(0));
    if (spec.grab.includes(k)) { save(`flow_${k}.bin`, tn.flow.read()); save(`view_${k}.bin`, tn.readView()); }
  });
  fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ times, states }));
} else {
  throw new Error('unknown case kind ' + spec.kind);
}
t.dispose();
