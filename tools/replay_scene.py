#!/usr/bin/env python3
"""Batch-render a scene: presets from a JSON table eased into each other while the particle path runs; every Nth view
image is written as a binary PPM.   python tools/replay_scene.py presets.json "Flow" "Wings" --frames 240 --out /tmp/scene"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tendrils_amd as ta  # noqa: E402
from tendrils_amd.scenes import Scene  # noqa: E402
from tendrils_amd.spawn.ball import spawnBall  # noqa: E402
from tendrils_amd.tendrils import View  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("table")
ap.add_argument("presets", nargs="+")
ap.add_argument("--frames", type=int, default=240)
ap.add_argument("--every", type=int, default=30)
ap.add_argument("--root", type=int, default=1024)
ap.add_argument("--view", default="960x540")
ap.add_argument("--out", default="scene")
args = ap.parse_args()

table = json.load(open(args.table))
w, h = (int(v) for v in args.view.split("x"))
t = ta.Tendrils(View(w, h))
t.resize()
t.setup(args.root)
scene = Scene(t).preset(table[args.presets[0]])
span = args.frames * t.timer.step / max(len(args.presets) - 1, 1)
for k, name in enumerate(args.presets[1:], 1):
    scene.keyframe(table[name], time=k * span, duration=0.6 * span, ease=[0, 0.95, 1])
first = table[args.presets[0]].get("spawn", {})


def save(k, tn):
    if k % args.every == args.every - 1:
        img = tn.read_view()[::-1, :, :3]
        with open("%s_%04d.ppm" % (args.out, k + 1), "wb") as f:
            f.write(b"P6 %d %d 255\n" % (w, h) + np.ascontiguousarray(img).tobytes())
        print("frame %d: %d fragments, view mean %.2f" % (k + 1, tn.view_fragments, float(img.mean())))


scene.run(args.frames, each=save, spawner=spawnBall(None, dict(uniforms=dict(radius=first.get("radius", 0.3), speed=first.get("speed", 0.005)))))
t.dispose()
