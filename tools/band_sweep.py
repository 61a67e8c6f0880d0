#!/usr/bin/env python3
"""What an 8-GPU strong-scaling run of the metric's 16 M particles is made of, measured on ONE GPU.

For the per-GPU share each point of the 1 / 2 / 4 / 8 curve implies - 4096, 2048, 1024, 512 rows of the 4096-wide texture -
run the bench's own timed region (`bench.py --config c3_strong --pretend-world P --force-dist`: the driver's K and W, the
library's RCCL communicator up at world 1, statistics fold + counter all-reduce after every launch) and print, per point:
particle-steps/s of the band, the wall per step, the integrator launch, and what a launch group costs beside its kernel
(host enqueue, statistics fold, all-reduce enqueue-to-done).  The ratio of a band's rate x P to the whole texture's rate is
the per-GPU efficiency the measured curve cannot beat (xGMI latency of the all-reduce comes on top).

    python3 tools/band_sweep.py [--steps 20] [--warmup 5] [--out profiles/r6_a_band_sweep.txt] [extra bench.py arguments]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--worlds", default="1,2,4,8")
    args, extra = ap.parse_known_args()
    rows, base = [], None
    for p in (int(v) for v in args.worlds.split(",")):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c3_strong", "--pretend-world", str(p), "--force-dist",
               "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu", "--no-traffic", "--no-frame-loop"] + extra
        env = dict(os.environ, MASTER_PORT=str(29600 + p))
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        line = None
        for text in r.stdout.splitlines():
            if text.startswith("{") and '"metric"' in text:
                line = json.loads(text)
        if line is None:
            rows.append("P=%d: no line (rc %d)\n%s" % (p, r.returncode, r.stderr[-1500:]))
            continue
        fc = line.get("fixed_costs") or {}
        rate = line["value"]
        if base is None:
            base = rate
        rl = line["roofline"]
        rows.append("P=%d rows=%4d particles=%9d | %7.2f G particle-steps/s (x P = %7.2f G, %.3f of the whole texture's rate) | wall %.4f ms/step | "
                    "launch %.4f ms (%d steps: %.4f ms/step) | group wall %.4f ms, beside the kernel %.4f ms: host enqueue %.4f, fold %.4f, "
                    "all-reduce %s, empty event pair %.4f, empty bracket %.4f | reps %s"
                    % (p, 4096 // p, line["config"]["particles_per_gpu"], rate / 1e9, rate * p / 1e9, rate * p / base, line["ms_per_step"],
                       rl["avg_launch_ms"], round(rl["steps_per_launch"]), rl["ms_per_step"], fc.get("group_wall_ms", float("nan")),
                       fc.get("beside_the_kernel_ms", float("nan")), fc.get("host_enqueue_ms", float("nan")), fc.get("stats_fold_ms", float("nan")),
                       ("%.4f" % fc["allreduce_ms"]) if fc.get("allreduce_ms") is not None else "-", fc.get("empty_event_pair_ms", float("nan")), fc.get("bracket_ms", float("nan")),
                       ["%.4f" % v for v in line["repetitions"]["ms_per_step"]]))
        if p == 1:
            rows.append("      rccl: %s" % json.dumps(line.get("rccl")))
    text = "\n".join(rows) + "\n"
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "a") as f:
            f.write("# tools/band_sweep.py --steps %d --warmup %d %s\n" % (args.steps, args.warmup, " ".join(extra)))
            f.write(text)


if __name__ == "__main__":
    main()
