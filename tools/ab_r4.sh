#!/bin/bash
# Same-box A/B of the round-4 tree (tools/bin/ab/r4: `git archive 57e4d85` + make; git-ignored, travels with gpurun) and the working
# tree: the driver command's headline and the single-step kernel of each, alternating.   usage: bash tools/ab_r4.sh [rounds]
rounds=${1:-3}
root=$(pwd)
for k in $(seq 1 "$rounds"); do
  for tree in tools/bin/ab/r4 .; do
    extra="--no-cpu --no-traffic --no-frame-loop --no-c4"
    [ "$tree" = "." ] && extra="$extra --no-c5"
    out=$(cd "$root/$tree" && python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra 2>/dev/null | tail -1)
    python3 - "$tree" "$k" "$out" <<'PY'
import json, sys
tree, k, line = sys.argv[1:4]
d = json.loads(line)
r = d.get("roofline", {})
print("%-5s run %s  value %.4g G  ms_per_step %.5f  fused launch %.4f ms  single step %.4f ms  flow-only fused %.4f / single %.4f ms" % (
    "r4" if "r4" in tree else "head", k, d["value"] / 1e9, d["ms_per_step"], r.get("avg_launch_ms", float("nan")),
    r.get("single_step_kernel", {}).get("ms_per_step", float("nan")), r.get("flow_only", {}).get("avg_launch_ms", float("nan")),
    r.get("flow_only", {}).get("single_step_kernel_ms", float("nan"))))
PY
  done
done
