#!/bin/bash
# Same-box A/B of the headline (driver command) between the round-2 tree (tools/bin/ab/r2, built from 1583d4f~1 by
# `git archive` + make; git-ignored, travels with gpurun) and the working tree: alternating runs, `value` of each.
# Set up (build container): mkdir -p tools/bin/ab/r2 && git archive 1583d4f~1 -- tendrils_amd oracle bench.py include __graft_entry__.py tests/helpers.py | tar -x -C tools/bin/ab/r2
#                           && make -C tools/bin/ab/r2/tendrils_amd/csrc all && make -C tools/bin/ab/r2/oracle all
# usage (on the GPU box): bash tools/ab_headline.sh [rounds] > gpurun_out/ab_headline.txt
rounds=${1:-3}
root=$(pwd)
for k in $(seq 1 "$rounds"); do
  for tree in tools/bin/ab/r2 .; do
    extra="--no-cpu --no-traffic --no-frame-loop"
    [ "$tree" = "." ] && extra="$extra --no-c4"
    out=$(cd "$root/$tree" && python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra 2>/dev/null | tail -1)
    python3 - "$tree" "$k" "$out" <<'PY'
import json, sys
tree, k, line = sys.argv[1:4]
d = json.loads(line)
r = d.get("roofline", {})
reps = d.get("repetitions", {}).get("ms_per_step")
print("%-5s run %s  value %.4g G  ms_per_step %.5f  fused launch %.4f ms  repetitions %s" % (
    "r2" if "r2" in tree else "head", k, d["value"] / 1e9, d["ms_per_step"], r.get("avg_launch_ms", float("nan")),
    [round(x, 5) for x in reps] if reps else None))
PY
  done
done
