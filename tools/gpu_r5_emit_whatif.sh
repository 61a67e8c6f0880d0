#!/bin/bash
# what-if variants of bins_fused_kernel (TH_EMIT_EXP): the emit's kernel time over the first 40 frames, both passes per draw
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1 2 3 0}; do
  export TH_EMIT_EXP=$v
  rm -rf /tmp/prof_w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o run -- python3 $R/tools/deposit_bench.py ${1:-40} --both > /tmp/prof_w.log 2>&1
  echo "== TH_EMIT_EXP=$v  $(grep -o '"draw_both_ms": [0-9.]*' /tmp/prof_w.log)  $(grep -o '"fragments_per_frame": [0-9.]*' /tmp/prof_w.log)"
  f=$(find /tmp/prof_w -name '*kernel_trace.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, statistics
d = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "bins_fused" in n or "bins_listed" in n or "logic_kernel" in n:
        d.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    print("   %-70s calls %5d  median %9.1f us  mean %9.1f  min %9.1f  max %9.1f" % (n[:70], len(v), statistics.median(v), statistics.mean(v), min(v), max(v)))
PY
done
