#!/bin/bash
# the frame loop at re-sort periods TH_RESORT_STEPS = $PERIODS (default 64 32 16 8): tools/deposit_bench.py N --both under
# rocprofv3 --kernel-trace - step / draw per frame (HIP events) and the kernels' medians
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in ${PERIODS:-64 32 16 8}; do
  export TH_RESORT_STEPS=$v
  rm -rf /tmp/prof_w
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_w -o run -- python3 $R/tools/deposit_bench.py ${1:-300} --both > /tmp/prof_w.log 2>&1
  echo "== TH_RESORT_STEPS=$v  $(grep -o '"step_ms": [0-9.]*' /tmp/prof_w.log)  $(grep -o '"draw_both_ms": [0-9.]*' /tmp/prof_w.log)  $(grep -o '"fragments_per_frame": [0-9.]*' /tmp/prof_w.log)"
  f=$(find /tmp/prof_w -name '*kernel_trace.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, statistics
d = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void th::", "").replace("th::", "")
    d.setdefault(n.split("(")[0][:44], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print("   %-46s calls %5d  median %8.1f us  mean %8.1f  total %8.1f ms" % (n, len(v), statistics.median(v), statistics.mean(v), sum(v) / 1e3))
PY
done
