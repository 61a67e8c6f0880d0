#!/bin/bash
# kernel medians (rocprofv3 --kernel-trace) of the frame loop with tools/bin/head and tools/bin/alt libraries, interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for v in head alt; do
  export TH_LIB=$R/tools/bin/$v/libtendrils_hip.so
  for frames in ${FRAMES:-40 300}; do
  rm -rf /tmp/prof_w
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_w -o run -- python3 $R/tools/deposit_bench.py $frames --both > /tmp/prof_w.log 2>&1
  echo "== $v frames $frames  $(grep -o '"draw_both_ms": [0-9.]*' /tmp/prof_w.log)"
  f=$(find /tmp/prof_w -name '*kernel_trace.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, os, sys, statistics
d = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void th::", "").replace("th::", "")
    d.setdefault(n.split("(")[0][:44], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    if any(k in n for k in (os.environ.get("KERNELS", "bins_fused bins_listed").split())):
        print("   %-46s calls %5d  median %8.1f us  mean %8.1f" % (n, len(v), statistics.median(v), statistics.mean(v)))
PY
  done
done
done
