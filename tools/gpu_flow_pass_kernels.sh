#!/bin/bash
# the flow pass alone (tools/deposit_bench.py N: flow pass and view pass one after the other): kernel medians under rocprofv3 --kernel-trace
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_w
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_w -o run -- python3 $R/tools/deposit_bench.py ${1:-40} > /tmp/prof_w.log 2>&1
grep -o '"draw_ms": [0-9.]*\|"view_ms": [0-9.]*\|"step_ms": [0-9.]*' /tmp/prof_w.log | tr '\n' ' '; echo
f=$(find /tmp/prof_w -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, statistics
d = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void th::", "").replace("th::", "")
    d.setdefault(n.split("(")[0][:44], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("   %-46s calls %5d  median %8.1f us  mean %8.1f  total %7.1f ms" % (n, len(v), statistics.median(v), statistics.mean(v), sum(v) / 1e3))
PY
