#!/bin/bash
# kernel stats of the crowded frame loop (tools/deposit_bench.py N --both under rocprofv3 --kernel-trace --stats): the draw's kernels
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o run -- python3 $R/tools/deposit_bench.py ${1:-300} --both > /tmp/prof_c.log 2>&1
grep draw_both /tmp/prof_c.log | cut -c1-260
f=$(find /tmp/prof_c -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-80s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"].replace("(anonymous namespace)::","")[:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
