#!/bin/bash
# kernel stats of the sharded draw() at world 1 (rocprofv3 --kernel-trace --stats): tools/sharded_draw_probe.py 30 sharded
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o run -- python3 $R/tools/sharded_draw_probe.py 30 sharded > /tmp/prof_s.log 2>&1
grep draw_both /tmp/prof_s.log
f=$(find /tmp/prof_s -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-100s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
