#!/bin/bash
# kernel stats of the frame loop (rocprofv3 --kernel-trace --stats): tools/frame_wall_probe.py, 180 frames at C3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o run -- python3 $R/tools/frame_wall_probe.py ${1:-60} > /tmp/prof_f.log 2>&1
f=$(find /tmp/prof_f -name '*kernel_stats.csv' | head -1)
grep wall_ms /tmp/prof_f.log
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-110s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
