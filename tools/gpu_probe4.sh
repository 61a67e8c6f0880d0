#!/bin/bash
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
PROBE_STEPS=32 TH_RESORT_STEPS=8 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/trace_sorted -- python3 $GRAFT_REPO_ROOT/tools/step_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r2/trace_sorted.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2/trace_sorted/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print("%-110s calls %5s avg %10.1f us  %6s%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
