#!/bin/bash
# kernel-trace stats of the single-step probe: gpu_trace_step.sh TAG [probe args]
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE_STEPS=${PROBE_STEPS:-24} timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/step_probe.py "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.5:
        print("%-100s calls %5s avg %10.1f us  %6s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
