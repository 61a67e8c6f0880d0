#!/bin/bash
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-dep}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 30 "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $OUT.log
python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3:
        print("%-95s calls %5s avg %9.1f us  %6s%%" % (r["Name"].replace("th::(anonymous namespace)::","")[:95], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
