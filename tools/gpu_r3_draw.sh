#!/bin/bash
# round 3: the binned draw() pipeline - tests, frame-loop timings of both pipelines, kernel traces
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
if [ "${1:-}" != "notest" ]; then
  timeout 1700 python -m pytest tests/test_gpu_binned_draw.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/binned_tests.log
fi
for pipe in stream bins; do
  TH_PIPE=$pipe timeout 300 python tools/deposit_bench.py 60 2>&1 | tail -1 | tee -a $OUT/deposit_bench.log
  TH_PIPE=$pipe timeout 300 python tools/deposit_bench.py 30 --both 2>&1 | tail -1 | tee -a $OUT/deposit_bench.log
done
TH_PIPE=bins timeout 300 python tools/deposit_bench.py 60 --in-view 2>&1 | tail -1 | tee -a $OUT/deposit_bench.log
cd /tmp
for pipe in bins; do
  D=$OUT/trace_$pipe
  TH_PIPE=$pipe timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/tools/deposit_bench.py 30 > $D.log 2>&1
  python3 - $D <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3:
        print("%-95s calls %5s avg %9.1f us  %6s%%" % (r["Name"].replace("th::(anonymous namespace)::","")[:95], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
