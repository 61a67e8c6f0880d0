#!/bin/bash
# round 3: the frame loop in its steady state (hundreds of frames: the wake has crowded the particles): kernel stats per pipeline
# usage: gpu_r3_steady.sh [frames] ["pipelines"]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
FRAMES=${1:-600}
PIPES=${2:-"bins stream auto"}
cd /tmp; export TMPDIR=/tmp
for pipe in $PIPES; do
  D=$OUT/steady_$pipe
  rm -rf $D
  TH_PIPE=$pipe timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/tools/deposit_bench.py $FRAMES --both > $D.log 2>&1
  echo "== $pipe"; grep '^{' $D.log | cut -c1-330
  python3 - $D <<'PY'
import csv,glob,sys
fs=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')
if fs:
  for r in csv.DictReader(open(fs[0])):
    if float(r["Percentage"]) > 0.8:
        print("%-80s calls %5s avg %9.1f us  max %9.1f  %6s%%" % (r["Name"].replace("th::(anonymous namespace)::","")[:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3, r["Percentage"]))
PY
done
