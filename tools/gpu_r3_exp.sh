#!/bin/bash
# round 3: the binned draw() pipeline: tests + kernel traces per environment variant.
# usage: gpu_r3_exp.sh "VAR=a VAR=b ..." [frames] [notest]     (each word = one variant's environment, "-" = none)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
VARIANTS=${1:--}
FRAMES=${2:-40}
if [ "${3:-}" != "notest" ]; then
  timeout 1500 python -m pytest tests/test_gpu_binned_draw.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/binned_tests.log
fi
cd /tmp
for v in $VARIANTS; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  D=$OUT/trace_$tag
  rm -rf $D
  e=""; [ "$v" != "-" ] && e="$v"
  env $e TH_PIPE=auto timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/tools/deposit_bench.py $FRAMES > $D.log 2>&1
  echo "== $v"; grep '^{' $D.log | cut -c1-330
  python3 - $D <<'PY'
import csv,glob,sys
fs=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')
if fs:
  for r in csv.DictReader(open(fs[0])):
    if float(r["Percentage"]) > 0.9 and ("bins_" in r["Name"] or "crowd_" in r["Name"]):
        print("%-75s calls %5s avg %9.1f us  %6s%%" % (r["Name"].replace("th::(anonymous namespace)::","")[:75], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
