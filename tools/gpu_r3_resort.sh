#!/bin/bash
# frame loop (step + binned draw) against the re-sort period of the slot order
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for period in 2 4 8 16 32 64; do
  echo "resort period $period" | tee -a $OUT/resort_sweep.log
  TH_RESORT_STEPS=$period TH_PIPE=bins timeout 300 python tools/deposit_bench.py 64 2>&1 | tail -1 | tee -a $OUT/resort_sweep.log
done
cd /tmp
D=$OUT/trace_bins_p4
TH_RESORT_STEPS=4 TH_PIPE=bins timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/tools/deposit_bench.py 32 > $D.log 2>&1
python3 - $D <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3:
        print("%-95s calls %5s avg %9.1f us  %6s%%" % (r["Name"].replace("th::(anonymous namespace)::","")[:95], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
