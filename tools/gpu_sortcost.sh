#!/bin/bash
# cost of the fused path's re-sort (tile_hist + tile_scan + tile_scatter every 256 steps)
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_logic_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/trace_sortcost
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1024 --warmup 128 --no-cpu --no-traffic --no-frame-loop > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $OUT.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.1f G ms/step %.5f' % (d['value']/1e9, d['ms_per_step']))"
python3 - $OUT <<'PY'
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'tile_' in r['Name'] or 'fused' in r['Name']: print("%-80s calls %5s avg %10.1f us  %6s%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
