#!/bin/bash
# round 5: the blocked sweeps (TH_STEP_VARIANT 5 / 6) against the interleaved one (0) over grids, with the memory-side request counters
# (variants 5 / 6 / 7 live in the trees of commits 0dc8af8 and its successor; not kept: profiles/r5_b_single_step_variants.txt)
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_step2; mkdir -p $O
for v in 5 6; do
  echo "=== parity TH_STEP_VARIANT=$v"
  TH_STEP_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_logic_parity.py tests/test_gpu_bucketed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
done 2>&1 | tee $O/parity.txt
for round in 1 2; do
  for cfg in "0 20" "5 7" "5 14" "5 20" "5 56" "6 8" "6 16" "6 56"; do
    set -- $cfg
    export TH_STEP_VARIANT=$1 TH_STEP_GRID=$2
    echo "=== variant $1 grid $2: default"; timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -1
    echo "=== variant $1 grid $2: flow-only"; timeout 120 python tools/step_probe.py --flow-only 2>&1 | grep "single step" | tail -1
  done
done 2>&1 | tee $O/probe.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "5 14" "6 16"; do
  set -- $cfg
  export TH_STEP_VARIANT=$1 TH_STEP_GRID=$2
  OUT=$GRAFT_REPO_ROOT/$O/pmc_v$1
  mkdir -p $OUT
  run () { name=$1; shift; PROBE_STEPS=24 timeout 100 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/step_probe.py > $OUT/$name.log 2>&1 || echo "$name failed"; }
  run ea1 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
  run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
  run tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
  run ta  TA_TA_BUSY_sum TA_BUSY_avr
done
cd $GRAFT_REPO_ROOT
python3 - $O <<'PY' | tee $O/pmc_summary.txt
import csv,glob,sys,collections
out=sys.argv[1]
for v in sorted(glob.glob(out+'/pmc_v*')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    dur={}
    for f in glob.glob(v+'/*/*/*counter_collection.csv'):
        run=f.split('/')[-3]
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'logic_kernel' not in k and 'logic_ring' not in k: continue
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            acc[k]['_ns_'+run].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
    print('##',v)
    for k,cs in acc.items():
        print('#',k)
        for c,vv in sorted(cs.items()):
            print('   %-40s %16.1f (n=%d)'%(c,sum(vv)/len(vv),len(vv)))
PY
