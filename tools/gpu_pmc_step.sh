#!/bin/bash
# PMC passes over the single-step probe. usage: gpu_pmc_step.sh TAG [env...]
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run () { name=$1; shift; PROBE_STEPS=24 timeout 150 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/step_probe.py ${PROBE_ARGS:-} > $OUT/$name.log 2>&1 || echo "$name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
dur={}
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/*/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[(f.split('/')[-3], r['Dispatch_Id'])]=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
for f in glob.glob(out+'/*/*/*counter_collection.csv'):
    run=f.split('/')[-3]
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'logic' not in k and 'tile' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[k]['_ns_'+run].append(dur.get((run,r['Dispatch_Id']),0))
for k,cs in acc.items():
    print('#',k)
    for c,v in sorted(cs.items()):
        print('   %-24s %16.1f (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
