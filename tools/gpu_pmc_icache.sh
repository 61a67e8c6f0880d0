#!/bin/bash
# instruction-fetch counters of the single-step probe
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQC_|INST_CYCLES|SQ_INST_LEVEL|SQ_WAIT_INST|SQ_LEVEL" | head -60 > $OUT/list.txt
run () { name=$1; shift; PROBE_STEPS=24 timeout 150 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/step_probe.py ${PROBE_ARGS:-} > $OUT/$name.log 2>&1 || echo "$name failed"; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run ic2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'logic' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,cs in acc.items():
    print('#',k)
    for c,v in sorted(cs.items()):
        print('   %-24s %16.1f (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
cat $OUT/list.txt | head -40
tail -3 $OUT/ic1.log
