#!/bin/bash
# PMC passes over the frame loop: the sort / deposit kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/pmc_sort
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run () { name=$1; shift; timeout 200 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 12 > $OUT/$name.log 2>&1 || echo "$name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
run mem FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('th::(anonymous namespace)::','').split('(')[0].replace('void ','')
        if not any(t in k for t in ('radix','deposit_','colscan')): continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,cs in sorted(acc.items()):
    print('#',k)
    for c,v in sorted(cs.items()):
        print('   %-24s %16.1f (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
