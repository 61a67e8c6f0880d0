#!/bin/bash
# round 5: where the single-step launch's time goes.  PMC passes (counters only, one group per run) over tools/step_probe.py:
# DRAM-destined requests at the L2's memory side (HBM apart from Infinity-Cache-served taps), the vector-memory path (TA / TCP
# busy and stalls), the sequencer.  usage: gpu_r5_step_pmc.sh TAG  [TH_STEP_VARIANT etc. in the environment]
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_pmc/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (per pass: at most 4 TCC counters, 8 SQ, 2 of the blocks whose slot counts MI355X_MICROARCH.md does not list; a pass that asks for
# more dies in rocprofiler_create_counter_config and then hangs until its timeout)
run () { name=$1; shift; PROBE_STEPS=24 timeout 100 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/step_probe.py ${PROBE_ARGS:-} > $OUT/$name.log 2>&1 || echo "$name failed"; }
run ea1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum GRBM_GUI_ACTIVE
run ea2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum
run ta  TA_TA_BUSY_sum TA_BUSY_avr
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY' | tee $OUT/summary.txt
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
        if 'logic_kernel' not in k and 'logic_ring' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[k]['_ns_'+run].append(dur.get((run,r['Dispatch_Id']),0))
for k,cs in acc.items():
    print('#',k)
    for c,v in sorted(cs.items()):
        print('   %-40s %16.1f (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
