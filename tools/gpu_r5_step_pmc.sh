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
run () { name=$1; shift; PROBE_STEPS=24 timeout 200 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/step_probe.py ${PROBE_ARGS:-} > $OUT/$name.log 2>&1 || echo "$name failed"; }
run ea1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum GRBM_GUI_ACTIVE
run ea2 TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_GMI_32B_sum
run ea3 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum
run ta  TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
run td  TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE
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
