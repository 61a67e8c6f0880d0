#!/bin/bash
# PMC passes over the frame loop of tools/deposit_bench.py (draw kernels). usage: [BENCH_ARGS='400 --both'] [PMC_GROUPS='sq1 sq2'] gpu_pmc_draw.sh TAG [pipeline]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-x}; PIPE=${2:-bins}
OUT=$ROOT/gpurun_out/pmc_$TAG
BENCH_ARGS=${BENCH_ARGS:-12}; GROUPS_WANTED=${PMC_GROUPS:-sq1 sq2 tcc mem mem2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run () { name=$1; shift; case " $GROUPS_WANTED " in *" $name "*) ;; *) return;; esac; TH_PIPE=$PIPE timeout 600 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/deposit_bench.py $BENCH_ARGS > $OUT/$name.log 2>&1 || echo "$name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE
run mem FETCH_SIZE SQ_INSTS_GDS SQ_INSTS_FLAT
run mem2 WRITE_SIZE TCC_ATOMIC_sum
cd $ROOT
python3 tools/pmc_summary.py $OUT
