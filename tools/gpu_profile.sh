#!/bin/bash
# rocprofv3 passes over a short bench run: kernel trace + PMC groups (one run per group).
# Usage: bash tools/gpu_profile.sh [extra bench args]   -> gpurun_out/prof_<tag>/
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r1}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 64 --warmup 32 --no-cpu --no-traffic $*"
cd /tmp
echo "== kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 64 --warmup 32 --no-cpu --no-traffic $* > $OUT/trace.log 2>&1
run_pmc () {
  name=$1; shift
  echo "== pmc $name: $*"
  timeout 600 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || echo "pmc $name failed"
}
run_pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run_pmc sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $ROOT
python3 tools/summarize_pmc.py $OUT | tee $OUT/summary.txt
