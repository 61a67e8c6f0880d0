#!/bin/bash
# PMC passes over bench.py's --pmc-child run (the launches the bench times: fused noise-on, single steps, fused flow-only).
# usage: gpu_pmc_fused.sh TAG [launch length]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-x}; LEN=${2:-20}
OUT=$ROOT/gpurun_out/r3/pmcf_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run () { name=$1; shift; timeout 200 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --pmc-child $LEN --no-cpu --no-traffic > $OUT/$name.log 2>&1 || echo "$name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE
run mem FETCH_SIZE SQ_INSTS_SMEM SQ_INSTS_FLAT
run mem2 WRITE_SIZE SQ_IFETCH SQ_INSTS_BRANCH
cd $ROOT
python3 tools/pmc_summary.py $OUT logic_ tile_ flow_decode
