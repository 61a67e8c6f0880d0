#!/bin/bash
# (GPU box) Same-box kernel traces of the C3 frame loop - tools/deposit_bench.py, both passes - with another commit's library
# (tools/bin/prev, built by tools/build_variant_libs.sh) and with this tree's: first frames (60) and the crowded target (400);
# where a workgroup of bins_blend_kernel spends its life (tools/blend_stamps.py over the -DTH_BLEND_STAMPS build).
# -> gpurun_out/ab_<tag>/   usage: tools/gpu_ab_prev.sh [tag] [rounds]
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_${1:-r6}
mkdir -p $OUT
cd /tmp
for k in $(seq 1 ${2:-2}); do
  TH_LIB=$ROOT/tools/bin/prev/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_prev_$k -- python3 $ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_prev_$k.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_head_$k -- python3 $ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_head_$k.log 2>&1
done
TH_LIB=$ROOT/tools/bin/prev/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_crowded_prev -- python3 $ROOT/tools/deposit_bench.py 400 --both > $OUT/trace_crowded_prev.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_crowded_head -- python3 $ROOT/tools/deposit_bench.py 400 --both > $OUT/trace_crowded_head.log 2>&1
cd $ROOT
TH_LIB=$ROOT/tools/bin/libtendrils_hip_stamps.so timeout 300 python3 tools/blend_stamps.py 30 5 > $OUT/blend_stamps_first.txt 2>&1
TH_LIB=$ROOT/tools/bin/libtendrils_hip_stamps.so timeout 300 python3 tools/blend_stamps.py 30 280 > $OUT/blend_stamps_crowded.txt 2>&1
for d in $OUT/trace_*/; do python3 tools/kernel_avgs.py $d bins_fused bins_blend_kernel bins_listed bins_span crowd_walk crowd_sort crowd_scatter logic_kernel; done | tee $OUT/summary.txt
