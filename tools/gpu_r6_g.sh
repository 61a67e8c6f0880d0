# round 6, seventh GPU call: emit without the span test, blend arms 0 / 1 again (rank loop scalar), C4
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_binned_shapes.py tests/test_gpu_binned_draw.py tests/test_gpu_wide_lines.py tests/test_gpu_deposit.py -x -q -m gpu > $OUT/draw_tests.log 2>&1; echo "rc=$?" >> $OUT/draw_tests.log
cd /tmp
for k in 1 2; do
TH_LIB=$GRAFT_REPO_ROOT/tools/bin/r5/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r5_$k -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_r5_$k.log 2>&1
for v in 0 1; do
  TH_EXP_BLEND=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_v${v}_$k -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_v${v}_$k.log 2>&1
done
done
TH_N=8192 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 40 --both > $OUT/c4_trace.log 2>&1
cd $GRAFT_REPO_ROOT
TH_N=8192 TH_BENCH_TRACE=1 timeout 600 python tools/deposit_bench.py 100 --both > $OUT/c4_loop.txt 2>&1
ls $OUT
