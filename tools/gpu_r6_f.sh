# round 6, sixth GPU call: the ordinary bins' blend - variants side by side on one box (kernel traces); C4 with the span kernel
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_binned_shapes.py tests/test_gpu_binned_draw.py -x -q -m gpu > $OUT/draw_tests.log 2>&1; echo "rc=$?" >> $OUT/draw_tests.log
cd /tmp
TH_LIB=$GRAFT_REPO_ROOT/tools/bin/r5/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r5 -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_r5.log 2>&1
for v in 0 1 2; do
  TH_EXP_BLEND=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_v$v -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_v$v.log 2>&1
done
for v in 0 1 2; do
  TH_EXP_BLEND=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_crowded_v$v -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 400 --both > $OUT/trace_crowded_v$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
TH_N=8192 TH_BENCH_TRACE=1 timeout 600 python tools/deposit_bench.py 100 --both > $OUT/c4_loop.txt 2>&1
cd /tmp
TH_N=8192 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 40 --both > $OUT/c4_trace.log 2>&1
ls $OUT
