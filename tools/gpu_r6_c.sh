# round 6, third GPU call: the fixed tests, where the ordinary bins' blend spends its time, the C4 frame loop's kernels, same-box A/B against round 5's library
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_async_sort.py tests/test_gpu_binned_shapes.py tests/test_gpu_loopback.py tests/test_gpu_release.py -x -q -m gpu > $OUT/new_tests.log 2>&1; echo "rc=$?" >> $OUT/new_tests.log
TH_LIB=$PWD/tools/bin/libtendrils_hip_stamps.so timeout 300 python tools/blend_stamps.py 30 5 > $OUT/blend_stamps_first.txt 2>&1
TH_LIB=$PWD/tools/bin/libtendrils_hip_stamps.so timeout 300 python tools/blend_stamps.py 30 280 > $OUT/blend_stamps_crowded.txt 2>&1
for k in 1 2; do
  TH_LIB=$PWD/tools/bin/r5/libtendrils_hip.so timeout 300 python tools/deposit_bench.py 60 --both | grep '^{' >> $OUT/ab_r5.txt
  timeout 300 python tools/deposit_bench.py 60 --both | grep '^{' >> $OUT/ab_head.txt
done
cd /tmp
TH_N=8192 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 40 --both > $OUT/c4_trace.log 2>&1
cd $GRAFT_REPO_ROOT
TH_N=8192 TH_BENCH_TRACE=1 timeout 600 python tools/deposit_bench.py 100 --both > $OUT/c4_loop.txt 2>&1
TH_N=8192 TH_PIPE=stream timeout 600 python tools/deposit_bench.py 30 --both > $OUT/c4_loop_stream.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log
ls $OUT
