# round 6: the emit's plain path against round 5's (literal loads), blend arms 0 / 1 with the rounds cut, then the whole GPU suite and the driver's command
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6i
mkdir -p $OUT
cd /tmp
for k in 1 2; do
TH_LIB=$GRAFT_REPO_ROOT/tools/bin/r5/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r5_$k -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_r5_$k.log 2>&1
for v in 0 1; do
  TH_EXP_BLEND=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_v${v}_$k -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_v${v}_$k.log 2>&1
done
done
TH_LIB=$GRAFT_REPO_ROOT/tools/bin/r5/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_crowded_r5 -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 400 --both > $OUT/trace_crowded_r5.log 2>&1
TH_EXP_BLEND=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_crowded_v1 -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 400 --both > $OUT/trace_crowded_v1.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
ls $OUT
