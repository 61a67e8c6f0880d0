# round 6, fourth GPU call: spanning lines a wave each (C4's draw), same-box kernel traces against round 5's library, the band sweep with the shared-memory barrier
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_binned_shapes.py tests/test_gpu_loopback.py tests/test_gpu_binned_draw.py tests/test_gpu_wide_lines.py -x -q -m gpu > $OUT/new_tests.log 2>&1; echo "rc=$?" >> $OUT/new_tests.log
TH_N=8192 TH_BENCH_TRACE=1 timeout 600 python tools/deposit_bench.py 100 --both > $OUT/c4_loop.txt 2>&1
cd /tmp
TH_N=8192 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 40 --both > $OUT/c4_trace.log 2>&1
TH_LIB=$GRAFT_REPO_ROOT/tools/bin/r5/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace_r5 -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/c3_trace_r5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace_head -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/c3_trace_head.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python tools/band_sweep.py --out $OUT/band_sweep.txt > $OUT/band_sweep.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
ls $OUT
