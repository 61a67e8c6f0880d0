# round 6, fifth GPU call: the windowed blend of the ordinary bins (A/B, stamps), the emit's plain path back, draw suites
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6e
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_binned_shapes.py tests/test_gpu_binned_draw.py tests/test_gpu_deposit.py tests/test_gpu_view.py tests/test_gpu_fuzz.py tests/test_gpu_scene.py tests/test_gpu_loopback.py -x -q -m gpu > $OUT/draw_tests.log 2>&1; echo "rc=$?" >> $OUT/draw_tests.log
for k in 1 2; do
  TH_LIB=$PWD/tools/bin/r5/libtendrils_hip.so timeout 300 python tools/deposit_bench.py 60 --both | grep '^{' >> $OUT/ab_r5.txt
  TH_EXP_BLEND_WIN=0 timeout 300 python tools/deposit_bench.py 60 --both | grep '^{' >> $OUT/ab_head_nowin.txt
  TH_EXP_BLEND_WIN=1 timeout 300 python tools/deposit_bench.py 60 --both | grep '^{' >> $OUT/ab_head_win.txt
done
TH_LIB=$PWD/tools/bin/libtendrils_hip_stamps.so TH_EXP_BLEND_WIN=1 timeout 300 python tools/blend_stamps.py 30 5 > $OUT/blend_stamps_win_first.txt 2>&1
TH_LIB=$PWD/tools/bin/libtendrils_hip_stamps.so TH_EXP_BLEND_WIN=1 timeout 300 python tools/blend_stamps.py 30 280 > $OUT/blend_stamps_win_crowded.txt 2>&1
cd /tmp
TH_EXP_BLEND_WIN=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace_nowin -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/c3_trace_nowin.log 2>&1
TH_EXP_BLEND_WIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace_win -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/c3_trace_win.log 2>&1
TH_EXP_BLEND_WIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace_win_crowded -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 400 --both > $OUT/c3_trace_win_crowded.log 2>&1
TH_EXP_BLEND_WIN=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace_nowin_crowded -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 400 --both > $OUT/c3_trace_nowin_crowded.log 2>&1
cd $GRAFT_REPO_ROOT
ls $OUT
