set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6a
# sensitivity of the new advisor test: must FAIL on the old th_step
TH_LIB=$PWD/tools/bin/libtendrils_hip_oldstep.so timeout 300 python -m pytest tests/test_gpu_async_sort.py -q -m gpu -k replayed_graph > gpurun_out/r6a/oldstep_test.log 2>&1; echo "old rc=$?" >> gpurun_out/r6a/oldstep_test.log
timeout 300 python -m pytest tests/test_gpu_async_sort.py -q -m gpu -k replayed_graph > gpurun_out/r6a/newstep_test.log 2>&1; echo "new rc=$?" >> gpurun_out/r6a/newstep_test.log
timeout 900 python tools/band_sweep.py --out gpurun_out/r6a/band_sweep.txt > gpurun_out/r6a/band_sweep.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a/bench_driver.json 2> gpurun_out/r6a/bench_driver.err
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r6a/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6a/gpu_tests.log
tail -5 gpurun_out/r6a/*.log
