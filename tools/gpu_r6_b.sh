set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_gpu_async_sort.py tests/test_gpu_binned_shapes.py -x -q -m gpu > gpurun_out/r6b/new_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6b/new_tests.log
timeout 900 python -m pytest tests/test_gpu_loopback.py -x -q -m gpu > gpurun_out/r6b/loopback.log 2>&1; echo "rc=$?" >> gpurun_out/r6b/loopback.log
timeout 600 python tools/band_sweep.py --out gpurun_out/r6b/band_sweep.txt > gpurun_out/r6b/band_sweep.log 2>&1
TH_EXP_FUSED_PAD=20480 timeout 600 python tools/band_sweep.py --out gpurun_out/r6b/band_sweep_pad4.txt > gpurun_out/r6b/band_sweep_pad4.log 2>&1
TH_EXP_FUSED_PAD=8192 timeout 600 python tools/band_sweep.py --worlds 8 --out gpurun_out/r6b/band_sweep_pad8k.txt > gpurun_out/r6b/band_sweep_pad8k.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6b/bench_driver.json 2> gpurun_out/r6b/bench_driver.err
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6b/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6b/gpu_tests.log
tail -5 gpurun_out/r6b/*.log
