#!/bin/bash
# full GPU suite + the driver's bench command + frame loop
mkdir -p gpurun_out/r2
timeout 1400 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r2/pytest_gpu.log | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2/bench_driver.json 2> gpurun_out/r2/bench_driver.err; tail -c 300 gpurun_out/r2/bench_driver.json
timeout 200 python tools/deposit_bench.py 100 2>&1 | tail -1
PROBE_STEPS=256 timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -1
