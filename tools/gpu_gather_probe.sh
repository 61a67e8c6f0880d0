#!/bin/bash
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_logic_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
echo "=== sorted"; PROBE_STEPS=256 timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -1
echo "=== texel order"; TH_BUCKET=0 PROBE_STEPS=64 timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -2
timeout 200 python tools/deposit_bench.py 60 2>&1 | tail -1
