#!/bin/bash
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_logic_parity.py tests/test_gpu_fuzz.py tests/test_gpu_packed_state.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
echo "=== sorted"; PROBE_STEPS=256 timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -2
echo "=== flow-only"; PROBE_STEPS=128 timeout 120 python tools/step_probe.py --flow-only 2>&1 | grep "single step" | tail -1
echo "=== in-view"; PROBE_STEPS=128 timeout 120 python tools/step_probe.py --in-view 2>&1 | grep "single step" | tail -1
PROBE_STEPS=128 bash tools/gpu_trace_step.sh xcd 2>&1 | grep -E "logic_kernel"
