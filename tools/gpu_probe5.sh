#!/bin/bash
mkdir -p gpurun_out/r2
{
echo "=== parity under forced sorting"
TH_BUCKET=1 TH_RESORT_STEPS=2 TH_REBUCKET_STEPS=2 timeout 300 python -m pytest -q -m gpu -x tests/test_gpu_logic_parity.py tests/test_gpu_fuzz.py 2>&1 | tail -3
TH_BUCKET=1 TH_RESORT_STEPS=3 TH_REBUCKET_STEPS=5 timeout 300 python -m pytest -q -m gpu -x tests/test_gpu_logic_parity.py tests/test_gpu_fuzz.py 2>&1 | tail -3
echo "=== noise on, R=8"; TH_RESORT_STEPS=8 bash tools/gpu_trace_step.sh a
for r in 4 8; do echo "=== loop R=$r"; TH_RESORT_STEPS=$r timeout 120 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids; done
echo "=== loop texel"; TH_BUCKET=0 timeout 120 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids
echo "=== loop in-view R=8"; TH_RESORT_STEPS=8 timeout 120 python tools/step_probe.py --in-view 2>&1 | grep -v amdgpu.ids
echo "=== loop flow-only R=8"; TH_RESORT_STEPS=8 timeout 120 python tools/step_probe.py --flow-only 2>&1 | grep -v amdgpu.ids
echo "=== loop flow-only texel"; TH_BUCKET=0 timeout 120 python tools/step_probe.py --flow-only 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2/probe5.log 2>&1
