#!/bin/bash
mkdir -p gpurun_out/r2
{
echo "=== parity under forced sorting"
TH_BUCKET=1 TH_RESORT_STEPS=2 TH_REBUCKET_STEPS=2 timeout 300 python -m pytest -q -m gpu -x tests/test_gpu_logic_parity.py tests/test_gpu_fuzz.py 2>&1 | tail -5
for r in ${RS:-4 8}; do
  echo "=== TH_RESORT_STEPS=$r"
  TH_RESORT_STEPS=$r timeout 120 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids
done
echo "=== in-view, TH_RESORT_STEPS=8"
TH_RESORT_STEPS=8 timeout 120 python tools/step_probe.py --in-view 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2/probe3.log 2>&1
timeout 200 bash tools/gpu_probe4.sh >> gpurun_out/r2/probe3.log 2>&1
