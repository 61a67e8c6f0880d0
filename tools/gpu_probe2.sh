#!/bin/bash
mkdir -p gpurun_out/r2
{
for g in 2048 2560 5120 10240 20480 65536; do
  echo "=== TH_FUSED_GRID=$g"
  TH_FUSED_GRID=$g python tools/fused_probe.py --short 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r2/probe2.log 2>&1
