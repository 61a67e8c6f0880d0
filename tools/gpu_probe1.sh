#!/bin/bash
# round-2 probe: launch length / idle / grid size
mkdir -p gpurun_out/r2
{
for g in 0 1280 2560 1536 1024; do
  echo "=== TH_FUSED_GRID=$g"
  TH_FUSED_GRID=$g python tools/fused_probe.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r2/probe1.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/pmc_clk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 32 --no-cpu --no-traffic > $GRAFT_REPO_ROOT/gpurun_out/r2/pmc_clk.log 2>&1
