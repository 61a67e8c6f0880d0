#!/bin/bash
# the GPU suite under the alternative paths the environment switches select
mkdir -p gpurun_out/r2
for v in "TH_DRAW_REUSE=0" "TH_SINGLE=window" "TH_BUCKET=1" "TH_BUCKET=0" "TH_FUSE=0"; do
  echo "=== $v"
  env $v timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2/variant.log 2>&1; grep -E "passed|failed|error" gpurun_out/r2/variant.log | tail -2
done
