#!/bin/bash
# whole-job bench line + fused probe: new lib vs libtendrils_hip_old.so on the same box, interleaved
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out/r2
L=tendrils_amd/lib
cp $L/libtendrils_hip.so /tmp/new.so
timeout 600 python -m pytest tests/test_gpu_logic_parity.py tests/test_gpu_bucketed.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for round in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp $L/libtendrils_hip_old.so $L/libtendrils_hip.so; else cp /tmp/new.so $L/libtendrils_hip.so; fi
    echo "=== $v"; timeout 120 python tools/fused_probe.py --short 2>&1 | grep "back-to-back" | tail -2
  done
done
cp /tmp/new.so $L/libtendrils_hip.so
