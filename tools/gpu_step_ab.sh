#!/bin/bash
# single-step kernel: new lib vs libtendrils_hip_old.so on the same box
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out/r2
L=tendrils_amd/lib
timeout 400 python -m pytest tests/test_gpu_bucketed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
cp $L/libtendrils_hip.so /tmp/new.so
for round in 1 2; do
  for v in new old; do
    if [ $v = old ]; then cp $L/libtendrils_hip_old.so $L/libtendrils_hip.so; else cp /tmp/new.so $L/libtendrils_hip.so; fi
    echo "=== $v default"; timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -2
    echo "=== $v flow-only"; timeout 120 python tools/step_probe.py --flow-only 2>&1 | grep "single step" | tail -1
    echo "=== $v no re-sort (TH_RESORT_STEPS=100000, 16 steps)"; PROBE_STEPS=16 TH_RESORT_STEPS=100000 timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | head -2
  done
done
cp /tmp/new.so $L/libtendrils_hip.so
