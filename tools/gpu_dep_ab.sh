#!/bin/bash
# draw(): new lib vs libtendrils_hip_old.so on the same box, interleaved; 300 frames (the wake has formed: long runs)
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=tendrils_amd/lib
cp $L/libtendrils_hip.so /tmp/new.so
timeout 600 python -m pytest tests/test_gpu_deposit.py tests/test_gpu_view.py tests/test_gpu_fuzz.py tests/test_gpu_deposit_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for round in 1 2; do
  for v in new old; do
    if [ $v = old ]; then cp $L/libtendrils_hip_old.so $L/libtendrils_hip.so; else cp /tmp/new.so $L/libtendrils_hip.so; fi
    echo "=== $v $(timeout 200 python tools/deposit_bench.py 300 2>&1 | tail -1 | grep -o '"draw_ms": [0-9.]*, "view_ms": [0-9.]*') in-view $(timeout 200 python tools/deposit_bench.py 300 --in-view 2>&1 | tail -1 | grep -o '"draw_ms": [0-9.]*, "view_ms": [0-9.]*')"
  done
done
cp /tmp/new.so $L/libtendrils_hip.so
