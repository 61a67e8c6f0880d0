#!/bin/bash
mkdir -p gpurun_out/r2
timeout 600 python -m pytest -q -m gpu -x tests/test_gpu_deposit_sharded.py tests/test_gpu_deposit.py tests/test_gpu_view.py tests/test_capi_exports.py 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
echo "=== frame loop, sharded draw at world size 1"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/frame_bench_dist.py 2>&1 | tail -1
timeout 200 python tools/deposit_bench.py 100 2>&1 | tail -1
