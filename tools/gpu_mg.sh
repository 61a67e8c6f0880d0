#!/bin/bash
mkdir -p gpurun_out/r2
{
timeout 900 python -m pytest -q -m gpu -x tests/test_gpu_packed_state.py tests/test_gpu_deposit_sharded.py 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -8
echo "=== frame loop, sharded draw at world size 1"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/frame_bench_dist.py 2>&1 | tail -3
echo "=== bench c4 (N=1)"
timeout 400 python bench.py --config c4 --steps 64 --warmup 16 --no-cpu --no-traffic 2>&1 | tail -1 | cut -c1-1200
echo "=== bench c5 (N=1)"
timeout 400 python bench.py --config c5 --steps 64 --warmup 16 --no-cpu --no-traffic 2>&1 | tail -1 | cut -c1-1200
echo "=== bench c3 forced dist (RCCL at world size 1)"
timeout 400 python bench.py --steps 64 --warmup 32 --no-cpu --no-traffic --force-dist 2>&1 | tail -1 | cut -c1-600
} > gpurun_out/r2/mg.log 2>&1
