#!/bin/bash
# quick GPU pass: debug script + parity tests + microbench + 2 bench lines
set -u
mkdir -p gpurun_out
echo "== debug_ood"; timeout 300 python tools/debug_ood.py 2>&1 | tail -30
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -8 gpurun_out/pytest_gpu.log
echo "== generic kernel (TH_FORCE_GENERIC=1) on the golden fixtures"
TH_FORCE_GENERIC=1 timeout 600 python -m pytest tests/test_gpu_logic_parity.py -m gpu -q -k "exact_mode" > gpurun_out/pytest_generic.log 2>&1; tail -4 gpurun_out/pytest_generic.log
echo "== microbench"; timeout 300 tools/bin/microbench 2>&1 | grep -E "policy|copy" | tee gpurun_out/microbench2.log
echo "== bench exact"; timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu > gpurun_out/bench_exact.log 2>&1; tail -1 gpurun_out/bench_exact.log
echo "== bench fast"; timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu --mode fast > gpurun_out/bench_fast.log 2>&1; tail -1 gpurun_out/bench_fast.log
echo "== bench force-dist"; timeout 600 python bench.py --steps 50 --warmup 5 --force-dist --no-cpu > gpurun_out/bench_forcedist.log 2>&1; tail -12 gpurun_out/bench_forcedist.log
