#!/bin/bash
# Standard GPU-box pass: parity tests, smoke, microbenchmarks, bench, rocprof kernel trace.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [quick]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== node: $(node --version 2>&1 | head -1)   nproc: $(nproc)"
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
if [ "${1:-}" != "quick" ]; then
  echo "== microbench"
  timeout 300 tools/bin/microbench 2>&1 | tee gpurun_out/microbench.log
fi
echo "== bench exact"
timeout 900 python bench.py > gpurun_out/bench_exact.log 2>&1; tail -2 gpurun_out/bench_exact.log
echo "== bench exact flow-only"
timeout 600 python bench.py --steps 64 --warmup 32 --flow-only --no-cpu > gpurun_out/bench_exact_flowonly.log 2>&1; tail -2 gpurun_out/bench_exact_flowonly.log
echo "== bench fast"
timeout 600 python bench.py --steps 256 --warmup 32 --mode fast --no-cpu > gpurun_out/bench_fast.log 2>&1; tail -2 gpurun_out/bench_fast.log
echo "== bench force-dist (RCCL path at world size 1)"
timeout 600 python bench.py --steps 128 --warmup 32 --force-dist --no-cpu > gpurun_out/bench_forcedist.log 2>&1; tail -12 gpurun_out/bench_forcedist.log
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-traffic > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name '*stats*' | head; for f in $(find gpurun_out/prof -name '*kernel_stats.csv'); do head -12 $f; done
