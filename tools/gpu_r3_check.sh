#!/bin/bash
# round 3: full GPU suite, the driver's bench command, the RCCL path at world size 1
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"; tail -c 1500 $OUT/bench_driver.json; tail -5 $OUT/bench_driver.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-cpu --no-frame-loop > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; echo "forcedist rc $?"; tail -c 600 $OUT/bench_forcedist.json; tail -5 $OUT/bench_forcedist.err
