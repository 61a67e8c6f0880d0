#!/bin/bash
# round 3: the ordinary bins' blend before (1) / after (0) the crowded bins' kernels: 600-frame loops, averages per 50 frames
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for e in 0 1 auto; do
  if [ $e = auto ]; then unset TH_BINS_EARLY; else export TH_BINS_EARLY=$e; fi
  echo "== TH_BINS_EARLY=$e"
  TH_BENCH_TRACE=1 timeout 300 python3 tools/deposit_bench.py 600 --both 2>&1 | grep "^frames\|^{" | cut -c1-200
done
