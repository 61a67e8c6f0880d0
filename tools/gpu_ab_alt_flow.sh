#!/bin/bash
# the flow pass and the view pass one after the other (tools/deposit_bench.py N): tools/bin/head against tools/bin/alt, interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for round in 1 2 3; do
  for v in head alt; do
    export TH_LIB=$R/tools/bin/$v/libtendrils_hip.so
    for n in ${FRAMES:-40 300}; do
      echo "$v frames $n: $(timeout 300 python3 tools/deposit_bench.py $n 2>&1 | tail -1 | grep -o '"step_ms": [0-9.]*\|"draw_ms": [0-9.]*\|"view_ms": [0-9.]*' | tr '\n' ' ')"
    done
  done
done
