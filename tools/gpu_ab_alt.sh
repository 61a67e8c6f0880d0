#!/bin/bash
# the working tree's library against tools/bin/head/libtendrils_hip.so (the last commit, built by hand) on one box, interleaved:
# the frame loop (tools/deposit_bench.py N --both --wall): step, draw (events) and wall per frame
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for round in 1 2 3; do
  for v in head alt; do
    if [ $v = head ]; then export TH_LIB=$R/tools/bin/head/libtendrils_hip.so; else export TH_LIB=$R/tools/bin/alt/libtendrils_hip.so; fi
    for n in ${FRAMES:-40 300}; do
      echo "$v frames $n: $(timeout 300 python3 tools/deposit_bench.py $n --both --wall 2>&1 | tail -1 | grep -o '"step_ms": [0-9.]*\|"draw_both_ms": [0-9.]*\|"wall_ms_per_frame": [0-9.]*' | tr '\n' ' ')"
    done
  done
done
