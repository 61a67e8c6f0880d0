#!/bin/bash
# the frame loop with and without an environment switch ($1=NAME): tools/deposit_bench.py N --both --wall, interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for round in 1 2 3; do
  for v in off on; do
    if [ $v = on ]; then export $1=1; else unset $1; fi
    for n in ${FRAMES:-40 300}; do
      echo "$1 $v frames $n: $(timeout 300 python3 tools/deposit_bench.py $n --both --wall 2>&1 | tail -1 | grep -o '"step_ms": [0-9.]*\|"draw_both_ms": [0-9.]*\|"wall_ms_per_frame": [0-9.]*' | tr '\n' ' ')"
    done
  done
done
