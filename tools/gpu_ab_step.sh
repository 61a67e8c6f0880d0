#!/bin/bash
# single-step launch: tools/bin/head against tools/bin/alt on one box, interleaved (tools/step_probe.py: kernel ms, loop ms)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for round in 1 2 3; do
  for v in head alt; do
    export TH_LIB=$R/tools/bin/$v/libtendrils_hip.so
    echo "$v: $(timeout 300 python3 tools/step_probe.py 2>&1 | tail -3 | tr '\n' ' ' | cut -c1-300)"
  done
done
