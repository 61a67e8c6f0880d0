#!/bin/bash
# round 4: same-box A/B of the crowded frame loop between the previous build (tools/bin/prev, TH_LIB) and the working tree
# Set up (build container): rm -rf tools/bin/prev && mkdir -p tools/bin/prev && git archive <commit> -- tendrils_amd/csrc include | tar -x -C tools/bin/prev
#                           && make -C tools/bin/prev/tendrils_amd/csrc ../lib/libtendrils_hip.so   (tools/bin is git-ignored and travels with gpurun;
#                           round 4 part g used 1a8f5e6..b6e2c88 = the head before that part: commit b6e2c88)
# usage: gpu_r4_crowd_ab.sh [rounds] [frames]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=${1:-3}; frames=${2:-600}
for k in $(seq 1 $rounds); do
  for which in prev head; do
    lib=""; [ $which = prev ] && lib=$ROOT/tools/bin/prev/tendrils_amd/lib/libtendrils_hip.so
    echo -n "$which run $k: "
    TH_LIB=$lib python3 $ROOT/tools/deposit_bench.py $frames --both --wall 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('step_ms','draw_both_ms','frames_per_s','wall_ms_per_frame','fragments_per_frame')})"
  done
done
