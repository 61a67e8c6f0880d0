#!/bin/bash
# kernel trace of the frame loop (tools/frame_wall_probe.py) -> where the GPU stands still inside a frame (tools/frame_timeline.py)
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$GRAFT_REPO_ROOT/gpurun_out/frame_timeline; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o run -- python3 $GRAFT_REPO_ROOT/tools/frame_wall_probe.py ${1:-40} > $O/probe.log 2>&1
f=$(find /tmp/prof_tl -name '*kernel_trace.csv' | head -1)
cd $GRAFT_REPO_ROOT
python3 tools/frame_timeline.py "$f" ${2:-10} | tee $O/timeline.txt
