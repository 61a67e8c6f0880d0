#!/bin/bash
# same-box comparison of several library builds: interleaved rounds. usage: gpu_abn.sh MODE lib1 lib2 ...
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=$1; shift
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "round $r $MODE $lib: "
    TH_LIB=$PWD/tendrils_amd/lib/$lib python bench.py --no-cpu --no-traffic --mode $MODE 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.2f G/s  fused/step %.4f ms'%(d['value']/1e9, r['avg_launch_ms']/r['steps_per_launch']))"
  done
done
