#!/bin/bash
# long runs: frame loop (both passes), single steps, fused steps - counters must stay sane, nothing may fail
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}

echo "=== 1500 frames step + draw (both passes)"; timeout 300 python tools/deposit_bench.py 1500 --both 2>&1 | tail -1
echo "=== 1500 frames, all in view"; timeout 300 python tools/deposit_bench.py 1500 --both --in-view 2>&1 | tail -1
echo "=== 4096 single steps"; PROBE_STEPS=4096 timeout 300 python tools/step_probe.py 2>&1 | grep "single step" | tail -1
echo "=== bench 8192 fused steps"; timeout 300 python bench.py --steps 8192 --warmup 128 --no-cpu --no-traffic 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.1f G' % (d['value']/1e9), d['counters'], {k:round(v,3) for k,v in d['frame_loop'].items() if isinstance(v,float)})"
echo "=== drifting shapes, bins against stream-ordered, bit for bit"; for a in "3000 300 f32" "1080 600 f32" "3000 150 f16" "8192 60 f32"; do timeout 400 python tools/soak_shapes.py $a 2>&1 | tail -1; done
