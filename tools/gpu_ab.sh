#!/bin/bash
# same-box A/B of two library builds: interleaved rounds (bench.py whole job + fused per-step kernel time)
A=${1:-libtendrils_hip_prev.so}; B=${2:-libtendrils_hip.so}; MODE=${3:-exact}
for r in 1 2 3; do
  for lib in $A $B; do
    echo -n "round $r $lib: "
    TH_LIB=$PWD/tendrils_amd/lib/$lib python bench.py --no-cpu --no-traffic --mode $MODE 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.2f G/s  fused/step %.4f ms'%(d['value']/1e9, r['avg_launch_ms']/r['steps_per_launch']))"
  done
done
