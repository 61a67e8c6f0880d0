#!/bin/bash
# whole-job and fused per-step kernel time vs bucketing policy and re-sort interval (default data, C3)
show() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.2f G/s  fused/step %.4f ms'%(d['value']/1e9, r['avg_launch_ms']/r['steps_per_launch']))"; }
for mode in exact fast; do
  echo -n "$mode TH_BUCKET=0: "; TH_BUCKET=0 python bench.py --steps 1024 --no-cpu --no-traffic --mode $mode 2>&1 | show
  for iv in 128 256 512 1024; do
    echo -n "$mode TH_BUCKET=1 every $iv: "; TH_BUCKET=1 TH_REBUCKET_STEPS=$iv python bench.py --steps 1024 --no-cpu --no-traffic --mode $mode 2>&1 | show
  done
done
