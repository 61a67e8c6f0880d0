#!/bin/bash
# whole-job bench line + fused probe: new lib vs libtendrils_hip_old.so on the same box, interleaved
mkdir -p gpurun_out/r2
L=tendrils_amd/lib
cp $L/libtendrils_hip.so /tmp/new.so
for round in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp $L/libtendrils_hip_old.so $L/libtendrils_hip.so; else cp /tmp/new.so $L/libtendrils_hip.so; fi
    echo "=== $v"; timeout 120 python tools/fused_probe.py --short 2>&1 | grep "back-to-back" | tail -2
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-traffic --no-frame-loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.1f G  ms_per_step %.5f  kernel %.5f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['ms_per_step']))"
  done
done
cp /tmp/new.so $L/libtendrils_hip.so
