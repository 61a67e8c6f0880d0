#!/bin/bash
# draw() (both passes): new lib vs libtendrils_hip_old.so on the same box, interleaved
L=tendrils_amd/lib
cp $L/libtendrils_hip.so /tmp/new.so
for round in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp $L/libtendrils_hip_old.so $L/libtendrils_hip.so; else cp /tmp/new.so $L/libtendrils_hip.so; fi
    echo "=== $v $(timeout 200 python tools/deposit_bench.py 100 --both 2>&1 | tail -1 | grep -o '"draw_both_ms": [0-9.]*') in-view $(timeout 200 python tools/deposit_bench.py 100 --both --in-view 2>&1 | tail -1 | grep -o '"draw_both_ms": [0-9.]*')"
  done
done
cp /tmp/new.so $L/libtendrils_hip.so
