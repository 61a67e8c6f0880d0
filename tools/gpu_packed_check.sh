#!/bin/bash
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_packed_state.py tests/test_gpu_bucketed.py -x -q -m gpu > gpurun_out/r2/packed.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r2/packed.log | tail -3
TH_BUCKET=1 timeout 900 python -m pytest tests/test_gpu_packed_state.py -x -q -m gpu > gpurun_out/r2/packed1.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r2/packed1.log | tail -3
for b in 0 2; do
  echo "=== c5 N=1 TH_BUCKET=$b"; TH_BUCKET=$b timeout 400 python bench.py --config c5 --steps 128 --warmup 32 --no-cpu --no-traffic 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.1f G ms/step %.5f kernel %.5f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['ms_per_step']))"
done
echo "=== c3 --state f16"; timeout 400 python bench.py --state f16 --steps 256 --warmup 32 --no-cpu --no-traffic --no-frame-loop 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.1f G ms/step %.5f kernel %.5f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['ms_per_step']))"
