#!/bin/bash
# perf A/B: parity smoke + bench variants (prints ms/launch and value)
python -m pytest tests/test_gpu_logic_parity.py -m gpu -q -x 2>&1 | tail -2
for args in "--mode exact" "--mode fast" "--mode exact --flow-size 480x270" "--mode fast --flow-size 480x270" "$@"; do
  echo "== $args"; python bench.py --steps 100 --warmup 10 --no-cpu --no-traffic $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.2f G/s   launch %.4f ms   frac %.3f'%(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
