#!/bin/bash
# regime matrix: layout x arithmetic x data distribution (per-launch logic_kernel ms)
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for data in "" "--in-view"; do
 for b in 0 1; do
  for m in "--mode exact" "--mode fast" "--flow-only --steps 40 --warmup 4"; do
    echo -n "data[${data:-default}] TH_BUCKET=$b $m: "
    TH_BUCKET=$b python bench.py --steps 80 --warmup 8 --no-cpu --no-traffic $data $m 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel %.4f ms  step %.4f ms'%(d['roofline']['avg_launch_ms'], d['roofline']['avg_step_ms_on_stream']))"
  done
 done
done
