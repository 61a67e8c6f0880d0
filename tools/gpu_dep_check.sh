#!/bin/bash
# deposit parity tests + kernel trace of the frame loop
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_deposit.py tests/test_gpu_view.py tests/test_gpu_deposit_sharded.py -x -q -m gpu 2>&1 | tail -4
timeout 250 bash tools/gpu_trace_dep.sh ${1:-dep}
timeout 200 python tools/deposit_bench.py 100 2>&1 | tail -2
timeout 200 python tools/deposit_bench.py 100 --in-view 2>&1 | tail -2
