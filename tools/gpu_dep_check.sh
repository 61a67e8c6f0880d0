#!/bin/bash
# deposit parity tests + kernel trace of the frame loop
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_deposit.py tests/test_gpu_view.py tests/test_gpu_deposit_sharded.py tests/test_gpu_scene.py tests/test_node_host.py tests/test_capi_exports.py -x -q -m gpu > gpurun_out/r2/dep_tests.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r2/dep_tests.log | tail -3
timeout 250 bash tools/gpu_trace_dep.sh ${1:-dep}
timeout 200 python tools/deposit_bench.py 100 --both 2>&1 | tail -1
timeout 200 python tools/deposit_bench.py 100 --in-view --both 2>&1 | tail -1
