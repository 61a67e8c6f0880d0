#!/bin/bash
# A/B of deposit rasteriser variants on one box: current build, then kWindow = 4 for hexagons (rebuilt on the box)
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_deposit.py tests/test_gpu_view.py -x -q -m gpu 2>&1 | tail -2
timeout 250 bash tools/gpu_trace_dep.sh depA
sed -i 's/constexpr int kWindow = 8;/constexpr int kWindow = N == 6 ? 4 : 8;/' tendrils_amd/csrc/th_deposit.hip
(cd tendrils_amd/csrc && make 2>&1 | tail -1)
timeout 300 python -m pytest tests/test_gpu_deposit.py -x -q -m gpu 2>&1 | tail -2
timeout 250 bash tools/gpu_trace_dep.sh depB
