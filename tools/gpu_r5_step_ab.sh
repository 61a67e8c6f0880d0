#!/bin/bash
# round 5: single-step integrator variants (TH_STEP_VARIANT: 0 = one texel prefetched in registers, 1 / 2 = register pipelines,
# (the variants live in the tree of commit 0dc8af8 - `git show 0dc8af8:tendrils_amd/csrc/th_kernels.hip`; none was faster and the
# product kept round 4's kernel: profiles/r5_b_single_step_variants.txt)
# 3 / 4 = LDS ring, 512 / 1024-thread workgroups) on one box: parity first, then the step probe.
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_step; mkdir -p $O
for v in ${VARIANTS:-0 1 2 3 4}; do
  echo "=== parity TH_STEP_VARIANT=$v"
  TH_STEP_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_logic_parity.py tests/test_gpu_bucketed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
done 2>&1 | tee $O/parity.txt
for round in 1 2; do
  for v in ${VARIANTS:-0 1 2 3 4}; do
    for g in ${GRIDS:-default}; do
      export TH_STEP_VARIANT=$v
      if [ $g != default ]; then export TH_RING_GRID=$g TH_STEP_GRID=$g; else unset TH_RING_GRID TH_STEP_GRID; fi
      echo "=== variant $v grid $g: default"; timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -2
      echo "=== variant $v grid $g: flow-only"; timeout 120 python tools/step_probe.py --flow-only 2>&1 | grep "single step" | tail -1
    done
  done
done 2>&1 | tee $O/probe.txt
