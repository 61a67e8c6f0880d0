#!/bin/bash
# round-2 evidence for the single-step launch: loop timings, kernel trace, PMC passes
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out/r2
O=gpurun_out/r2/r2_e_single_step.txt
{
echo "# tools/step_probe.py at C3 (4096^2 particles, flow 1920x1080, exact, default uniforms unless noted); same box"
echo "## default: gathered taps over tile-sorted slots, re-sort every 64"; PROBE_STEPS=256 timeout 120 python tools/step_probe.py 2>&1 | grep "single step"
echo "## TH_BUCKET=0: texel order"; TH_BUCKET=0 PROBE_STEPS=128 timeout 120 python tools/step_probe.py 2>&1 | grep "single step"
for R in 16 32 128; do echo "## default, TH_RESORT_STEPS=$R"; TH_RESORT_STEPS=$R PROBE_STEPS=256 timeout 120 python tools/step_probe.py 2>&1 | grep "single step" | tail -1; done
echo "## default, flow-only"; PROBE_STEPS=128 timeout 120 python tools/step_probe.py --flow-only 2>&1 | grep "single step" | tail -1
echo "## default, all particles in view"; PROBE_STEPS=128 timeout 120 python tools/step_probe.py --in-view 2>&1 | grep "single step" | tail -1
echo "# rocprofv3 --kernel-trace --stats of the default probe (128 steps)"
PROBE_STEPS=128 bash tools/gpu_trace_step.sh r2e 2>&1 | grep -E "logic|tile|flow_decode"
echo "# rocprofv3 --pmc passes of the default probe (24 steps), averages per dispatch"
bash tools/gpu_pmc_step.sh r2e 2>&1 | grep -v "^W2026\|^E2026"
} > $O 2>&1
tail -5 $O
