#!/bin/bash
# ISA of one kernel (default: the exact fused integrator) and an instruction histogram of its step loop
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
K=${1:-_ZN2th18logic_fused_kernelILb0ELb1ELb0ELb1ELb1ELb1EEEvNS_11LogicParamsE}
cd "$(dirname "$0")/../tendrils_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../../include -S --cuda-device-only -o /tmp/th_kernels.s th_kernels.hip 2>/dev/null
awk -v k="$K:" '$1==k{f=1} f{print} f&&/^\.Lfunc_end/{exit}' /tmp/th_kernels.s > /tmp/kernel.s
grep -A30 "^\s*.amdhsa_kernel $K" /tmp/th_kernels.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed" 
# inner loop = the deepest-nesting block range (Depth=2 header to its backedge)
hdr=$(grep -n "Inner Loop Header: Depth=2" /tmp/kernel.s | head -1 | cut -d: -f1)
start=$((hdr-1))
# hot part of the step loop: from its header to the first block label more than 300 lines further on
# (the reference-order fallback for out-of-domain lanes follows there)
end=$(awk -v s="$start" 'NR>s+300 && /^\.LBB/{print NR; exit}' /tmp/kernel.s)
echo "step loop lines $start..$end"
sed -n "${start},${end}p" /tmp/kernel.s | grep -E "^\s+[vsdg][a-z_]" | awk '{print $1}' | sed 's/_e32$//; s/_e64$//' | sort | uniq -c | sort -rn | awk '{printf "%s:%s  ", $2, $1} END{print ""}'
echo -n "VALU total: "; sed -n "${start},${end}p" /tmp/kernel.s | grep -cE "^\s+v_"
echo -n "VALU with SGPR source: "; sed -n "${start},${end}p" /tmp/kernel.s | grep -E "^\s+v_" | grep -vE "v_cndmask|v_cmp|v_readfirstlane" | grep -cE ", s[0-9]+|, s\["
