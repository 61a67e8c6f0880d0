#!/bin/bash
# VGPRs / SGPRs / LDS / scratch of the kernels of one object of tendrils_amd/lib/obj (the code object's metadata notes):
#   tools/kernel_regs.sh th_bins [pattern]
obj=$(dirname "$0")/../tendrils_amd/lib/obj/$1.o
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.o --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.o | python3 -c "
import sys,re
t=sys.stdin.read()
pat=sys.argv[1] if len(sys.argv)>1 else ''
for m in re.finditer(r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', t, re.S):
    lds,name,scr,sg,vg=m.groups()
    if pat in name: print('%-110s vgpr %3s sgpr %3s lds %6s scratch %4s'%(name[:110],vg,sg,lds,scr))
" "$2"
rm -rf $tmp
