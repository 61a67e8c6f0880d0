#!/bin/bash
# (build container) The libraries the same-box comparisons load through TH_LIB - built here, they travel to the GPU box with the tree
# (tools/bin/ is git-ignored, not gpurun-ignored):
#   tools/bin/prev/libtendrils_hip.so        the library of another commit (default: the round-5 tree e4e4c4d), from a git worktree
#   tools/bin/libtendrils_hip_stamps.so      this tree's with -DTH_BLEND_STAMPS (tools/blend_stamps.py)
# usage: tools/build_variant_libs.sh [commit]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
COMMIT=${1:-e4e4c4d}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DTH_TESTING=1"
make -C $ROOT/tendrils_amd/csrc -j6 all > /dev/null
mkdir -p $ROOT/tools/bin/prev
TREE=$(mktemp -d)
git -C $ROOT worktree add -f $TREE $COMMIT -q
make -C $TREE/tendrils_amd/csrc -j6 ../lib/libtendrils_hip.so > /dev/null
cp $TREE/tendrils_amd/lib/libtendrils_hip.so $ROOT/tools/bin/prev/
git -C $ROOT worktree remove --force $TREE
cd $ROOT/tendrils_amd/csrc
/opt/rocm/bin/hipcc $FLAGS -DTH_BLEND_STAMPS=1 -c -o /tmp/th_bins_stamps.o th_bins.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libtendrils_hip_stamps.so $(ls ../lib/obj/*.o | grep -v th_bins.o) /tmp/th_bins_stamps.o -ldl
ls -la $ROOT/tools/bin/prev/libtendrils_hip.so $ROOT/tools/bin/libtendrils_hip_stamps.so
