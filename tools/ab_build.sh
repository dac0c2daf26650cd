#!/bin/bash
# Builds the C-ABI library of another revision next to the working tree's, for same-box A/B runs (bench.py --lib build/ab_<rev>/liblang2seg_hip.so):
#   tools/ab_build.sh <git rev>        (the Python host stays the working tree's; only the kernels differ)
set -e
REV=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build/ab_$REV
rm -rf "$D"; mkdir -p "$D"
git -C "$R" archive "$REV" lang2seg_amd/csrc include | tar -x -C "$D"
cd "$D"
ls lang2seg_amd/csrc/*.hip | xargs -P 6 -I{} sh -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c {} -o $(basename {} .hip).o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblang2seg_hip.so *.o
rm -f *.o
ls -la "$D/liblang2seg_hip.so"
