#!/bin/bash
# tools/ab_variant.sh <name> <sed expression>: builds the working tree's library with one sed edit applied to conv_igemm.hip into build/ab_<name>/
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/build/ab_$1
rm -rf "$D"; mkdir -p "$D/lang2seg_amd/csrc" "$D/include"
cp $R/lang2seg_amd/csrc/*.h* "$D/lang2seg_amd/csrc/"; cp $R/include/*.h "$D/include/"
sed -i "$2" "$D/lang2seg_amd/csrc/conv_igemm.hip"
if cmp -s "$D/lang2seg_amd/csrc/conv_igemm.hip" "$R/lang2seg_amd/csrc/conv_igemm.hip"; then echo "sed changed nothing"; exit 1; fi
cd "$D"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c lang2seg_amd/csrc/conv_igemm.hip -o conv_igemm.o
for f in $R/lang2seg_amd/lib/obj/*.o; do [ "$(basename $f)" = conv_igemm.o ] || cp $f .; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblang2seg_hip.so *.o
rm -rf *.o lang2seg_amd include; ls -la "$D"
