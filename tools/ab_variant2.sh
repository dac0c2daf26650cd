#!/bin/bash
# tools/ab_variant2.sh <name> <file under lang2seg_amd/csrc> <sed expression>: like ab_variant.sh for any source file
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/build/ab_$1
rm -rf "$D"; mkdir -p "$D/lang2seg_amd/csrc" "$D/include"
cp $R/lang2seg_amd/csrc/*.h* "$D/lang2seg_amd/csrc/"; cp $R/include/*.h "$D/include/"
sed -i "$3" "$D/lang2seg_amd/csrc/$2"
if cmp -s "$D/lang2seg_amd/csrc/$2" "$R/lang2seg_amd/csrc/$2"; then echo "sed changed nothing"; exit 1; fi
cd "$D"
o=$(basename $2 .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c lang2seg_amd/csrc/$2 -o $o
for f in $R/lang2seg_amd/lib/obj/*.o; do [ "$(basename $f)" = $o ] || cp $f .; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblang2seg_hip.so *.o
rm -rf *.o lang2seg_amd include; ls -la "$D"
