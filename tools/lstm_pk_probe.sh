#!/bin/bash
# builds of the library whose lang.hip is compiled WITH packed fp32 VALU ops (every other file as the product: without), in four
# forms of lstm_step_fwd_kernel's hand-off from the v_pk_fma_f32 loop to the DPP reduction (L2S_LSTM_PROBE in csrc/lang.hip):
#   pk0 = as the compiler emits it, pk1 = + 16 wait states, pk2 = + a plain v_mov of every accumulator, pk3 = + an empty asm with the same constraints
# Run here (cross-compiles); tools/lstm_pk_fuzz.sh runs tools/vgg_corun_fuzz.py on each on the GPU box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/_lstm_pk; mkdir -p $D
OBJ=$R/lang2seg_amd/lib/obj
python $R/__graft_entry__.py > /dev/null
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
for v in ${VARIANTS:-0 1 2 3 4 5 7}; do
  DEF=""; [ $v != 0 ] && DEF="-DL2S_LSTM_PROBE=$v"
  FLV="$FL"; [ $v = 9 ] && FLV="$FL -Xclang -target-feature -Xclang -packed-fp32-ops"      # probe 9: WITHOUT packed ops (the product's flags)
  /opt/rocm/bin/hipcc $FLV $DEF -c $R/lang2seg_amd/csrc/lang.hip -o $D/lang_pk$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libpk$v.so $D/lang_pk$v.o $(ls $OBJ/*.o | grep -v '/lang.o$')
  rm -f $D/lang_pk$v.o
  /opt/rocm/bin/hipcc $FLV $DEF -S --cuda-device-only -o $D/lang_pk$v.s $R/lang2seg_amd/csrc/lang.hip 2>/dev/null
done
ls -la $D
