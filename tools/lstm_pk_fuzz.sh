#!/bin/bash
# on the GPU box: the encoder beside the VGG backbone, replayed REPS times per library variant of tools/lstm_pk_probe.sh (and the product build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
REPS=${1:-1500}
{
echo "# tools/lstm_pk_fuzz.sh: tools/vgg_corun_fuzz.py $REPS replays, first 8 / all layers of the VGG backbone beside the encoder"
if [ -z "$NO_PRODUCT" ]; then
echo "## product build (no packed fp32 ops anywhere)"
timeout 600 python tools/vgg_corun_fuzz.py $REPS 8,99 2>/dev/null | grep "replays gave"
fi
for v in ${VARIANTS:-0 1 2 3 4 5 7}; do
  echo "## lang.hip with packed fp32 ops, L2S_LSTM_PROBE=$v"
  L2S_FUZZ_LIB=tools/_lstm_pk/libpk$v.so timeout 600 python tools/vgg_corun_fuzz.py $REPS 8,99 2>/dev/null | grep "replays gave"
done
} | tee $O/lstm_pk_fuzz${TAG}.txt
