#!/bin/bash
# Same positional interface as the reference's experiments/scripts/eval_vgg.sh: GPU_ID DATASET SPLITBY OUTPUT_POSTFIX MODEL_ITER
GPU_ID=$1
DATASET=$2
SPLITBY=$3
OUTPUT_POSTFIX=$4
MODEL_ITER=$5
NET="vgg16"
ID="mrcn_cmr_with_st"
ANCHORS="[4,8,16,32]"
RATIOS="[0.5,1,2]"
case ${DATASET} in
  refcocog) SPLITS="val test" ;;
  *) SPLITS="val testA testB" ;;
esac
for SPLIT in ${SPLITS}; do
  HIP_VISIBLE_DEVICES=${GPU_ID} python ./tools/eval_vgg.py --dataset ${DATASET} --splitBy ${SPLITBY} --output_postfix ${OUTPUT_POSTFIX} \
    --model_iter ${MODEL_ITER} --split ${SPLIT} --id ${ID} --cfg experiments/cfgs/${NET}.yml \
    --set ANCHOR_SCALES ${ANCHORS} ANCHOR_RATIOS ${RATIOS}
done
