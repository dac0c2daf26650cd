#!/bin/bash
# Same positional interface as the reference's experiments/scripts/train_vgg.sh:
#   GPU_ID DATASET SPLITBY OUTPUT_POSTFIX
# GPU_ID may be a comma list ("0,1,2,3,4,5,6,7") -> one process per GPU over RCCL (torchrun).
GPU_ID=$1
DATASET=$2
SPLITBY=$3
OUTPUT_POSTFIX=$4

IMDB="coco_minus_refer"
ITERS=1250000
TAG="notime"
NET="vgg16"
ID="mrcn_cmr_with_st"
STEPSIZE="[360000]"
ANCHORS="[4,8,16,32]"
RATIOS="[0.5,1,2]"

NGPU=$(echo ${GPU_ID} | tr ',' '\n' | wc -l)
ARGS="--imdb_name ${IMDB} --net_name ${NET} --iters ${ITERS} --tag ${TAG} --dataset ${DATASET} --splitBy ${SPLITBY} \
  --output_postfix ${OUTPUT_POSTFIX} \
  --max_iters ${MAX_ITERS:-600000} --with_st 1 --id ${ID} \
  --cfg experiments/cfgs/${NET}.yml --set ANCHOR_SCALES ${ANCHORS} ANCHOR_RATIOS ${RATIOS} TRAIN.STEPSIZE ${STEPSIZE}"
if [ "${NGPU}" -gt 1 ]; then
  HIP_VISIBLE_DEVICES=${GPU_ID} python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NGPU} --master-addr 127.0.0.1 \
    --master-port ${MASTER_PORT:-29511} ./tools/train_vgg.py ${ARGS}
else
  HIP_VISIBLE_DEVICES=${GPU_ID} python ./tools/train_vgg.py ${ARGS}
fi
