#!/bin/bash
mkdir -p gpurun_out
python tools/vgg_tape_diff.py vgg > gpurun_out/r4_vgg_tape_diff.txt 2>&1
bash tools/prof_step.sh > gpurun_out/r4_prof_step.log 2>&1
head -45 gpurun_out/prof_step/summary.csv
