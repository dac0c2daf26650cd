#!/bin/bash
mkdir -p gpurun_out
python tools/vgg_determinism.py vgg > gpurun_out/r4_vgg_determinism.txt 2>&1
tail -30 gpurun_out/r4_vgg_determinism.txt
run() { timeout 300 python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$* : %.2f img/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2; do
  for D in 0 1; do
    for C in 0 64 104 128 192 256; do
      run --defer $D --wgrad-cap $C
    done
  done
done | tee gpurun_out/r4_cap_ab.txt
python tools/step_timeline.py --defer 1 --wgrad-cap 104 > gpurun_out/r4_timeline_defer1_cap104.txt 2>&1
python tools/step_timeline.py --defer 0 --wgrad-cap 104 > gpurun_out/r4_timeline_defer0_cap104.txt 2>&1
