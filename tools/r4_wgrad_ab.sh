#!/bin/bash
# same-box A/B of the LDS-DMA filter-row weight-gradient tile inside the step (bench.py --wgrad-row3-dma 0/1)
for i in 1 2 3; do
  for D in 0 1; do
    timeout 300 python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 --wgrad-row3-dma $D 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('row3-dma=$D %.2f img/s  %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
