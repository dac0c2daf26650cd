#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_train_step_gpu.py -x -q -k "deferred_heads or encoder_is_deterministic or bit_reproducible" > gpurun_out/r4_fourth_tests.log 2>&1
tail -4 gpurun_out/r4_fourth_tests.log
for i in 1 2 3; do
  for L in "" tools/_lib_packed.so; do
    if [ -n "$L" ]; then X="--lib $L"; else X=""; fi
    timeout 300 python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 $X 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-24s %.2f img/s  %.3f ms' % ('${L:-in-tree (no packed)}', d['value'], d['ms_per_step']))"
  done
done | tee gpurun_out/r4_packed_ab.txt
