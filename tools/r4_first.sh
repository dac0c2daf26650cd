#!/bin/bash
# round 4, first GPU call: deferred heads stage - parity first, then same-box A/B and the two timelines
mkdir -p gpurun_out
python -m pytest tests/test_train_step_gpu.py -x -q -k "deferred_heads or pipelined_tape or bit_reproducible or early_partial or snapshot_and_resume or dp_tape or no_gradient_lands" > gpurun_out/r4_first_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_first_tests.log
tail -5 gpurun_out/r4_first_tests.log
for i in 1 2 3; do
  for D in 0 1; do
    timeout 300 python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 --defer $D 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('defer=$D %.2f img/s  %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done | tee gpurun_out/r4_first_ab.txt
python tools/step_timeline.py --defer 0 > gpurun_out/r4_timeline_defer0.txt 2>&1
python tools/step_timeline.py --defer 1 > gpurun_out/r4_timeline_defer1.txt 2>&1
cat gpurun_out/r4_timeline_defer1.txt
