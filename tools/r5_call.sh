#!/bin/bash
# generic round-5 GPU call: tests named in $1 (pytest -k expression), then same-box A/B lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
K="$1"; shift
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_step_gpu.py -x -q -m gpu -k "$K" 2>&1 | tail -15
for arm in "$@"; do
  timeout 300 python tools/ab.py $arm -- --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 2>$O/ab_err.txt | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
  d=json.loads(l); print('%-40s %.2f img/s  %.3f ms' % ('${arm:-product}', d['value'], d['ms_per_step']))
except Exception as e:
  print('FAILED arm ${arm}:', l[:200]); print(open('$O/ab_err.txt').read()[-1500:])"
done
