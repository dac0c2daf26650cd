#!/bin/bash
# same-box A/B over environment settings: tools/r4_env_matrix.sh <rounds> "VAR=val ..." "VAR=val ..."   ("-" = nothing set)
N=$1; shift
for i in $(seq $N); do
  for A in "$@"; do
    E="$A"; [ "$A" = "-" ] && E=""
    env $E timeout 300 python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-36s %.2f img/s  %.3f ms' % ('$A', d['value'], d['ms_per_step']))"
  done
done
