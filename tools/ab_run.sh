#!/bin/bash
# Same-box A/B: tools/ab_run.sh "<tools/ab.py flags of arm A, '' = product>" "<flags of arm B>" [rounds] [extra bench.py args]
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in "$A" "$B"; do
    timeout 300 python tools/ab.py $L -- --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-40s %.2f img/s  %.3f ms' % ('${L:-product}', d['value'], d['ms_per_step']))"
  done
done
