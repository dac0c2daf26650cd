#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_run.sh <lib A or ''> <lib B or ''> [rounds] [extra bench args]; '' = the in-tree build
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in "$A" "$B"; do
    if [ -n "$L" ]; then X="--lib $L"; else X=""; fi
    timeout 300 python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 $X "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-40s %.2f img/s  %.3f ms' % ('${L:-in-tree}', d['value'], d['ms_per_step']))"
  done
done
