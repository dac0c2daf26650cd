#!/bin/bash
# one-rank runs of the data-parallel step (RCCL executed, world 1), same box
for i in 1 2; do
for A in "" "--force-dp 1 --dp-wire fp32 --dp-algo allreduce --dp-bucket-update 0" "--force-dp 1 --dp-wire fp32 --dp-algo allreduce --dp-bucket-update 1" "--force-dp 1 --dp-wire bf16 --dp-algo rs_ag --dp-shard-update 0 --dp-bucket-update 1" "--force-dp 1 --dp-wire bf16 --dp-algo rs_ag --dp-shard-update 1"; do
  timeout 300 python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 $A 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-44s %.2f img/s  %.3f ms  exposed wait %s ms' % (d['config']['parallelism'], d['value'], d['ms_per_step'], (d.get('dp') or {}).get('exposed_wait_ms')))"
done
done
