#!/bin/bash
# one rank through RCCL: the fixed costs of every data-parallel form (nothing is saved on the wire at one rank) -> gpurun_out/r5/dp_onerank_matrix.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
B="--no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10"
run() { timeout 400 python tools/ab.py $1 -- $B $2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-58s %7.2f img/s  %.3f ms' % ('$3', d['value'], d['ms_per_step']))"; }
{
echo "# tools/r5_dp_matrix.sh: bench.py at ONE rank through RCCL (tools/ab.py --force-dp 1), same box, 100 steps each"
for rep in 1 2; do
run "" "" "no reducer (the single-process step)"
run "--force-dp 1" "--dp-wire fp32 --dp-algo allreduce --dp-shard-update 0" "fp32 buckets, all-reduce"
run "--force-dp 1" "--dp-wire bf16 --dp-algo rs_ag --dp-shard-update 0" "bf16 buckets, reduce-scatter + all-gather"
run "--force-dp 1" "--dp-wire bf16 --dp-algo rs_ag --dp-shard-update 1" "bf16 buckets, rs_ag, sharded update (default for N > 1)"
done
} | tee $O/dp_onerank_matrix.txt
