#!/bin/bash
# one rank through RCCL: the fixed costs of the data-parallel forms (nothing is saved on the wire at one rank) -> gpurun_out/r6/dp_onerank_matrix.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
B="--no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10"
run() { timeout 400 python tools/ab.py $1 -- $B $2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-78s %7.2f img/s  %.3f ms' % ('$3', d['value'], d['ms_per_step']))"; }
{
echo "# tools/r6_dp_matrix.sh: bench.py at ONE rank through RCCL (tools/ab.py --force-dp 1), same box, 100 steps each"
for rep in 1 2; do
run "" "" "no reducer (the single-process step)"
run "--force-dp 1" "--dp-wire fp32 --dp-algo allreduce --dp-shard-update 0" "fp32 buckets, all-reduce"
run "--force-dp 1 --dp-g16 0 --dp-skip-stages ," "--dp-wire bf16 --dp-algo rs_ag --dp-shard-update 1" "bf16 rs_ag sharded, 7 buckets, cast back (round 5 form + master/shadow plan)"
run "--force-dp 1 --dp-skip-stages ," "--dp-wire bf16 --dp-algo rs_ag --dp-shard-update 1" "bf16 rs_ag sharded, 7 buckets, update reads the bf16 shard"
run "--force-dp 1 --dp-skip-stages caption,layer3:16" "--dp-wire bf16 --dp-algo rs_ag --dp-shard-update 1" "bf16 rs_ag sharded, 5 buckets (caption+heads, language, layer3 x2, layer2)"
run "--force-dp 1 --dp-skip-stages caption,language,layer3:16,layer3:8" "--dp-wire bf16 --dp-algo rs_ag --dp-shard-update 1" "bf16 rs_ag sharded, 3 buckets (heads, layer3, layer2)"
done
} | tee $O/dp_onerank_matrix.txt
