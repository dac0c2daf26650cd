#!/bin/bash
# round-4 final measurements in one GPU call; everything lands in gpurun_out/r4_final/ (copied to profiles/ by hand afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_final; mkdir -p $O
cd $R
bash tools/prof_step.sh > $O/prof_step.log 2>&1
cp gpurun_out/prof_step/summary.csv $O/step_kernel_stats.csv; cp gpurun_out/prof_step/summary.meta.json $O/step_kernel_stats.meta.json
cp gpurun_out/prof_step/s_kernel_stats.csv $O/rocprofv3_kernel_stats.csv
bash tools/pmc_traffic.sh dominant igemm_p3_kernel 1 38 63 256 256 3 1 1 dgrad > $O/pmc_traffic_dominant.log 2>&1; cp gpurun_out/pmc_traffic_dominant.json $O/ 2>/dev/null
bash tools/pmc_traffic.sh best igemm_dma_kernel 256 7 7 512 512 3 1 1 fwd > $O/pmc_traffic_best.log 2>&1; cp gpurun_out/pmc_traffic_best.json $O/ 2>/dev/null
WG_ARGS="--only layer4" bash tools/pmc_wgrad.sh > $O/pmc_wgrad.txt 2>&1
python tools/wgrad_group_bench.py > $O/wgrad_group_bench.txt 2>&1
python tools/conv_bench.py > $O/conv_bench.txt 2>&1
python tools/gemm_yardstick.py > $O/gemm_yardstick.txt 2>&1
python tools/step_timeline.py > $O/step_timeline.txt 2>&1
python bench.py > $O/bench_final.log 2>&1; grep '^{' $O/bench_final.log | tail -1 > $O/bench_final.json
for w in fp32:allreduce:0 bf16:rs_ag:0 bf16:rs_ag:1; do
  IFS=: read wire algo sh <<< "$w"
  python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 --force-dp 1 --dp-wire $wire --dp-algo $algo --dp-shard-update $sh 2>/dev/null | grep '^{' | tail -1 > $O/dp_onerank_${wire}_${algo}_shard${sh}.json
done
python bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 100 --warmup 10 2>/dev/null | grep '^{' | tail -1 > $O/dp_onerank_none.json
ls -la $O
tail -c 600 $O/bench_final.json
