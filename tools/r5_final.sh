#!/bin/bash
# round-5 final measurements in one GPU call; everything lands in gpurun_out/r5_final/ (copied to profiles/r05_* by hand afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_final; mkdir -p $O
cd $R
bash tools/prof_step.sh > $O/prof_step.log 2>&1
cp gpurun_out/prof_step/summary.csv $O/step_kernel_stats.csv; cp gpurun_out/prof_step/summary.meta.json $O/step_kernel_stats.meta.json
cp gpurun_out/prof_step/s_kernel_stats.csv $O/rocprofv3_kernel_stats.csv
bash tools/pmc_traffic.sh dominant igemm_p3_kernel 1 38 63 256 256 3 1 1 dgrad > $O/pmc_traffic_dominant.log 2>&1; cp gpurun_out/pmc_traffic_dominant.json $O/ 2>/dev/null
bash tools/pmc_traffic.sh best igemm_dma_kernel 256 7 7 512 512 3 1 1 fwd > $O/pmc_traffic_best.log 2>&1; cp gpurun_out/pmc_traffic_best.json $O/ 2>/dev/null
python tools/wgrad_group_bench.py > $O/wgrad_group_bench.txt 2>&1
python tools/conv_bench.py > $O/conv_bench.txt 2>&1
python tools/step_timeline.py > $O/step_timeline.txt 2>&1
python bench.py > $O/bench_final.log 2>&1; grep '^{' $O/bench_final.log | tail -1 > $O/bench_final.json
bash tools/r5_variants.sh > $O/variants.log 2>&1; cp gpurun_out/r5/bench_variants.json $O/bench_variants.json
bash tools/r5_dp_matrix.sh > /dev/null 2>&1; cp gpurun_out/r5/dp_onerank_matrix.txt $O/
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/cap_allgather_probe.hip -o /tmp/cap_ag_probe 2>/dev/null && timeout 120 /tmp/cap_ag_probe > $O/cap_allgather_probe.txt 2>&1
ls -la $O
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5_final/bench_final.json'))
r = d['roofline']
print('value %.2f img/s  %.3f ms  sync %.1f  dropin %.1f' % (d['value'], d['ms_per_step'], d.get('sync_train_step_value', 0), d.get('dropin_train_step_value', 0)))
print('roofline frac %.4f (%s)  best %.3f  stack3x3 %.3f  kernel_time %s' % (r['frac'], r['kernel'][:60], r['best']['frac'], r['stack3x3']['frac'], r.get('kernel_time_ms_per_step')))
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
