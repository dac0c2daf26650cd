#!/bin/bash
# round-6 final measurements in one GPU call; everything lands in gpurun_out/r6_final/ (copied to profiles/r06_* by hand afterwards).
# Every command runs under its own timeout: a hung profiler run must not take the whole call (and its GPU minutes) with it.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_final; mkdir -p $O
cd $R
timeout 900 bash tools/prof_step.sh > $O/prof_step.log 2>&1
cp gpurun_out/prof_step/summary.csv $O/step_kernel_stats.csv; cp gpurun_out/prof_step/summary.meta.json $O/step_kernel_stats.meta.json
cp gpurun_out/prof_step/s_kernel_stats.csv $O/rocprofv3_kernel_stats.csv
# (bench.py reports roofline.kernel_time_ms_per_step / roofline.rocprof from the committed table while its source hash matches: put this run's table there first)
cp $O/step_kernel_stats.csv profiles/r06_step_kernel_stats.csv; cp $O/step_kernel_stats.meta.json profiles/r06_step_kernel_stats.meta.json
timeout 400 bash tools/pmc_traffic.sh dominant igemm_p3_kernel 1 38 63 256 256 3 1 1 dgrad > $O/pmc_traffic_dominant.log 2>&1; cp gpurun_out/pmc_traffic_dominant.json $O/ 2>/dev/null
timeout 400 bash tools/pmc_traffic.sh best igemm_dma_kernel 256 7 7 512 512 3 1 1 fwd > $O/pmc_traffic_best.log 2>&1; cp gpurun_out/pmc_traffic_best.json $O/ 2>/dev/null
timeout 300 python tools/wgrad_group_bench.py > $O/wgrad_group_bench.txt 2>&1
timeout 300 python tools/conv_bench.py > $O/conv_bench.txt 2>&1
timeout 300 python tools/step_timeline.py > $O/step_timeline.txt 2>&1
# the proposal chain (round 5: staged column-form NMS) and the diagnostics of DESIGN.md 4.7h-k
{ timeout 200 python tools/nms_bench.py; timeout 200 python tools/proposal_depth.py; } > $O/nms_bench.txt 2>&1
timeout 200 python tools/build_tools_lib.py > /dev/null 2>&1
{ timeout 200 python tools/nms_cycles.py tests/golden/nms_deep_boxes.npy; } > $O/nms_cycles.txt 2>&1
{ timeout 200 python tools/backbone_alone.py; timeout 200 python tools/cold_weights_bench.py; timeout 200 python tools/host_time.py; timeout 300 python tools/clock_probe.py; } > $O/step_diagnostics.txt 2>&1
timeout 900 python bench.py > $O/bench_final.log 2>&1; grep '^{' $O/bench_final.log | tail -1 > $O/bench_final.json
timeout 2400 bash tools/r5_variants.sh > $O/variants.log 2>&1; cp gpurun_out/r5/bench_variants.json $O/bench_variants.json
timeout 2400 bash tools/r6_dp_matrix.sh > /dev/null 2>&1; cp gpurun_out/r6/dp_onerank_matrix.txt $O/
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/cap_allgather_probe.hip -o /tmp/cap_ag_probe 2>/dev/null && timeout 120 /tmp/cap_ag_probe > $O/cap_allgather_probe.txt 2>&1
ls -la $O
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6_final/bench_final.json'))
r = d['roofline']
print('value %.2f img/s  %.3f ms  sync %.1f  dropin %.1f' % (d['value'], d['ms_per_step'], d.get('sync_train_step_value', 0), d.get('dropin_train_step_value', 0)))
print('roofline frac %.4f (%s)  best %.3f  stack3x3 %.3f  kernel_time %s' % (r['frac'], r['kernel'][:60], r['best']['frac'], r['stack3x3']['frac'], r.get('kernel_time_ms_per_step')))
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
