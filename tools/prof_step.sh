cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_step; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -o s -- python3 $R/bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 10 --warmup 5 > $R/gpurun_out/prof_step/bench.log 2>&1
cd $R
tail -1 gpurun_out/prof_step/bench.log | cut -c1-200
python3 tools/trace_summary.py gpurun_out/prof_step/s_kernel_trace.csv 8 gpurun_out/prof_step/summary.csv
python3 -c "import json, bench; json.dump({'src_hash': bench.src_hash(), 'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --extras 0 --mixed-shapes 0 --steps 10 --warmup 5'}, open('gpurun_out/prof_step/summary.meta.json', 'w'))"
