cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/lstm_trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lstm_trace -o t -- python3 $R/tools/vgg_tape_diff.py vgg > $R/gpurun_out/lstm_trace/run.log 2>&1
cd $R
grep "====\|differ," gpurun_out/lstm_trace/run.log
python3 tools/lstm_overlap_from_trace.py gpurun_out/lstm_trace/t_kernel_trace.csv
rm -f gpurun_out/lstm_trace/t_kernel_trace.csv
