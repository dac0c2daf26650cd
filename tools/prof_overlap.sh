cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ov -o s -- python3 $R/tools/overlap_test.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/prof_ov/s_kernel_trace.csv')))
c=collections.Counter((r['Queue_Id'], r['Stream_Id'], r['Kernel_Name'][:40]) for r in rows)
for k,v in sorted(c.items()): print(k,v)
PY
