# SQ counters of the grouped weight-gradient launches (tools/wgrad_group_bench.py), separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wg_$i -o p -- python3 $R/tools/wgrad_group_bench.py $WG_ARGS > /dev/null 2>&1
done
cd $R; python3 - <<'PY'
import glob, csv, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc_wg_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_grouped_kernel" in r["Kernel_Name"] or "wgrad_row3" in r["Kernel_Name"]:
            k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void (anonymous namespace)::', ''))
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]): print('  %-28s %14.0f (per launch, n=%d)' % (c, agg[k][c] / n[k][c], n[k][c]))
PY
rm -rf gpurun_out/pmc_wg_*
