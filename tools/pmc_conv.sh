cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in 128 256; do
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${T}_$i -o p -- python3 $R/tools/one_conv.py 256 7 7 512 512 3 1 1 fwd $T > /dev/null 2>&1
done
done
cd $R; python3 - <<'PY'
import glob, csv, collections
for T in (128,256):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob('gpurun_out/pmc_%d_*/**/*counter_collection.csv' % T, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'igemm' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print('tile', T)
    for k in sorted(agg): print('  %-28s %14.0f (per launch, n=%d)' % (k, agg[k]/n[k], n[k]))
PY
