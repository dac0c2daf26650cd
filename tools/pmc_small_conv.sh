# PMC study of one small-M convolution launch (default: layer3 3x3, M=2394, N=256, K=2304, 64x64 tile): SQ issue / wait / LDS / VMEM
# counters and L2 hit rates, one rocprofv3 --pmc pass per counter group (counters only together with --kernel-trace).
# usage: tools/pmc_small_conv.sh [n H W Cin Cout k s p]   -> gpurun_out/pmc_small/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${@:-1 38 63 256 256 3 1 1}
OUT=$R/gpurun_out/pmc_small
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CYCLES GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/one_conv.py $ARGS fwd 0 > $OUT/p$i.log 2>&1
done
cd $R; python3 - <<'PY' > gpurun_out/pmc_small/summary.txt
import glob, csv, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
for f in glob.glob('gpurun_out/pmc_small/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for f in glob.glob('gpurun_out/pmc_small/p*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'igemm' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('kernel duration us: median %.2f (n=%d)' % (sorted(dur)[len(dur)//2], len(dur)))
for k in sorted(agg): print('  %-34s %16.0f (per launch, n=%d)' % (k, agg[k]/n[k], n[k]))
PY
cat gpurun_out/pmc_small/summary.txt
