# HBM-side traffic of the dominant launch (layer4@RoIs 3x3 igemm, 224x128 tile): separate --pmc passes, as
# MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 16-B/lane loads at half their bytes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_$C -o p -- python3 $R/tools/one_conv.py 256 7 7 512 512 3 1 1 fwd 0 > /dev/null 2>&1
done
cd $R; python3 - <<'PY'
import glob, csv, json
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for f in glob.glob('gpurun_out/pmc_traffic_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'igemm_sp_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c:
                vals.append(float(r['Counter_Value']))
    out[c] = vals
f = sum(out['FETCH_SIZE'][2:]) / max(len(out['FETCH_SIZE'][2:]), 1); w = sum(out['WRITE_SIZE'][2:]) / max(len(out['WRITE_SIZE'][2:]), 1)
res = dict(kernel='igemm_sp_kernel<bf16,224,128>, layer4@RoIs conv3x3 forward (M=12544,N=512,K=4608), bias+ReLU epilogue',
           launches=len(out['FETCH_SIZE']), FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w,
           read_bytes_corrected=2 * f * 1024, write_bytes=w * 1024, traffic_bytes=2 * f * 1024 + w * 1024,
           algorithmic_bytes=12544 * 512 * 2 * 2 + 512 * 4608 * 2,
           note='read side = 2 x FETCH_SIZE (16-byte-per-lane buffer loads are tallied at half their bytes on gfx950); write side = WRITE_SIZE as is (16-byte bf16x8 stores from the LDS-staged epilogue)')
print(json.dumps(res, indent=1))
json.dump(res, open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
PY
