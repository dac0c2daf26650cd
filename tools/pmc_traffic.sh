# HBM-side (fabric) traffic of one convolution launch: separate --pmc passes, as MI355X_MICROARCH.md section HBM prescribes
# (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 16-B/lane loads at half their bytes on gfx950).
# usage: pmc_traffic.sh <tag> <kernel-name substring> <n H W Cin Cout k s p> <mode: fwd | dgrad> [algo]
#   pmc_traffic.sh dominant igemm_ws64_kernel 1 38 63 256 256 3 1 1 dgrad      (the time-dominant group: layer3 data gradients)
#   pmc_traffic.sh best igemm_dma_kernel 256 7 7 512 512 3 1 1 fwd             (layer4@RoIs 3x3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; KN=$2; shift 2
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_traffic_${TAG}_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_${TAG}_$C -o p -- python3 $R/tools/one_conv.py "$@" > /dev/null 2>&1
done
cd $R; python3 - "$TAG" "$KN" "$@" <<'PY'
import glob, csv, json, sys
tag, kn = sys.argv[1], sys.argv[2]
n, H, W, Cin, Cout, k, s, p = [int(v) for v in sys.argv[3:11]]
mode = sys.argv[11] if len(sys.argv) > 11 else 'fwd'
out = {}
name = None
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = []
    for f in glob.glob('gpurun_out/pmc_traffic_%s_%s/**/*counter_collection.csv' % (tag, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if kn in r['Kernel_Name'] and r['Counter_Name'] == c:
                vals.append(float(r['Counter_Value'])); name = r['Kernel_Name']
    out[c] = vals
f = sum(out['FETCH_SIZE'][2:]) / max(len(out['FETCH_SIZE'][2:]), 1); w = sum(out['WRITE_SIZE'][2:]) / max(len(out['WRITE_SIZE'][2:]), 1)
OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
M = n * OH * OW
alg = n * H * W * Cin * 2 + Cout * k * k * Cin * 2 + M * Cout * 2 * (2 if mode == 'dgrad' else 1)     # input + weights + output (+ the ReLU-mask operand)
res = dict(kernel=name, shape=dict(n_img=n, H=H, W=W, Cin=Cin, Cout=Cout, k=k, stride=s, pad=p, form=mode), launches=len(out['FETCH_SIZE']),
           FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, read_bytes_corrected=2 * f * 1024, write_bytes=w * 1024, traffic_bytes=2 * f * 1024 + w * 1024,
           algorithmic_bytes=alg,
           note='read side = 2 x FETCH_SIZE (16-byte-per-lane buffer loads are tallied at half their bytes on gfx950); write side = WRITE_SIZE as is (16-byte bf16x8 stores from the LDS-staged epilogue)')
print(json.dumps(res, indent=1))
json.dump(res, open('gpurun_out/pmc_traffic_%s.json' % tag, 'w'), indent=1)
PY
