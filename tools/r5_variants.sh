#!/bin/bash
# every BASELINE.json GPU config at its stated size through bench.py (--variant); one JSON line each -> gpurun_out/r5/bench_variants.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
: > $O/bench_variants.json
for v in cycle spatial cycle_response vgg baseline; do
  timeout 600 python bench.py --variant $v --no-cpu-baseline --mixed-shapes 0 --steps 50 --warmup 10 2>$O/variant_$v.err | grep '^{' | tail -1 >> $O/bench_variants.json || echo "FAILED $v"
  tail -3 $O/variant_$v.err | cut -c1-300
done
python - <<'PY'
import json
for l in open('gpurun_out/r5/bench_variants.json'):
    d = json.loads(l); r = d.get('roofline') or {}
    g = r.get('groups') or {}
    top = sorted(g.items(), key=lambda kv: -kv[1]['ms_per_step'])[:4]
    print('%-15s %7.2f img/s %6.3f ms  step frac %.3f  dropin %s | %s' % (d['config']['variant'], d['value'], d['ms_per_step'], d['step_frac_of_bf16_peak'],
          ('%.1f' % d['dropin_train_step_value']) if 'dropin_train_step_value' in d else '-', '; '.join('%s %.2fms %.3f' % (k, v['ms_per_step'], v['frac']) for k, v in top)))
PY
