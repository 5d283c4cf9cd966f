#!/bin/bash
# bench.py at the driver's protocol for the three single-GPU configurations (config 2 with its CPU leg)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-cfgs}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
timeout 900 python bench.py --steps 20 --warmup 5 --config 3 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
timeout 900 python bench.py --steps 20 --warmup 5 --config 4 --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
for c in 2 3 4; do python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_cfg$c.json').read().strip().splitlines()[-1])
    print('cfg$c', d['metric'], '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], d['roofline']['kernel'], 'frac %.3f'%d['roofline']['frac'], 'us %.1f'%d['roofline']['kernel_us'], 'K %.1f'%d['roofline']['mean_contacts'], 'iters', d['config']['mean_divergence_iters'], d['config']['mean_pressure_iters'])
    print('   regimes', d['regimes'])
    if d['cpu_baseline']: print('   cpu', d['cpu_baseline'])
except Exception as e:
    print('cfg$c failed', e); print(open('$OUT/bench_cfg$c.err').read()[-1500:])
PY
done
