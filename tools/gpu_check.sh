#!/bin/bash
# the GPU suite without the long full-size runs, then the three configurations at the driver's protocol (no CPU leg)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/check
mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -q -m gpu --durations=5 -k "not 32_steps and not 30_steps" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -12 $OUT/tests.log
for c in 2 3 4; do
  timeout 600 python bench.py --steps 20 --warmup 5 --config $c --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_cfg$c.json').read().strip().splitlines()[-1])
    print('cfg$c', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], d['roofline']['kernel'], 'frac %.3f'%d['roofline']['frac'], 'us %.1f'%d['roofline']['kernel_us'], 'grid_ms %.3f'%d['config']['grid_ms'], 'solver_ms %.3f'%d['config']['solver_ms'])
except Exception as e:
    print('cfg$c failed', e); print(open('$OUT/bench_cfg$c.err').read()[-1500:])
PY
done
