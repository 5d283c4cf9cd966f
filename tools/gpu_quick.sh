#!/bin/bash
# quick GPU check: the GPU suite without the long full-size runs, then the bench line at both protocols
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-quick}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu --durations=5 -k "not 32_steps and not 30_steps" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -15 $OUT/tests.log
for P in "20 5" "50 5"; do
  set -- $P
  timeout 600 python bench.py --steps $1 --warmup $2 --no-cpu-baseline > $OUT/bench_$2_$1.json 2> $OUT/bench_$2_$1.err
  python -c "
import sys,json; d=json.loads(open('$OUT/bench_$2_$1.json').read().strip().splitlines()[-1]); print('$2+$1:', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'kernel_us %.1f'%d['roofline']['kernel_us'], 'grid_ms %.3f'%d['config']['grid_ms'], 'solver_ms %.3f'%d['config']['solver_ms'])"
done
