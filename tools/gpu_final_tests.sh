#!/bin/bash
# end-of-round session, part 1: the whole GPU suite (with the long full-size runs), then the bench line at the driver's
# protocol (5 + 20, with the CPU leg) and at the survey's (5 + 50)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_final
mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -14 $OUT/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_5_20.json 2> $OUT/bench_5_20.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_5_50.json 2> $OUT/bench_5_50.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
for f in bench_5_20 bench_5_50; do python - <<PY
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'us %.1f'%d['roofline']['kernel_us'], d['regimes'], d['cpu_baseline'])
PY
done
