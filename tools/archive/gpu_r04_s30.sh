#!/bin/bash
# round 4, session 30: end-of-step kernel folds the bounding box and restores the solve control blocks: full suite, sha, bench
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s30
mkdir -p $O
timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,4 --reps 20 2>&1 | grep -E "^AB lib" > $O/probe.log
timeout 200 python tools/ab_probe.py --config 3 --steps 25 --kernels 2 --reps 20 2>&1 | grep -E "^AB lib" >> $O/probe.log
cat $O/probe.log | cut -c1-230
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 4 $O/tests.log
for rep in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_$rep.json
done
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_cfg3.json
python - <<'PY'
import json
for f in ('bench_1','bench_2','bench_cfg3'):
    d=json.load(open('gpurun_out/s30/%s.json'%f)); print(f, round(d['ms_per_step'],4), [round(x,3) for x in d['per_step_ms'][:6]], d['config']['warmup_grid_ms'], d['regimes']['settled'])
PY
