#!/bin/bash
# round 4, session 33: IISPH d_ii inside the density pass: tests, config 3 A/B (SALVA_HIP_NO_FUSED_DIV=1 = separate k_iisph_dii)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s33
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 4 $O/tests.log
for rep in 1 2; do
  timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_cfg3_fused_$rep.json
  SALVA_HIP_NO_FUSED_DIV=1 timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_cfg3_separate_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s33/bench_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],4), d['config']['mean_pressure_iters'], [tuple(x) for x in d['iters'][-3:]])
PY
