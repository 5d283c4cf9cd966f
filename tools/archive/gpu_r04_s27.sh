#!/bin/bash
# round 4, session 27: counting sort by cell against the radix sort it replaces
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s27
mkdir -p $O
for steps in 25 60; do
  timeout 200 python tools/ab_probe.py --steps $steps --kernels 0,4 --reps 20 2>&1 | grep -E "^AB lib" | sed "s/^/counting /" >> $O/sort.log
  SALVA_HIP_RADIX_SORT=1 timeout 200 python tools/ab_probe.py --steps $steps --kernels 0,4 --reps 20 2>&1 | grep -E "^AB lib" | sed "s/^/radix    /" >> $O/sort.log
done
cat $O/sort.log | cut -c1-230
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_fuzz_gpu.py tests/test_dynamic_sampling_gpu.py tests/test_speculation_gpu.py -q -m gpu -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 6 $O/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench.json
SALVA_HIP_RADIX_SORT=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_radix.json
timeout 300 python bench.py --side 200 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_8m.json
python - <<'PY'
import json
for f in ('bench','bench_radix','bench_8m'):
    d=json.load(open('gpurun_out/s27/%s.json'%f)); print(f, round(d['ms_per_step'],4), [round(x,3) for x in d['per_step_ms'][:6]], d['config']['warmup_grid_ms'])
PY
