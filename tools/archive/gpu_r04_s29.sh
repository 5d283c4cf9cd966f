#!/bin/bash
# round 4, session 29: fewer launches in the grid phase (one-workgroup tile slots / TileAcc scan): sha, tests, bench
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s29
mkdir -p $O
timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,4 --reps 20 2>&1 | grep -E "^AB lib" > $O/probe.log
timeout 300 python tools/ab_probe.py --side 200 --steps 6 --kernels 0 --reps 5 2>&1 | grep -E "^AB lib" >> $O/probe.log
cat $O/probe.log | cut -c1-230
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_fuzz_gpu.py tests/test_speculation_gpu.py tests/test_kernels_gpu.py tests/test_config5_gpu.py -q -m gpu -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -n 4 $O/tests.log
for rep in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_$rep.json
done
timeout 300 python bench.py --side 200 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_8m.json
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_cfg3.json
python - <<'PY'
import json
for f in ('bench_1','bench_2','bench_8m','bench_cfg3'):
    d=json.load(open('gpurun_out/s29/%s.json'%f)); print(f, round(d['ms_per_step'],4), [round(x,3) for x in d['per_step_ms'][:6]], d['config']['warmup_grid_ms'])
PY
