#!/bin/bash
# end-of-round session on the GPU box (one gpurun call): the whole GPU suite, the bench lines (driver protocol 5 + 20 with the
# CPU leg, survey protocol 5 + 50, no flags, the other configurations), the smoke entry, then the rocprofv3 summaries of the
# three single-GPU configurations and of the 8M-particle scene (tools/profile_r03.sh), and the device timeline report.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_final
mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -14 $OUT/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_5_20.json 2> $OUT/bench_5_20.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_5_50.json 2> $OUT/bench_5_50.err
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --config 3 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_5_50.json 2> $OUT/bench_cfg3.err
timeout 600 python bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4_5_50.json 2> $OUT/bench_cfg4.err
timeout 600 python bench.py --steps 20 --warmup 5 --force-slabs --no-cpu-baseline 2> $OUT/bench_force_slabs.err | grep "^{" > $OUT/bench_force_slabs_5_20.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
for f in bench_5_20 bench_5_50 bench_default bench_cfg3_5_50 bench_cfg4_5_50 bench_force_slabs_5_20; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$OUT/$f.json').read().strip().splitlines() if l.startswith('{')][-1])
    print('$f', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'us %.1f'%d['roofline']['kernel_us'], d.get('regimes'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
bash tools/profile_r03.sh r03_cfg2 2>&1 | tail -25
bash tools/profile_r03.sh r03_cfg3 --config 3 2>&1 | tail -22
bash tools/profile_r03.sh r03_cfg4 --config 4 2>&1 | tail -22
bash tools/profile_r03.sh r03_8m --side 200 2>&1 | tail -22
STEPS=25 bash tools/gap_trace.sh > /dev/null 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 3 8 > $OUT/gap_report_free_fall.txt 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv > $OUT/gap_report_last_steps.txt 2>&1
head -4 $OUT/gap_report_free_fall.txt
