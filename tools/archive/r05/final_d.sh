# end-of-round session D (after the folded grid and the workgroup sizing): the whole GPU suite once from a cold process, the smoke entry,
# then the bench lines: driver protocol 5 + 20 with every leg, survey protocol 5 + 50, configs 3 and 4, 8 x 10^6.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_final
mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for k in 1; do
  timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests_cold_$k.log 2>&1; echo "tests rc=$?" >> $OUT/tests_cold_$k.log
  tail -n 3 $OUT/tests_cold_$k.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_5_20.json 2> $OUT/bench_5_20.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-big-leg > $OUT/bench_5_50.json 2> $OUT/bench_5_50.err
timeout 600 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_5_20.json 2> $OUT/bench_cfg3.err
timeout 600 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4_5_20.json 2> $OUT/bench_cfg4.err
timeout 600 python bench.py --side 200 --steps 20 --warmup 5 --no-cpu-baseline --no-download-leg > $OUT/bench_8m_5_20.json 2> $OUT/bench_8m.err
for f in bench_5_20 bench_5_50 bench_cfg3_5_20 bench_cfg4_5_20 bench_8m_5_20; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$OUT/$f.json').read().strip().splitlines() if l.startswith('{')][-1])
    wd=d.get('with_download') or {}
    print('$f', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'us %.1f'%d['roofline']['kernel_us'], {k:(v and round(v.get('ms_per_step',0),3)) for k,v in d['regimes'].items()}, 'dl', wd.get('ms_per_step'), '8m', (d['roofline'].get('at_8m') or {}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
