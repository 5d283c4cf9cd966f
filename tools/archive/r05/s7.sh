cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench_5_20.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-big-leg > $O/bench_5_50.json 2> $O/bench_5_50.err
timeout 900 python tools/pcie_probe.py > $O/pcie.log 2>&1; cat $O/pcie.log
for f in bench_5_20 bench_5_50; do python - <<PY
import json
d=json.loads([l for l in open('$O/$f.json').read().strip().splitlines() if l.startswith('{')][-1])
wd=d.get('with_download') or {}
print('$f', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'us %.1f'%d['roofline']['kernel_us'], 'traffic', d['roofline']['traffic'], d['roofline']['frac_measured_traffic'], {k:(v and round(v.get('ms_per_step',0),3)) for k,v in d['regimes'].items()}, 'dl', wd.get('ms_per_step'), wd.get('value'), '8m', (d['roofline'].get('at_8m') or {}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
done
