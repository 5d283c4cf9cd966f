#!/bin/bash
# round 3, eighth GPU pass: read-backs through host-mapped publication, the reorder behind the tile-totals wait, lazy Python counters:
# whole GPU suite, bench lines, free-fall timeline.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/${TAG:-r03h}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -6 $O/tests.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_5_50.json 2> $O/bench_5_50.err
python - <<PY
import json
for f in ['$O/bench_5_20.json','$O/bench_5_50.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, 'ms/step', round(d['ms_per_step'],4), 'value %.3e'%d['value'], 'first20', round(d['regimes']['first20']['ms_per_step'],4), 'settled', d['regimes']['settled'] and round(d['regimes']['settled']['ms_per_step'],3), 'kernel_us', round(d['roofline']['kernel_us'],2), 'frac', round(d['roofline']['frac'],3), 'per_step', d['per_step_ms'][:4])
PY
STEPS=8 bash tools/gap_trace.sh > $O/gaps.log 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 3 8 > $O/gap_report.txt 2>&1 || true; head -24 $O/gap_report.txt | cut -c1-160
