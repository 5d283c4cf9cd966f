#!/bin/bash
# round 3, sixth GPU pass: DPP reductions in; k_nbr_tile variants; the whole GPU suite; a 5 + 20 and a 5 + 50 bench line.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
export AB_PROBE_WATCHDOG=140
timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,1,4 --save /tmp/ref.npy > $O/ab_new.log 2>&1
SALVA_HIP_NBR_VARIANT=2 timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,1,4 --ref /tmp/ref.npy > $O/ab_nbr2.log 2>&1
grep -hE "^AB |Error|error|Traceback|File " $O/ab_*.log | cut -c1-420
timeout 1200 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -6 $O/tests.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err; tail -c 700 $O/bench_5_20.json; echo
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_5_50.json 2> $O/bench_5_50.err; python - <<PY
import json
d=json.loads(open('$O/bench_5_50.json').read().strip().splitlines()[-1])
print('5+50:', d['ms_per_step'], d['regimes']['settled'], d['roofline']['kernel_us'], d['roofline']['frac'], d['config']['tiles'])
PY
