#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r03s; mkdir -p $O
for rep in 1 2; do
  AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 25 --reps 30 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab.log
done
AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --config 3 --steps 15 --reps 10 --kernels 2,3 2>&1 | grep "^AB lib" >> $O/ab.log
AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --side 200 --steps 8 --reps 10 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab.log
cat $O/ab.log
timeout 900 python -m pytest -q -x tests -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench_5_20.err; cat $O/bench_5_20.json
