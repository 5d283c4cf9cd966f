#!/bin/bash
# k_nbr_tile with plane-staged positions + packed distance tests: timing, identical trajectories (sha), contact-set tests
export TMPDIR=/tmp; O=gpurun_out/r03n2; mkdir -p $O
for rep in 1 2; do
  AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 25 --reps 30 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab.log
done
AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --config 3 --steps 15 --reps 10 --kernels 4 2>&1 | grep "^AB lib" >> $O/ab.log
AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --side 200 --steps 8 --reps 10 --kernels 4 2>&1 | grep "^AB lib" >> $O/ab.log
cat $O/ab.log
timeout 600 python -m pytest -q -x tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_dist_gpu.py > $O/tests.log 2>&1; grep -n "passed\|failed\|Error" $O/tests.log | tail -3
