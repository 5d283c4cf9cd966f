#!/bin/bash
# heavy-tiles-first slot numbering + grouped XCD mapping (SALVA_HIP_XCD_LG = 1 + log2 group; 0 = no remapping)
export TMPDIR=/tmp; O=gpurun_out/r03p; mkdir -p $O
for rep in 1 2; do for k in 7 5 8; do
  echo -n "lg=$k " >> $O/ab.log
  SALVA_HIP_XCD_LG=$k AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 25 --reps 30 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab.log
done; done
for k in 7 5 8; do
  echo -n "lg=$k " >> $O/ab8m.log
  SALVA_HIP_XCD_LG=$k AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --side 200 --steps 8 --reps 10 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab8m.log
done
for k in 7 5 8; do
  echo -n "cfg3 lg=$k " >> $O/ab3.log
  SALVA_HIP_XCD_LG=$k AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --config 3 --steps 15 --reps 10 --kernels 2,3 2>&1 | grep "^AB lib" >> $O/ab3.log
done
cat $O/ab.log $O/ab8m.log $O/ab3.log
timeout 300 python -m pytest -q -x tests/test_parity_gpu.py tests/test_dist_gpu.py > $O/tests.log 2>&1; tail -3 $O/tests.log
