#!/bin/bash
# experiment: stagger half of the first round of tiles (s_sleep) so co-resident tiles are out of phase
export TMPDIR=/tmp; O=gpurun_out/r03r; mkdir -p $O
for v in "" sc sd se sf "" sc sd se sf; do
  SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 25 --reps 30 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab.log
done
cat $O/ab.log
