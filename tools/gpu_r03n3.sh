#!/bin/bash
# where does k_nbr_tile's time go?  ns = never store a list dword, na = no append loop at all (tests + popcount only)
export TMPDIR=/tmp; O=gpurun_out/r03n3; mkdir -p $O
for v in "" ns na; do
  SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 3 --reps 30 --kernels 4 2>&1 | grep "^AB lib" >> $O/ab.log
done
cat $O/ab.log
