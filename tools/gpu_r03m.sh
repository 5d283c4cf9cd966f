#!/bin/bash
# A/B: block -> slot mapping across the 8 XCDs: contiguous eighths (product), off, groups of 4/16/64 slots in turn
export TMPDIR=/tmp; O=gpurun_out/r03m; mkdir -p $O
for v in "" g16 g32 g64 g128 "" g16 g32 g64 g128; do
  SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 25 --reps 30 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab.log
done
for v in g32 g64 g128; do
  SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --side 200 --steps 8 --reps 10 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab8m.log
done
cat $O/ab.log $O/ab8m.log
