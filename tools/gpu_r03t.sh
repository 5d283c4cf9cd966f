#!/bin/bash
# experiment: would a third resident tile per CU pay for the evaluate kernels?  Less dense scene (smoothing 1.8) whose P|W
# halo fits 51 KB: e2 = 92 VGPR (2 tiles/CU), e3 = 80 VGPR + 51 KB (3 tiles/CU), e3p = 80 VGPR, LDS padded to 70 KB (2 tiles/CU)
export TMPDIR=/tmp; O=gpurun_out/r03t; mkdir -p $O
for rep in 1 2; do for v in e2 e3 e3p; do
  AB_SMOOTHING=1.8 SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 25 --reps 30 --kernels 0,1 2>&1 | grep "^AB lib" >> $O/ab.log
done; done
cat $O/ab.log
