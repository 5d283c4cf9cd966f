#!/bin/bash
# A/B: brick slot order (current) vs linear slot order (libsalva_hip_lin.so) — kernel timings + whole-step times
export TMPDIR=/tmp; O=gpurun_out/r03k; mkdir -p $O
for v in lin "" lin ""; do
  SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=90 timeout 120 python tools/ab_probe.py --steps 25 --reps 30 --kernels 0,1,4 >> $O/ab.log 2>&1
done
SALVA_HIP_LIB_VARIANT=lin timeout 120 python tools/ab_probe.py --side 200 --steps 8 --reps 10 --kernels 0,1,4 >> $O/ab8m.log 2>&1
timeout 120 python tools/ab_probe.py --side 200 --steps 8 --reps 10 --kernels 0,1,4 >> $O/ab8m.log 2>&1
cat $O/ab.log $O/ab8m.log
