#!/bin/bash
# round 4, session 31: A/B of the end-of-step changes against the previous commit's library (libsalva_hip_prev.so), same box
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s31
mkdir -p $O
for rep in 1 2 3; do
  for v in prev product; do
    if [ $v = prev ]; then export SALVA_HIP_LIB_VARIANT=prev; else unset SALVA_HIP_LIB_VARIANT; fi
    timeout 200 python tools/ab_probe.py --steps 25 --kernels 0 --reps 5 2>&1 | grep -E "^AB lib" >> $O/ab.log
  done
done
cut -c1-200 $O/ab.log
