#!/bin/bash
# round 4, session 35: the IISPH density pass held to 80 VGPRs (three tiles per CU) against its natural 84 (variant da5)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s35
mkdir -p $O
for rep in 1 2 3; do
  for v in product da5; do
    if [ $v = product ]; then unset SALVA_HIP_LIB_VARIANT; else export SALVA_HIP_LIB_VARIANT=$v; fi
    timeout 200 python tools/ab_probe.py --config 3 --steps 25 --kernels 2 --reps 10 2>&1 | grep -E "^AB lib" >> $O/ab.log
  done
done
cut -c1-200 $O/ab.log
