#!/bin/bash
# round 4, session 34: apply kernels at four tiles per CU with a shorter preloaded list head (variants lr12w8, lr8w8, lr12)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s34
mkdir -p $O
for steps in 25 60; do
  for v in product lr12w8 lr8w8 lr12; do
    if [ $v = product ]; then unset SALVA_HIP_LIB_VARIANT; else export SALVA_HIP_LIB_VARIANT=$v; fi
    timeout 200 python tools/ab_probe.py --steps $steps --kernels 0,1,6,4 --reps 30 2>&1 | grep -E "^AB lib" >> $O/ab.log
  done
done
cut -c1-210 $O/ab.log
