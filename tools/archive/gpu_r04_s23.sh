#!/bin/bash
# round 4, session 23: k_nbr_tile bounding experiment (variant build nbrexp; the patch is in profiles/r04_experiments/r04n_*)
set -x
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s23
mkdir -p $O
export SALVA_HIP_LIB_VARIANT=nbrexp
for steps in 25 60; do
  for e in 1 2 4 3; do
    SALVA_NBR_EXP=$e timeout 200 python tools/ab_probe.py --steps $steps --kernels 4 --reps 30 2>&1 | grep -E "^AB lib|NBR_EXP" | sed "s/^/steps=$steps exp=$e /" >> $O/nbr_bounds.log
  done
done
cat $O/nbr_bounds.log
