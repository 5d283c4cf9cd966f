#!/bin/bash
# round 4, session 26: XCD slot-group size and waves per tile (variant build xcdexp)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s26
mkdir -p $O
export SALVA_HIP_LIB_VARIANT=xcdexp
for steps in 25 60; do
  for lg in 7 8 9 6 0; do
    SALVA_HIP_XCD_LG=$lg timeout 200 python tools/ab_probe.py --steps $steps --kernels 0,1,6,4 --reps 30 2>&1 | grep -E "^AB lib" | sed "s/^/xcd_lg=$lg /" >> $O/xcd.log
  done
  for th in 512 384 256; do
    SALVA_HIP_TILE_THREADS=$th timeout 200 python tools/ab_probe.py --steps $steps --kernels 0,1,6,4 --reps 30 2>&1 | grep -E "^AB lib" | sed "s/^/threads=$th /" >> $O/threads.log
  done
done
cat $O/xcd.log $O/threads.log | cut -c1-230
