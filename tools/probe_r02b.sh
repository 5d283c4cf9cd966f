#!/bin/bash
OUT=gpurun_out/r02b_probe.log
: > $OUT
for steps in 6 30; do
  SALVA_HIP_PIPE_WAVES=8 timeout 300 python tools/variant_probe.py --steps $steps --jitter 0.1 --variants 0,3,4,2 >> $OUT 2>&1
done
SALVA_HIP_PIPE_WAVES=9 timeout 300 python tools/variant_probe.py --steps 30 --jitter 0.1 --variants 0,4 >> $OUT 2>&1
SALVA_HIP_PIPE_WAVES=8 SALVA_HIP_TILE_THREADS=576 timeout 300 python tools/variant_probe.py --steps 30 --jitter 0.1 --variants 0,3 >> $OUT 2>&1
cat $OUT
