#!/bin/bash
OUT=gpurun_out/r02f_probe.log
: > $OUT
SALVA_HIP_PIPE_WAVES=8 timeout 300 python tools/variant_probe.py --steps 6 --jitter 0.1 --variants 0,3,8:0,8:1 >> $OUT 2>&1
SALVA_HIP_PIPE_WAVES=8 timeout 300 python tools/variant_probe.py --steps 30 --jitter 0.1 --variants 0,3,8:0,8:1 >> $OUT 2>&1
SALVA_HIP_PIPE_WAVES=8 SALVA_HIP_TILE_THREADS=576 timeout 300 python tools/variant_probe.py --steps 30 --jitter 0.1 --variants 0,3,7 >> $OUT 2>&1
cat $OUT
