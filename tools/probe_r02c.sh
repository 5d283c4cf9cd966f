#!/bin/bash
OUT=gpurun_out/r02c_probe.log
: > $OUT
SALVA_HIP_TILE_TIMING=1 SALVA_HIP_PIPE_WAVES=8 timeout 300 python tools/variant_probe.py --steps 6 --jitter 0.1 --variants 0,4,2 >> $OUT 2>&1
SALVA_HIP_TILE_TIMING=1 timeout 300 python -c "
import sys
sys.path.insert(0,'.')
import bench
fl, sh = bench.build_scene(100)
w, f = bench.make_world(fl, sh, 0)
for k in range(6): w.step(bench.DT, bench.GRAVITY)
print('classic us', w.time_pred_density(20))
" >> $OUT 2>&1
cat $OUT
