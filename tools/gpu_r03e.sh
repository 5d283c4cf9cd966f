#!/bin/bash
# round 3, fifth GPU pass: the single-buffer persistent skeleton (variant 4) carrying the round-3 pair loop, against the one-tile kernel
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
export SALVA_HIP_SCHED=0 SALVA_HIP_LIB_VARIANT=diag
SALVA_HIP_TILE_TIMING=1 timeout 300 python tools/variant_probe.py --steps 12 --variants 0,4,0,4 --reps 30 > $O/variants12.log 2>&1
timeout 300 python tools/variant_probe.py --steps 25 --variants 0,4,0,4 --reps 30 > $O/variants25.log 2>&1
grep -hE "timing|variant=|Error|error|Traceback" $O/*.log | cut -c1-300
