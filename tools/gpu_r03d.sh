#!/bin/bash
# round 3, fourth GPU pass: what the 31 us outside the pair loop are made of — tile shapes with more tiles in flight per CU
# (2x4x4: four, 3x4x4: three), phase stamps of the loop-less kernel, and the round-2 persistent pipelines re-measured.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
export AB_PROBE_WATCHDOG=140 SALVA_HIP_SCHED=0
SALVA_HIP_LIB_VARIANT=t2 timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,1 > $O/ab_t2.log 2>&1
SALVA_HIP_LIB_VARIANT=t3 timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,1 > $O/ab_t3.log 2>&1
SALVA_HIP_LIB_VARIANT=dexp1 SALVA_HIP_TILE_TIMING=1 timeout 200 python tools/ab_probe.py --steps 12 --kernels 0 > $O/ab_dexp1.log 2>&1
SALVA_HIP_LIB_VARIANT=diag SALVA_HIP_TILE_TIMING=1 timeout 200 python tools/ab_probe.py --steps 12 --kernels 0 > $O/ab_diag12.log 2>&1
SALVA_HIP_LIB_VARIANT=diag SALVA_HIP_TILE_TIMING=1 timeout 300 python tools/variant_probe.py --steps 12 --variants 0,3,4,2 --reps 20 > $O/variants.log 2>&1
grep -hE "^AB |timing|variant=|Error|error|Traceback|File " $O/*.log | cut -c1-420
