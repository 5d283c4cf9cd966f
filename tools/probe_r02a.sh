#!/bin/bash
# GPU session: sanity tests, then the k_pred_density variant matrix.
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02a_tests.log
tail -3 gpurun_out/r02a_tests.log
OUT=gpurun_out/r02a_probe.log
: > $OUT
for lib in "" t3; do
  for steps in 6 30; do
    for jit in 0.1 0; do
      SALVA_HIP_LIB_VARIANT=$lib SALVA_HIP_PIPE_WAVES=8 timeout 300 python tools/variant_probe.py --steps $steps --jitter $jit >> $OUT 2>&1
    done
  done
  SALVA_HIP_LIB_VARIANT=$lib SALVA_HIP_PIPE_WAVES=12 timeout 300 python tools/variant_probe.py --steps 30 --jitter 0.1 --variants 0,2 >> $OUT 2>&1
  SALVA_HIP_LIB_VARIANT=$lib SALVA_HIP_PIPE_WAVES=6 timeout 300 python tools/variant_probe.py --steps 30 --jitter 0.1 --variants 0,2 >> $OUT 2>&1
done
cat $OUT
