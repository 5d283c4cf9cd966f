#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/nbr
mkdir -p $OUT
cd $R
for CFG in "2 3" "2 22" "4 3"; do
  set -- $CFG
  for V in 0 1; do
    SALVA_HIP_NBR_VARIANT=$V timeout 200 python tools/nbr_probe.py $1 $2 2>&1 | tail -1
  done
done | tee $OUT/probe2.log
timeout 400 python -m pytest tests/test_parity_gpu.py tests/test_mirrors_gpu.py tests/test_fuzz_gpu.py tests/test_dynamic_sampling_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee -a $OUT/probe2.log
