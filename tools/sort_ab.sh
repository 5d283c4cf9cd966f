#!/bin/bash
# A/B of the cell sort's digit width: rocPRIM's 8-bit configuration against digits of up to 11 bits (fewer passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for side in 100 200; do
  for v in default wide; do
    if [ $v = default ]; then export SALVA_HIP_SORT_DEFAULT_DIGITS=1; else unset SALVA_HIP_SORT_DEFAULT_DIGITS; fi
    timeout 300 python bench.py --steps 12 --warmup 3 --side $side --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side $side $v: ms/step %.3f grid_ms %.3f solver_ms %.3f'%(d['ms_per_step'], d['config']['grid_ms'], d['config']['solver_ms']), d['per_step_ms'][:4])"
  done
done
unset SALVA_HIP_SORT_DEFAULT_DIGITS
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -q -m gpu -k "not 32_steps and not 30_steps" 2>&1 | tail -2
