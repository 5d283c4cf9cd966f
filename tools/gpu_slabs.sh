#!/bin/bash
# the decomposed code path: its tests, then bench.py over a 1-rank RCCL communicator and 2 / 4 loopback slabs at bench size
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/slabs
mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_rccl_multi_gpu.py -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --force-slabs --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_force_slabs.json 2> $OUT/bench_force_slabs.err; echo "rc=$?"
python -c "
import json; d=json.loads(open('$OUT/bench_force_slabs.json').read().strip().splitlines()[-1]); print('force-slabs ms/step %.3f'%d['ms_per_step'], 'grid %.3f solver %.3f'%(d['config']['grid_ms'], d['config']['solver_ms']), d['per_step_ms'])"
timeout 600 python tools/loopback_scale.py > $OUT/loopback_scale.log 2>&1; echo "rc=$?"; tail -4 $OUT/loopback_scale.log
