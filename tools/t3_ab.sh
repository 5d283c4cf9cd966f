#!/bin/bash
# 4x4x4-cell tiles against the 3x4x4 build on the bench protocols: is the smaller tile better once the block is compressed
# (9-10 slices of 64 particles per 4x4x4 tile for the 8 waves of a workgroup)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in ":0" "t3:0" "t3:512" ":640" "t3:448"; do
  v=${cfg%%:*}; t=${cfg##*:}
  export SALVA_HIP_LIB_VARIANT=$v
  if [ "$t" != "0" ]; then export SALVA_HIP_TILE_THREADS=$t; else unset SALVA_HIP_TILE_THREADS; fi
  timeout 400 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['regimes']; print('variant [$v] threads $t: 5+50 ms/step %.3f first20 %.3f settled %.3f (n=%d)'%(d['ms_per_step'], r['first20']['ms_per_step'], r['settled']['ms_per_step'], r['settled']['steps']), d['config']['tiles']['tile_threads'], [round(x,2) for x in d['per_step_ms'][:2]], [round(x,2) for x in d['per_step_ms'][16:20]])"
done
