for i in 1 2; do
for v in "" "SALVA_HIP_NO_PUBLISH=1"; do
  for P in "20 5" "50 5"; do set -- $P
    env $v python bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$2+$1', 'ms/step %.3f'%d['ms_per_step'], 'grid %.3f solver %.3f'%(d['config']['grid_ms'], d['config']['solver_ms']))"
  done
done
done
