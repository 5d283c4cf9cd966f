cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for v in prev "" post; do
  for st in 8 40; do
    echo "variant=[$v] steps=$st"; SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps $st --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -1
  done
done
done
for v in prev "" post; do SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --side 200 --steps 8 --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -1; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg --config 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); o=d['roofline']['other_kernels']
print('cfg4', 'ms/step %.3f'%d['ms_per_step'], 'settled', (d['regimes']['settled'] or {}).get('ms_per_step'), d['roofline']['kernel'], 'us %.1f frac %.3f'%(d['roofline']['kernel_us'], d['roofline']['frac']), {k:round(v.get('kernel_us',0),1) for k,v in o.items()})"
