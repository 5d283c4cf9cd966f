set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_s3; mkdir -p $O
timeout 900 python -m pytest tests/test_cfl_gpu.py tests/test_queries_gpu.py tests/test_parity_gpu.py tests/test_coupling_gpu.py tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -n 15 $O/tests.log | cut -c1-300
run() { # name, env..., args
  name=$1; shift
  env "$@" | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); o=d['roofline']['other_kernels']
print('$name', 'ms/step %.3f'%d['ms_per_step'], 'first20 %.3f'%d['regimes']['first20']['ms_per_step'], 'settled', (d['regimes']['settled'] or {}).get('ms_per_step'), d['roofline']['kernel'], 'us %.1f frac %.3f'%(d['roofline']['kernel_us'], d['roofline']['frac']), {k:round(v.get('kernel_us',0),1) for k,v in o.items()})"
}
B="timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg"
run cfg3 $B --config 3 2>/dev/null
run cfg4_fork $B --config 4 2>/dev/null
run cfg4_nofork SALVA_HIP_NO_CLASS_FORK=1 $B --config 4 2>/dev/null
run cfg4_noclasses SALVA_HIP_NO_TILE_CLASSES=1 $B --config 4 2>/dev/null
run cfg4_fork_50 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg --config 4 2>/dev/null
run cfg4_noclasses_50 SALVA_HIP_NO_TILE_CLASSES=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg --config 4 2>/dev/null
run cfg2_spec SALVA_HIP_SPECULATE=1 $B 2>/dev/null
run cfg2 $B 2>/dev/null
