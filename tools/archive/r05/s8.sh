cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_s8; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_numpy_reading_gpu.py tests/test_fuzz_gpu.py tests/test_mirrors_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -n 12 $O/tests.log | cut -c1-400
run() { name=$1; shift; env "$@" | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); o=d['roofline']['other_kernels']
print('$name', 'ms/step %.3f'%d['ms_per_step'], 'settled', (d['regimes']['settled'] or {}).get('ms_per_step'), d['roofline']['kernel'], 'us %.1f frac %.3f'%(d['roofline']['kernel_us'], d['roofline']['frac']), {k:round(v.get('kernel_us',0),1) for k,v in o.items()})"; }
B="timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg"
run cfg4_two_mass $B --config 4 2>/dev/null
run cfg4_general SALVA_HIP_NO_TWO_MASS=1 $B --config 4 2>/dev/null
run cfg4_two_mass_50 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg --config 4 2>/dev/null
run cfg2 $B 2>/dev/null
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -m gpu -x -k "config4" 2>&1 | tail -3
