# round 6, session 14: the halo's positions gathered once per step into tile order (SALVA_HIP_TILE_ORDER=1; the round-5 review's item 3)
# against the default, alternating on one lease: bit-identical (the parity suites under the switch), the bench protocol, kernel times
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_tile_order; mkdir -p $O
for m in 1 2; do SALVA_HIP_TILE_ORDER=$m timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_kernels_gpu.py tests/test_split_gpu.py tests/test_classes_gpu.py tests/test_chain_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tee -a $O/tests.log; done
for rep in 1 2 3; do for v in "" "SALVA_HIP_TILE_ORDER=1" "SALVA_HIP_TILE_ORDER=2"; do
  env $v python bench.py --steps 20 --warmup 5 --no-big-leg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$v]', 'ms/step %.4f' % d['ms_per_step'], r['kernel'], '%.2f us' % r['kernel_us'], ' '.join('%s %.2f' % (k, v['kernel_us']) for k, v in r['other_kernels'].items()))"
done; done | tee $O/bench_ab.log
