set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s19; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_speculation_gpu.py tests/test_fuzz_gpu.py tests/test_mirrors_gpu.py tests/test_custom_force_gpu.py -x -q > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
timeout 300 python tools/host_overhead.py > $O/host_overhead.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err
timeout 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_5_20.json 2> $O/bench_cfg4.err
cat $O/rc.log $O/host_overhead.log; grep -E "passed|failed|^E " $O/tests.log | tail -n 5
python - <<PY
import json
for f in ['bench_5_20','bench_cfg4_5_20']:
    j=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
    print(f, '%.4f ms'%j['ms_per_step'], '%.4g'%j['value'], 'frac %.3f us %.1f'%(j['roofline']['frac'], j['roofline']['kernel_us']), j['per_step_ms'][:6], j['regimes']['settled'])
PY
