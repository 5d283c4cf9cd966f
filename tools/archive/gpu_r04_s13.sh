set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err
timeout 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_5_20.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg3_5_20.json 2> $O/bench_cfg3.err
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_numpy_reading_gpu.py tests/test_download_gpu.py tests/test_host_shape_gpu.py tests/test_coupling_gpu.py tests/test_custom_force_gpu.py tests/test_queries_gpu.py tests/test_speculation_gpu.py tests/test_mirrors_gpu.py tests/test_dynamic_sampling_gpu.py -x -q > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
cat $O/rc.log; grep -E "passed|failed" $O/tests.log | tail -n 3
python - <<PY
import json
for f in ['bench_5_20','bench_cfg4_5_20','bench_cfg3_5_20']:
    try:
        j=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
        print(f, '%.4f ms'%j['ms_per_step'], '%.4g'%j['value'], 'frac %.3f us %.1f'%(j['roofline']['frac'], j['roofline']['kernel_us']), j['per_step_ms'][:8], {k:round(v.get('kernel_us',0),1) for k,v in j['roofline'].get('other_kernels',{}).items()})
    except Exception as e:
        print(f,'ERR',e)
PY
