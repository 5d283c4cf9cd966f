set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py -x -q > $O/tests_parity.log 2>&1; echo "rc parity $?" >> $O/rc.log
for rep in 1 2; do
timeout 300 python tools/ab_probe.py --steps 25 >> $O/ab_prod.log 2>&1
SALVA_HIP_LIB_VARIANT=p2w8 timeout 300 python tools/ab_probe.py --steps 25 >> $O/ab_p2w8.log 2>&1
done
timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_prod.log 2>&1
SALVA_HIP_LIB_VARIANT=p2w8 timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_p2w8.log 2>&1
SALVA_HIP_NO_PLANES=1 timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_legacy.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_5_50.json 2> $O/bench_5_50.err
SALVA_HIP_NO_PLANES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20_legacy.json 2> $O/bench_5_20_legacy.err
grep "^AB " $O/ab_*.log
cat $O/rc.log
