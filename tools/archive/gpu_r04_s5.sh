set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_coupling_gpu.py -x -q > $O/tests_parity.log 2>&1; echo "rc parity $?" >> $O/rc.log
timeout 300 python tools/ab_probe.py --steps 25 >> $O/ab_prod.log 2>&1
timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_prod.log 2>&1
timeout 300 python tools/ab_probe.py --steps 90 >> $O/ab_prod.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_5_50.json 2> $O/bench_5_50.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err
grep "^AB " $O/ab_*.log
cat $O/rc.log
