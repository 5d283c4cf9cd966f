set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
./tools/ubench/d2h_bw > $O/d2h_bw.log 2>&1
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_kernels_gpu.py -x -q > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
for rep in 1 2; do
timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep "^AB " >> $O/ab.log
SALVA_HIP_NO_PLANES=1 timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep "^AB " | sed "s/^/legacy /" >> $O/ab.log
done
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg3_5_20.json 2> $O/bench_cfg3.err
cat $O/d2h_bw.log $O/ab.log $O/rc.log; tail -n 3 $O/tests.log; tail -c 600 $O/bench_cfg3_5_20.json
