set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_numpy_reading_gpu.py tests/test_fuzz_gpu.py -x -q -s > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
SALVA_HIP_TILE_TRACE=1 timeout 300 python tools/ab_probe.py --config 3 --steps 25 > $O/ab_cfg3_trace.log 2>&1
timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep "^AB " >> $O/ab.log
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg3_5_20.json 2> $O/bench_cfg3.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace3 -o trace -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace3.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg3_kernel_stats.csv; rm -rf $O/trace3
grep "^AB" $O/ab.log $O/ab_cfg3_trace.log; grep "salva_hip tiles" $O/ab_cfg3_trace.log | tail -n 2; cat $O/rc.log; grep -E "passed|failed|device vs numpy" $O/tests.log | tail -n 5; head -n 16 $O/cfg3_kernel_stats.csv | cut -c1-150
