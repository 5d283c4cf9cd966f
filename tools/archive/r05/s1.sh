set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_s1; mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu --durations=12 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -n 30 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench_5_20.err; tail -c 3000 $O/bench_5_20.json
