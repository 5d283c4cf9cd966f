#!/bin/bash
# tools/gpu_session.sh TAG — one GPU session: the GPU test suite, the bench line at the driver's protocol (5 + 20) and at the
# survey's (5 + 50), and a kernel trace of the latter.  Everything lands in gpurun_out/TAG/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_5_20.json 2> $OUT/bench_5_20.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_5_50.json 2> $OUT/bench_5_50.err
cat $OUT/bench_5_20.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('5+20:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'], d['config']['grid_ms'], d['config']['solver_ms'])"
cat $OUT/bench_5_50.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('5+50:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'], d['config']['grid_ms'], d['config']['solver_ms'])"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/trace.log 2>&1
cd $R
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/trace_kernel_stats.csv && head -30 $OUT/trace_kernel_stats.csv | cut -d, -f1-8
# keep only the summaries (the raw trace is large)
find $OUT/trace -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
