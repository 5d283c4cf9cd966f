#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 1700 python -m pytest tests -q -m gpu --durations=8 -s -k "basic3 or dam_break or coupl or fuzz" > $OUT/newtests.log 2>&1; echo "rc=$?" >> $OUT/newtests.log
tail -25 $OUT/newtests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/trace.log 2>&1
cd $R
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/trace_kernel_stats.csv && head -12 $OUT/trace_kernel_stats.csv | cut -d, -f1-6
find $OUT/trace -name "*kernel_trace.csv" -delete
tail -2 $OUT/trace.log
