#!/bin/bash
# tools/gap_trace.sh — kernel trace of a short bench run with per-kernel timestamps, to attribute the idle time between kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
python $R/tools/gap_report.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/gap_report.txt 2>&1
rm -rf $OUT/trace
