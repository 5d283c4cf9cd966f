cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_now -o t -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/trace_now.log 2>&1
rm -f $R/gpurun_out/trace_now/*kernel_trace.csv
