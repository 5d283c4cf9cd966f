cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/leak -o t -- python $R/tools/leak_probe.py 1200 > $R/gpurun_out/leak.log 2>&1
rm -f $R/gpurun_out/leak/*kernel_trace.csv
