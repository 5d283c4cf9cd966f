cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r01c.json 2> gpurun_out/bench_r01c.err
bash tools/profile.sh r01c > gpurun_out/profile_r01c.log 2>&1
# keep only the summaries (the raw kernel trace is hundreds of MB)
find gpurun_out/r01c -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/r01c
