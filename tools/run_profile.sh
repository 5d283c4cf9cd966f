# tools/run_profile.sh TAG — full bench line (with cpu_baseline) + rocprofv3 kernel trace and PMC passes; summaries only are kept.
TAG=${1:-r01d}
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
bash tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
python tools/summarize_pmc.py gpurun_out/$TAG > gpurun_out/$TAG/hbm_traffic.txt 2>&1
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
find gpurun_out/$TAG -name "*counter_collection.csv" -delete
du -sh gpurun_out/$TAG
