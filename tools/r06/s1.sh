# round 6, session 1: where this round starts on today's box — the 5 + 20 line, the host-placement probe, the device timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_s1
mkdir -p $O
nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -i -E "model name|socket|numa|mhz" | head -12
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg > $O/bench_5_20.json 2> $O/bench_5_20.err; tail -c 1500 $O/bench_5_20.json | head -c 1500; echo
timeout 900 python tools/r06/hostile_host.py 2>&1 | tee $O/hostile_host.log
STEPS=20 bash tools/gap_trace.sh > $O/gap_trace.log 2>&1; tail -2 $O/gap_trace.log
python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 6 14 > $O/gap_report_free_fall.txt 2>&1; head -24 $O/gap_report_free_fall.txt
