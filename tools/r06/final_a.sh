# round 6, end-of-round session A: the GPU suite from a cold process, smoke(), the bench lines, the slow-host table (this build; the
# round-5 library's rows are in profiles/r06_final/slow_host_r05_vs_r06.log of the first final session), the device timeline -> gpurun_out/r06_final
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
gcc -O2 -shared -fPIC -o tools/r06/slow_host.so tools/r06/slow_host.c -ldl
timeout 2000 python -m pytest tests -x -q -m gpu > $O/tests_cold_1.log 2>&1; tail -3 $O/tests_cold_1.log | head -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench_5_20.err
timeout 900 python bench.py --steps 50 --warmup 5 --no-big-leg > $O/bench_5_50.json 2> $O/bench_5_50.err
timeout 900 python bench.py --steps 20 --warmup 5 --config 3 --no-cpu-baseline > $O/bench_cfg3_5_20.json 2> $O/bench_cfg3.err
timeout 900 python bench.py --steps 20 --warmup 5 --config 4 --no-cpu-baseline > $O/bench_cfg4_5_20.json 2> $O/bench_cfg4.err
timeout 900 python bench.py --steps 20 --warmup 5 --side 200 --no-cpu-baseline --no-download-leg > $O/bench_8m_5_20.json 2> $O/bench_8m.err
python - <<PY
import json
for f in ("bench_5_20","bench_5_50","bench_cfg3_5_20","bench_cfg4_5_20","bench_8m_5_20"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],4), "%.3e"%d["value"], d["roofline"]["kernel"], round(d["roofline"]["kernel_us"],2), round(d["roofline"]["frac"],4), d.get("with_download") and round(d["with_download"]["ms_per_step"],3))
    except Exception as e: print(f, "FAILED", e)
PY
for lib in ""; do for us in 0 5 10; do
  echo "lib=[$lib] SLOW_HOST_US=$us $(SALVA_HIP_LIB_VARIANT=$lib SLOW_HOST_US=$us LD_PRELOAD=$PWD/tools/r06/slow_host.so HH_ROLE=child python tools/r06/hostile_host.py 2>&1 | grep '^HH')"
done; done | tee $O/slow_host_r06_late.log
timeout 900 python tools/r06/hostile_host.py 2>&1 | tee $O/hostile_host.log | tail -12
STEPS=20 bash tools/gap_trace.sh > $O/gap_trace.log 2>&1
python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 6 14 > $O/gap_report_free_fall.txt 2>&1; head -3 $O/gap_report_free_fall.txt
