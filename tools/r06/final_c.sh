# round 6, end-of-round session C: the bench lines again, now that profiles/r06_cfg* hold the counters of THIS tree (final_b.sh):
# roofline.traffic is quoted when the committed profile's kernel-source sha is the tree's -> gpurun_out/r06_final
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench_5_20.err
timeout 900 python bench.py --steps 50 --warmup 5 --no-big-leg > $O/bench_5_50.json 2> $O/bench_5_50.err
timeout 900 python bench.py --steps 20 --warmup 5 --config 3 --no-cpu-baseline > $O/bench_cfg3_5_20.json 2> $O/bench_cfg3.err
timeout 900 python bench.py --steps 20 --warmup 5 --config 4 --no-cpu-baseline > $O/bench_cfg4_5_20.json 2> $O/bench_cfg4.err
timeout 900 python bench.py --steps 20 --warmup 5 --side 200 --no-cpu-baseline --no-download-leg > $O/bench_8m_5_20.json 2> $O/bench_8m.err
python - <<PY
import json
for f in ("bench_default","bench_5_20","bench_5_50","bench_cfg3_5_20","bench_cfg4_5_20","bench_8m_5_20"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["steps"], d["warmup"], round(d["ms_per_step"],4), "%.3e"%d["value"], r["kernel"], round(r["kernel_us"],2), round(r["frac"],4), "traffic", r.get("traffic"), r.get("frac_measured_traffic"), (r.get("traffic_source") or "")[:40])
    except Exception as e: print(f, "FAILED", e)
PY
