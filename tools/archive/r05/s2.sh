set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_s2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=8 -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|FAILED|Error|loose iteration" $O/tests.log | cut -c1-300 | tail -n 30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err; tail -c 1500 $O/bench_5_20.json; tail -3 $O/bench_5_20.err
timeout 600 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-download-leg > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python - <<'PY'
import json
for f in ("bench_5_20","bench_cfg4"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r05_s2/{f}.json") if l.startswith("{")][-1])
        print(f, "ms/step %.3f"%d["ms_per_step"], "frac %.3f us %.1f"%(d["roofline"]["frac"], d["roofline"]["kernel_us"]), {k:(v.get("kernel_us") if isinstance(v,dict) else v) for k,v in d["roofline"]["other_kernels"].items()}, d.get("with_download"), d["roofline"].get("at_8m"))
    except Exception as e: print(f, "FAILED", e)
PY
SALVA_HIP_NO_TILE_CLASSES=1 timeout 600 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-download-leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 no classes ms/step %.3f frac %.3f us %.1f'%(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us']))"
