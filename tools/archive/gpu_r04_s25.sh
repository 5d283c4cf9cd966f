#!/bin/bash
# round 4, session 25: single-rank communicator with repeatable passes (deferred list check): tests + --force-slabs overhead
set -x
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s25
mkdir -p $O
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_peer_transport_gpu.py tests/test_speculation_gpu.py -q -m gpu -x > $O/tests.log 2>&1
echo "tests rc=$?"
tail -n 8 $O/tests.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench_plain_$rep.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-slabs 2>/dev/null | tail -n 1 > $O/bench_slabs_rccl_$rep.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-slabs --transport peer 2>/dev/null | tail -n 1 > $O/bench_slabs_peer_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s25/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],4), d['config'].get('discarded_passes'))
    except Exception as e: print(f,'ERR',e)
PY
