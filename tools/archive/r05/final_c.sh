#!/bin/bash
# Last session of round 5: smoke(), then the default bench line (traffic from the committed r05_cfg2 profile).
set -x
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r05_final_c; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_5_20.json 2> $O/bench_5_20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_final_c/bench_5_20.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_us'], r['frac'], r['traffic'], r.get('traffic_source'), r.get('frac_measured_traffic'))
print(d.get('with_download'), r.get('at_8m'))
PY
