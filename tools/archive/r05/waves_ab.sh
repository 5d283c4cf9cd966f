#!/bin/bash
# Workgroup size of the tile kernels on one box.  First session: the round-4 rule (one wave per slice of the average tile, 4..12
# waves) against +1 / -1 / -2 waves (variants wm1, wm2, wp1 = -DSALVA_TILE_WAVES_BIAS).  Second session (this list): the previous
# commit's library against the cap of eight waves and a cap of seven (-DSALVA_TILE_WAVES_CAP=7).
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
O=gpurun_out/r05_waves_ab; mkdir -p $O
one() {  # tag variant args...
    local tag=$1 var=$2; shift 2
    SALVA_HIP_LIB_VARIANT=$var timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-download-leg --no-big-leg "$@" > $O/$tag.json 2> $O/$tag.err
    python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
ms = d["per_step_ms"]; w = [sum(ms[a:a+50]) / len(ms[a:a+50]) for a in range(0, len(ms), 50)]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "threads", d["config"]["tiles"]["tile_threads"], "windows", ["%.3f" % x for x in w], flush=True)
PY
}
for v in prev "" c7 prev ""; do
  one c2s_${v:-new} "$v" --steps 20 --warmup 5
  one c2l_${v:-new} "$v" --steps 250 --warmup 5
  one c4_${v:-new} "$v" --steps 200 --warmup 5 --config 4
  one c3_${v:-new} "$v" --steps 100 --warmup 5 --config 3
done
