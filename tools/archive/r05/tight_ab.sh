#!/bin/bash
# Folding target: stop at the trigger (4 n + 2^20 cells: variant "loose") against folding down to n / 2 cells (the build), on one box:
# the stray-heavy ends of the long runs.  Second use: variant "wt" (-DSALVA_TILE_WAVES_WEIGHTED: workgroups sized for the tile the
# average particle lives in, capped at eight waves like the default) against the build.
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
O=gpurun_out/r05_tight_ab; mkdir -p $O
one() {  # tag variant args...
    local tag=$1 var=$2; shift 2
    SALVA_HIP_LIB_VARIANT=$var timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-download-leg --no-big-leg "$@" > $O/$tag.json 2> $O/$tag.err
    python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
ms = d["per_step_ms"]; w = [sum(ms[a:a+100]) / len(ms[a:a+100]) for a in range(0, len(ms), 100)]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "tiles", d["config"]["tiles"], "windows of 100", ["%.3f" % x for x in w], flush=True)
PY
}
for v in wt "" wt ""; do
  one c2_${v:-tight} "$v" --steps 1000 --warmup 5
  one c3_${v:-tight} "$v" --steps 500 --warmup 5 --config 3
  one c4_${v:-tight} "$v" --steps 300 --warmup 5 --config 4
done
