#!/bin/bash
# A/B on one box: the previous commit's library (libsalva_hip_prev.so) against the folded-grid build, alternating.
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
O=gpurun_out/r05_fold_ab; mkdir -p $O
one() {  # tag variant args...
    local tag=$1 var=$2; shift 2
    SALVA_HIP_LIB_VARIANT=$var timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-download-leg --no-big-leg "$@" > $O/$tag.json 2> $O/$tag.err
    python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
ms = d["per_step_ms"]; w = [sum(ms[a:a+50]) / len(ms[a:a+50]) for a in range(0, len(ms), 50)]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "kernel_us %.2f" % d["roofline"]["kernel_us"], "windows", ["%.3f" % x for x in w], flush=True)
PY
}
for rep in 1 2; do
  one c2_prev_$rep prev --steps 20 --warmup 5
  one c2_new_$rep "" --steps 20 --warmup 5
done
one c2l_prev prev --steps 150 --warmup 5
one c2l_new "" --steps 150 --warmup 5
one c4_prev prev --steps 200 --warmup 5 --config 4
one c4_new "" --steps 200 --warmup 5 --config 4
one c3_prev prev --steps 100 --warmup 5 --config 3
one c3_new "" --steps 100 --warmup 5 --config 3
