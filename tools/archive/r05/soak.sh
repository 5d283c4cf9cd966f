#!/bin/bash
# Long runs of the three single-GPU configurations: where each scene settles (iterations, ms per step, throughput per window of
# 100 steps), that nothing drifts or blows up over a thousand steps, and that the device memory in use stops growing.
set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
O=gpurun_out/r05_soak; mkdir -p $O
run() {  # tag, steps, bench args...
    local tag=$1 steps=$2; shift 2
    timeout 900 python bench.py --gpus 1 --steps $steps --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg "$@" > $O/$tag.json 2> $O/$tag.err
    echo "$tag rc=$?"
}
run cfg2_1000 1000
run cfg3_500 500 --config 3
run cfg4_300 300 --config 4
python tools/r05/soak_report.py $O/cfg2_1000.json $O/cfg3_500.json $O/cfg4_300.json | tee $O/report.txt
