#!/bin/bash
# round 3, third GPU pass: list scheduling on/off (kernel times, trajectories), the decomposition experiments (exp1: no pair loop,
# exp2: LDS reads only, exp3: arithmetic only) and the parity tests with scheduling forced on.  Logs under gpurun_out/r03c/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
export AB_PROBE_WATCHDOG=140
timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,1,5,4 --sched 0,1 --save /tmp/ref.npy --ref /tmp/ref.npy > $O/ab_sched.log 2>&1
for e in 1 2 3; do
  SALVA_HIP_LIB_VARIANT=exp$e timeout 200 python tools/ab_probe.py --steps 25 --kernels 0,1 --sched 0,1 > $O/ab_exp$e.log 2>&1
done
grep -hE "^AB |Error|error|Traceback|File " $O/ab_*.log | cut -c1-420
SALVA_HIP_SCHED=1 timeout 600 python -m pytest -q -x tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_dist_gpu.py > $O/tests_sched1.log 2>&1; tail -8 $O/tests_sched1.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err; tail -c 900 $O/bench_5_20.json
