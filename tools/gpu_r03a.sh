#!/bin/bash
# round 3, first GPU pass: A/B of the round-2 library, the new product build and the no-mad16 fallback on the bench scene; the GPU
# test suite; a 5 + 20 bench line.  Logs under gpurun_out/r03a/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
SALVA_HIP_LIB_VARIANT=r02 timeout 300 python tools/ab_probe.py --steps 25 --save /tmp/ref25.npy > $O/ab_r02.log 2>&1
timeout 300 python tools/ab_probe.py --steps 25 --ref /tmp/ref25.npy > $O/ab_new.log 2>&1
SALVA_HIP_LIB_VARIANT=nomad timeout 300 python tools/ab_probe.py --steps 25 --ref /tmp/ref25.npy > $O/ab_nomad.log 2>&1
cat $O/ab_*.log | grep -E "^AB|Error|error" 
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_full_size_gpu.py > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5_20.json 2> $O/bench_5_20.err; tail -c 1500 $O/bench_5_20.json
