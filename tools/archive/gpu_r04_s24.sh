#!/bin/bash
# round 4, session 24: every LDS layout instantiation on hardware (SALVA_HIP_DS_LEVEL)
set -x
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s24
mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > $O/tests.log 2>&1
echo "tests rc=$?"
tail -n 40 $O/tests.log
