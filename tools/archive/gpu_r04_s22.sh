#!/bin/bash
# round 4, session 22: DynamicContactSampling in decomposed worlds
set -x
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s22
mkdir -p $O
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_dynamic_sampling_gpu.py tests/test_host_shape_gpu.py -q -m gpu -x > $O/tests.log 2>&1
echo "tests rc=$?"
tail -n 40 $O/tests.log
