#!/bin/bash
# DynamicContactSampling tests + the coupling / parity files the change touches
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dcs
mkdir -p $OUT
cd $R
timeout 500 python -m pytest tests/test_dynamic_sampling_gpu.py tests/test_coupling_gpu.py tests/test_parity_gpu.py tests/test_queries_gpu.py -q -m gpu --durations=5 -s > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -40 $OUT/tests.log
