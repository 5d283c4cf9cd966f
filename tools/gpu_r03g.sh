#!/bin/bash
# round 3, seventh GPU pass: the committed profile of config 2 (trace + PMC), the device timeline of free-fall steps (gaps), the new
# decomposed-query test.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 300 python -m pytest -q -x tests/test_dist_gpu.py::test_queries_in_a_decomposed_run tests/test_queries_gpu.py tests/test_parity_gpu.py > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
bash tools/profile_r03.sh r03_cfg2 > $O/profile_cfg2.log 2>&1; tail -32 $O/profile_cfg2.log | cut -c1-200
STEPS=8 bash tools/gap_trace.sh > $O/gaps.log 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 3 8 > $O/gap_report.txt 2>&1 || true; head -60 $O/gap_report.txt | cut -c1-160
