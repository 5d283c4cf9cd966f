#!/bin/bash
# The last session of round 5: the whole GPU suite from a cold process on the committed tree, then final_c.sh (smoke(), the default bench line).
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_cold_5.log 2>&1; echo "tests rc=$?" >> $O/tests_cold_5.log
grep -E "passed|failed" $O/tests_cold_5.log | tail -1; tail -n 1 $O/tests_cold_5.log
bash tools/r05/final_c.sh 2>&1 | grep -v "^+" | tail -6
