#!/bin/bash
# The whole GPU suite on the folded-grid build, then the long runs (tools/r05/soak.sh).
set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
O=gpurun_out/r05_fold; mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu > $O/tests_all.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests_all.log
bash tools/r05/soak.sh 2>&1 | grep -v "^+" | tail -40
