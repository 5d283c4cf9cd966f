#!/bin/bash
# The folded fluid grid (device_types.h TileGrid): its tests, then the long runs that asked for it.
set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
O=gpurun_out/r05_fold; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "fold or stray or isolated" > $O/tests_fold.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests_fold.log
timeout 300 python tools/r05/extent_probe.py 2 1000 > $O/extent_cfg2.txt 2>&1; cat $O/extent_cfg2.txt
timeout 300 python tools/r05/extent_probe.py 3 500 > $O/extent_cfg3.txt 2>&1; cat $O/extent_cfg3.txt
