cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/fullsize.log
