cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/dist_test.log
