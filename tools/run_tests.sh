cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu --durations=4 2>&1 | tail -30 > gpurun_out/tests.log
