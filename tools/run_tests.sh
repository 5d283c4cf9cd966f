cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -60 > gpurun_out/tests.log
