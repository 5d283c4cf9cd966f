set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s18; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -n 30 $O/tests.log
