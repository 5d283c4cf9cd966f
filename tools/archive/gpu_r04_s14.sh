set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s14; mkdir -p $O
timeout 1200 python -m pytest tests/test_dist_gpu.py tests/test_speculation_gpu.py tests/test_custom_force_gpu.py tests/test_config5_gpu.py tests/test_peer_transport_gpu.py -x -q > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
cat $O/rc.log; grep -E "passed|failed|Error|^E " $O/tests.log | tail -n 12
