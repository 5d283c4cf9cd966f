set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
timeout 600 python -m pytest tests/test_download_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " >> $O/ab.log
timeout 300 python tools/ab_probe.py --steps 60 2>&1 | grep "^AB " >> $O/ab.log
timeout 600 python tools/pcie_probe.py > $O/pcie.log 2>&1
cat $O/ab.log $O/pcie.log $O/rc.log; tail -n 5 $O/tests.log
