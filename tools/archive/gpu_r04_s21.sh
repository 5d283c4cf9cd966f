set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s21; mkdir -p $O
for rep in 1 2; do
timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " >> $O/ab.log
SALVA_HIP_NO_SPLIT_SUM=1 timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " | sed "s/^/nosplit /" >> $O/ab.log
done
timeout 300 python tools/ab_probe.py --steps 60 2>&1 | grep "^AB " >> $O/ab.log
SALVA_HIP_NO_SPLIT_SUM=1 timeout 300 python tools/ab_probe.py --steps 60 2>&1 | grep "^AB " | sed "s/^/nosplit /" >> $O/ab.log
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_config5_gpu.py -x -q > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
cat $O/ab.log $O/rc.log; grep -E "passed|failed|^E " $O/tests.log | tail -n 5
