set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_speculation_gpu.py -x -q > $O/tests.log 2>&1; echo "rc tests $?" >> $O/rc.log
for v in "" dad5; do
SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " >> $O/ab.log
SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 60 2>&1 | grep "^AB " >> $O/ab.log
done
SALVA_HIP_NO_FUSED_DIV=1 timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " | sed "s/^/nofuse /" >> $O/ab.log
SALVA_HIP_NO_FUSED_DIV=1 timeout 300 python tools/ab_probe.py --steps 60 2>&1 | grep "^AB " | sed "s/^/nofuse /" >> $O/ab.log
cat $O/ab.log $O/rc.log; grep -E "passed|failed|^E " $O/tests.log | tail -n 5
