set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
for v in "" p2even nbrodd; do
SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " >> $O/ab.log
SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 60 2>&1 | grep "^AB " >> $O/ab.log
done
for v in "" fx2448; do
SALVA_HIP_NO_PLANES=1 SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " | sed "s/^/legacy /" >> $O/ab.log
SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep "^AB " >> $O/ab.log
done
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py -x -q > $O/tests_parity.log 2>&1; echo "rc parity $?" >> $O/rc.log
cat $O/ab.log $O/rc.log
