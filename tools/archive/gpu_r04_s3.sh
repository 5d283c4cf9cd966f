set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
for rep in 1 2; do
timeout 300 python tools/ab_probe.py --steps 25 >> $O/ab_nw8.log 2>&1
SALVA_HIP_LIB_VARIANT=nw7 timeout 300 python tools/ab_probe.py --steps 25 >> $O/ab_nw7.log 2>&1
done
timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_nw8.log 2>&1
SALVA_HIP_LIB_VARIANT=nw7 timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_nw7.log 2>&1
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q > $O/tests_parity.log 2>&1; echo "rc parity $?" >> $O/rc.log
grep "^AB " $O/ab_nw8.log $O/ab_nw7.log
cat $O/rc.log
