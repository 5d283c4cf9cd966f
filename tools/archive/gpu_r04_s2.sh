set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_kernels_gpu.py -x -q > $O/tests_parity.log 2>&1; echo "rc parity $?" >> $O/rc.log
for rep in 1 2; do
timeout 300 python tools/ab_probe.py --steps 25 >> $O/ab_p3.log 2>&1
SALVA_HIP_NO_PLANES=1 timeout 300 python tools/ab_probe.py --steps 25 >> $O/ab_legacy.log 2>&1
done
timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_p3.log 2>&1
SALVA_HIP_NO_PLANES=1 timeout 300 python tools/ab_probe.py --steps 60 >> $O/ab_legacy.log 2>&1
timeout 600 python -m pytest tests/test_config5_gpu.py tests/test_fuzz_gpu.py -x -q > $O/tests_c5_fuzz.log 2>&1; echo "rc c5fuzz $?" >> $O/rc.log
grep "^AB " $O/ab_p3.log $O/ab_legacy.log
cat $O/rc.log
