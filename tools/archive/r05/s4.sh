set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_s4; mkdir -p $O
timeout 900 python -m pytest tests/test_cfl_gpu.py tests/test_queries_gpu.py tests/test_parity_gpu.py tests/test_coupling_gpu.py tests/test_kernels_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -n 12 $O/tests.log | cut -c1-400
for v in "" t6; do
  for st in 8 25 40; do
    echo "variant=[$v] steps=$st"; SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps $st --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -2
  done
done
SALVA_HIP_LIB_VARIANT=t6 timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep -v AB-progress | tail -2
timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep -v AB-progress | tail -2
SALVA_HIP_LIB_VARIANT=t6 timeout 300 python tools/ab_probe.py --side 200 --steps 8 --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -2
timeout 300 python tools/ab_probe.py --side 200 --steps 8 --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -2
