set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "" p2w8 p2w8n ""; do
  for st in 8 40; do
    echo "variant=[$v] steps=$st"; SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps $st --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -1
  done
done
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py -q -m gpu -x 2>&1 | tail -5
