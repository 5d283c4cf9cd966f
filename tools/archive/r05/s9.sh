cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for v in prev "" ; do
  for st in 8 25 40; do
    echo "variant=[$v] steps=$st"; SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps $st --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -1
  done
done
done
for v in prev ""; do SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --side 200 --steps 8 --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -1; done
