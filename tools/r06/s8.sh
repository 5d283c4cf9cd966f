cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r06_s8}; mkdir -p $O
timeout 600 python -m pytest tests/test_split_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.log
for mode in split nosplit; do
  echo "== $mode"
  ( [ $mode = nosplit ] && export SALVA_HIP_NO_SPLIT=1; STEPS=${STEPS:-500} bash tools/r06/soak.sh ) 2>&1 | tee $O/soak_$mode.log
done
