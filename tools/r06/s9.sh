cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r06_s9}; mkdir -p $O
for mode in wide nowide; do
  echo "== $mode"
  ( [ $mode = nowide ] && export SALVA_HIP_NO_WIDE=1; STEPS=${STEPS:-400} bash tools/r06/soak.sh ) 2>&1 | tee $O/soak_$mode.log
done
