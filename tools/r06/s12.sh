# 1000 steps of the bench scene: two launch classes (default) against one launch per pass (SALVA_HIP_NO_CLASSES=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r06_s12}; mkdir -p $O
for mode in classes noclasses; do
  echo "== $mode"
  ( [ $mode = noclasses ] && export SALVA_HIP_NO_CLASSES=1; STEPS=1000 SALVA_HIP_TILE_TRACE=${TRACE:-} bash tools/r06/soak.sh ) 2>&1 | grep -v "salva_hip tiles" | tee $O/soak_$mode.log
done
SALVA_HIP_TILE_TRACE=1 STEPS=1000 bash tools/r06/soak.sh 2>&1 | grep "salva_hip tiles" | awk 'NR%100==0' | cut -c1-130 | tee $O/tiles_trace.log
