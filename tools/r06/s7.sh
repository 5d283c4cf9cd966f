# A/B of the product library against libsalva_hip_prev.so: steps (hostile_host child) and kernels (ab_probe), three rounds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r06_s7}
mkdir -p $O
for rep in 1 2 3; do
for v in prev ""; do
  echo "lib=[$v] $(SALVA_HIP_LIB_VARIANT=$v HH_ROLE=child python tools/r06/hostile_host.py 2>&1 | grep '^HH')"
done; done | tee $O/ab_steps.log
for v in prev "" prev ""; do
  SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 25 --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -1
done | tee $O/ab_kernels.log
