# round 6, session 6: the done-word behind the set-up loads, one k_init_ctl per step, list statistics and box fold inside the
# end-of-step publication — against the library of the commit before (libsalva_hip_prev.so), same session
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r06_s6}
mkdir -p $O
timeout 1500 python -m pytest tests/test_chain_gpu.py tests/test_parity_gpu.py tests/test_speculation_gpu.py tests/test_fuzz_gpu.py tests/test_coupling_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/tests.log
for rep in 1 2 3; do
for v in prev ""; do
  echo "lib=[$v] $(SALVA_HIP_LIB_VARIANT=$v HH_ROLE=child python tools/r06/hostile_host.py 2>&1 | grep '^HH')"
done; done | tee $O/ab_steps.log
for v in prev "" prev ""; do
  SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 25 --kernels 0,1,6,4 2>&1 | grep -v AB-progress | tail -1
done | tee $O/ab_kernels.log
