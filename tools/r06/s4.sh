# round 6, session 4: chained steps + the pre-enqueued grid — parity suites, then the step time against a slowed-down host:
# both on / pre-grid off / both off, in the same session
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r06_s4}
mkdir -p $O
gcc -O2 -shared -fPIC -o tools/r06/slow_host.so tools/r06/slow_host.c -ldl
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_speculation_gpu.py tests/test_fuzz_gpu.py tests/test_coupling_gpu.py tests/test_queries_gpu.py tests/test_custom_force_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.log
for rep in 1 2; do
for mode in both nopre none; do
for us in 0 5 10; do
  echo "mode=$mode SLOW_HOST_US=$us $( [ $mode = none ] && export SALVA_HIP_NO_CHAIN=1 SALVA_HIP_NO_PREGRID=1; [ $mode = nopre ] && export SALVA_HIP_NO_PREGRID=1; SLOW_HOST_US=$us LD_PRELOAD=$PWD/tools/r06/slow_host.so HH_ROLE=child python tools/r06/hostile_host.py 2>&1 | grep '^HH')"
done; done; done | tee $O/slow_host.log
