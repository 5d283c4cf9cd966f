# round 6, session 3: chained steps (device_types.h StepCtx::gate) — parity suites, then the step time against a slowed-down host,
# chained against SALVA_HIP_NO_CHAIN=1, in the same session
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r06_s3}
mkdir -p $O
gcc -O2 -shared -fPIC -o tools/r06/slow_host.so tools/r06/slow_host.c -ldl
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_speculation_gpu.py tests/test_fuzz_gpu.py tests/test_coupling_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.log
for rep in 1 2; do
for nc in 0 1; do
for us in 0 5 10; do
  echo "NO_CHAIN=$nc SLOW_HOST_US=$us $( [ $nc = 1 ] && export SALVA_HIP_NO_CHAIN=1; SLOW_HOST_US=$us LD_PRELOAD=$PWD/tools/r06/slow_host.so HH_ROLE=child python tools/r06/hostile_host.py 2>&1 | grep '^HH')"
done; done; done | tee $O/slow_host.log
