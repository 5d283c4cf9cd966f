# round 6, session 2: slope of the step time against an artificially slow host (tools/r06/slow_host.c)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r06_s2}
mkdir -p $O
gcc -O2 -shared -fPIC -o tools/r06/slow_host.so tools/r06/slow_host.c -ldl
for us in 0 2 5 10 20; do
  echo "SLOW_HOST_US=$us"
  SLOW_HOST_US=$us LD_PRELOAD=$PWD/tools/r06/slow_host.so HH_ROLE=child python tools/r06/hostile_host.py 2>&1 | grep "^HH"
done | tee $O/slow_host.log
