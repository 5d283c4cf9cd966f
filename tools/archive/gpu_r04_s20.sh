set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s20; mkdir -p $O
for rep in 1 2; do
timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep "^AB " >> $O/ab.log
SALVA_HIP_LIB_VARIANT=np5 timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep "^AB " >> $O/ab.log
SALVA_HIP_NP_LEGACY=1 timeout 300 python tools/ab_probe.py --config 3 --steps 25 2>&1 | grep "^AB " | sed "s/^/nplegacy /" >> $O/ab.log
done
cat $O/ab.log
