set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
for v in "" ds2112 ds2208; do
SALVA_HIP_LIB_VARIANT=$v timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " >> $O/ab.log
done
SALVA_HIP_LIB_VARIANT=ds2208 timeout 300 python tools/ab_probe.py --steps 60 2>&1 | grep "^AB " >> $O/ab.log
cat $O/ab.log
