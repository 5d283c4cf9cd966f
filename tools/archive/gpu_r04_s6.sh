set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
SALVA_HIP_TILE_TRACE=1 timeout 300 python tools/ab_probe.py --steps 25 > $O/ab_trace25.log 2>&1
for pad in 0 512 1024 1536 2048 2560 3072 4096; do
SALVA_HIP_P3_PAD=$pad timeout 300 python tools/ab_probe.py --steps 25 2>&1 | grep "^AB " | sed "s/^/pad $pad /" >> $O/ab_pad.log
done
SALVA_HIP_TILE_TRACE=1 timeout 300 python tools/ab_probe.py --steps 60 > $O/ab_trace60.log 2>&1
cat $O/ab_pad.log; grep "^AB" $O/ab_trace25.log $O/ab_trace60.log; grep "salva_hip tiles" $O/ab_trace25.log | tail -n 2; grep "salva_hip tiles" $O/ab_trace60.log | tail -n 2
