#!/bin/bash
# round 4, session 32: sweeps on the end-of-round code -> profiles/r04_final (fuzz 100 seeds, mirrors 24 seeds, soak, loopback scale)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r04_final
mkdir -p $O
timeout 900 python tools/fuzz_sweep.py > $O/fuzz_sweep.log 2>&1; tail -n 2 $O/fuzz_sweep.log
timeout 900 python tools/mirror_sweep.py > $O/mirror_sweep.log 2>&1; tail -n 2 $O/mirror_sweep.log
timeout 900 python tools/soak.py > $O/soak.log 2>&1; tail -n 6 $O/soak.log
timeout 600 python tools/loopback_scale.py > $O/loopback_scale.log 2>&1; tail -n 6 $O/loopback_scale.log
