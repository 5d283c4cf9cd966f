#!/bin/bash
# A/B of the product library against libsalva_hip_prev.so (a saved earlier build) on configs 2 and 3, then the whole GPU suite
export TMPDIR=/tmp; O=gpurun_out/${TAG:-ab_prev}; mkdir -p $O
for v in prev "" prev ""; do SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=100 timeout 150 python tools/ab_probe.py --steps 25 --reps 20 --kernels 0,1,4 2>&1 | grep "^AB lib" >> $O/ab2.log; done
for v in prev ""; do SALVA_HIP_LIB_VARIANT=$v AB_PROBE_WATCHDOG=100 timeout 150 python tools/ab_probe.py --config 3 --steps 25 --reps 20 --kernels 2,3 2>&1 | grep "^AB lib" >> $O/ab3.log; done
cat $O/ab2.log $O/ab3.log
timeout 1200 python -m pytest tests -q -x -m gpu > $O/tests.log 2>&1; grep -n "passed\|failed\|Error" $O/tests.log | tail -4
