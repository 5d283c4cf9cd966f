#!/bin/bash
# round 3, second GPU pass: where does k_pred_density's time go (tile phase stamps, old vs new), 2x4x4 tiles, k_nbr_tile variant 2,
# and the new full-size tests.  Logs under gpurun_out/r03b/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
export AB_PROBE_WATCHDOG=100
timeout 150 python tools/ab_probe.py --steps 25 --save /tmp/new25.npy > $O/ab_new.log 2>&1
SALVA_HIP_NBR_VARIANT=2 timeout 150 python tools/ab_probe.py --steps 25 --ref /tmp/new25.npy > $O/ab_nbr2.log 2>&1
SALVA_HIP_LIB_VARIANT=t2 timeout 150 python tools/ab_probe.py --steps 25 --ref /tmp/new25.npy > $O/ab_t2.log 2>&1
SALVA_HIP_LIB_VARIANT=diag SALVA_HIP_TILE_TIMING=1 timeout 150 python tools/ab_probe.py --steps 25 --kernels 0 --ref /tmp/new25.npy > $O/ab_diag.log 2>&1
SALVA_HIP_LIB_VARIANT=r02 SALVA_HIP_TILE_TIMING=1 timeout 150 python tools/ab_probe.py --steps 25 --kernels 0 --ref /tmp/new25.npy > $O/ab_r02.log 2>&1
SALVA_HIP_TILE_THREADS=640 timeout 150 python tools/ab_probe.py --steps 25 --kernels 0,1 --ref /tmp/new25.npy > $O/ab_w10.log 2>&1
grep -hE "^AB |tile timing|Error|error|Traceback|File " $O/ab_*.log | cut -c1-400
timeout 900 python -m pytest -q -x tests/test_config5_gpu.py "tests/test_dist_gpu.py::test_rebalance_moves_cuts_by_whole_slabs_between_ranks_of_different_extent" "tests/test_dist_gpu.py::test_rebalance_recuts_the_slabs_and_keeps_the_physics" "tests/test_dist_gpu.py::test_slabs_match_single_domain" -s > $O/tests_new.log 2>&1; tail -25 $O/tests_new.log | cut -c1-300
timeout 600 python -m pytest -q -x "tests/test_full_size_gpu.py::test_config3_iisph_akinci_1m_tank_30_steps" -s > $O/tests_cfg3.log 2>&1; tail -12 $O/tests_cfg3.log | cut -c1-600
