#!/bin/bash
# tools/profile.sh TAG [bench args...] — rocprofv3 recipes used for profiles/ (run on the GPU box via gpurun).
#   1. kernel trace + stats of a short bench run            -> gpurun_out/TAG/trace
#   2. PMC passes (one counter set per run, own runs, no trace domains) for the neighbour-sum kernels
# Counter sets follow /opt/skills/guides/MI355X_MICROARCH.md §rocprofv3 PMC slots (SQ 8 / TCC 4; FETCH_SIZE and WRITE_SIZE separately).
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline $@"
# the trace is taken over the bench's own default protocol (5 warm-up + 50 timed steps), the PMC passes over a shorter run
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --no-cpu-baseline $@ > $OUT/trace.log 2>&1
KRE='k_pred_density|k_divergence|k_pressure_apply|k_nbr|k_density_alpha|k_xsph|k_tile'
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc$i -o pmc$i -- python $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.csv" | head -40
du -sh $OUT
