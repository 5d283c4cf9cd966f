#!/bin/bash
# tools/profile_r06.sh TAG [bench args...] — rocprofv3 recipes of round 6 (run on the GPU box via gpurun):
#   1. kernel trace + stats of `bench.py --steps 20 --warmup 5 ARGS`                 -> gpurun_out/TAG/trace_kernel_stats.csv
#   2. separate PMC passes (never combined with a trace domain), neighbour kernels only:
#      SQ busy / LDS counters, FETCH_SIZE, WRITE_SIZE                                 -> gpurun_out/TAG/pmc*/ -> pmc_summary.csv, hbm_traffic.json
#   3. the bench line itself (un-profiled)                                            -> gpurun_out/TAG/bench.json
TAG=${1:-r06}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-download-leg --no-big-leg $@"
cd $R && timeout 900 python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/trace_kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
KRE='k_pred_density|k_divergence|k_pressure_apply|k_nbr_tile|k_density_alpha|k_xsph|k_iisph|k_akinci|k_tile|k_reorder'
PARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-download-leg --no-big-leg $@"
i=0
for PMC in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-include-regex "$KRE" --output-format csv -d $OUT/pmc$i -o pmc$i -- python $R/bench.py $PARGS > $OUT/pmc$i.log 2>&1
done
cd $R
python tools/summarize_pmc.py $OUT > $OUT/hbm_traffic.txt 2>&1
# raw per-dispatch counter files are large: keep the summaries
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
python - <<PY
import csv, json
rows=list(csv.DictReader(open('$OUT/trace_kernel_stats.csv')))
print('$TAG', open('$OUT/bench.json').read().strip().splitlines()[-1][:200])
for r in rows[:14]:
    n=r['Name'].split('(')[0].replace('salva::','')[:48]
    print(f"  {n:48s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
print(open('$OUT/hbm_traffic.txt').read())
PY
du -sh $OUT
