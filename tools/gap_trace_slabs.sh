#!/bin/bash
# kernel trace with timestamps of the decomposed code path (1-rank RCCL communicator) in free fall
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gaps_slabs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --no-cpu-baseline --force-slabs --steps 6 --warmup 3 > $OUT/trace.log 2>&1
cd $R
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
with open("$OUT/kernels.tsv","w") as o:
    for r in rows:
        o.write(f"{r['Kernel_Name'].split('(')[0][:60]}\t{int(r['Start_Timestamp'])-t0}\t{int(r['End_Timestamp'])-t0}\n")
print(len(rows))
PY
rm -rf $OUT/trace
