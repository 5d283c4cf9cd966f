#!/bin/bash
# Device timeline of a 1000-step run of config 2: the kernels of steps 300..400 (sloshing, few strays) beside those of steps 900..1000
# (3 400 leaked particles with a tile each) — what the strays' tiles cost, kernel by kernel (tools/gap_tsv_report.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_late; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --no-cpu-baseline --no-download-leg --no-big-leg --steps 1000 --warmup 5 > $OUT/trace.log 2>&1
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
PY
python tools/gap_tsv_report.py $OUT/kernels.tsv 305 405 > $OUT/steps_300_400.txt 2>&1
python tools/gap_tsv_report.py $OUT/kernels.tsv 905 1005 > $OUT/steps_900_1000.txt 2>&1
rm -rf $OUT/trace $OUT/kernels.tsv
head -3 $OUT/steps_300_400.txt; head -3 $OUT/steps_900_1000.txt
