#!/bin/bash
# kernel trace with timestamps of a short bench run: how long is the GPU idle between dependent kernels?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gaps
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_abi.py -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --no-cpu-baseline --steps ${STEPS:-25} --warmup 5 > $OUT/trace.log 2>&1
cd $R
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(len(rows), rows[0].keys())
# keep a compact version: name, start, end (ns relative)
t0=int(rows[0]['Start_Timestamp'])
with open("$OUT/kernels.tsv","w") as o:
    for r in rows:
        o.write(f"{r['Kernel_Name'].split('(')[0][:60]}\t{int(r['Start_Timestamp'])-t0}\t{int(r['End_Timestamp'])-t0}\n")
PY
rm -rf $OUT/trace
du -sh $OUT
