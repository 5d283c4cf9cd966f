cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02d/trace -o trace -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/r02d/trace.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02d/trace/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    n=r['Name'].split('(')[0].replace('salva::','')[:50]
    print(f"{n:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
PY
