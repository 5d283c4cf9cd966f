#!/bin/bash
# round 4, session 28: kernel trace of the grid phase, counting sort against radix sort
cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out/s28
mkdir -p $O
cd /tmp
for mode in counting radix; do
  if [ $mode = radix ]; then export SALVA_HIP_RADIX_SORT=1; else unset SALVA_HIP_RADIX_SORT; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o t -- python /root/repo/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/$mode.log 2>&1
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1); cp $f $O/${mode}_kernel_stats.csv
  find $O/$mode -name "*kernel_trace.csv" -delete
done
cd /root/repo
python - <<'PY'
import csv
for mode in ('counting','radix'):
    rows=list(csv.DictReader(open('gpurun_out/s28/%s_kernel_stats.csv'%mode)))
    print(mode)
    for r in rows:
        n=r['Name'].replace('salva::','').replace('void ','')[:70]
        if any(k in n for k in ('cell','rocprim','fill','Fill','tile_','reorder','scan','lookback')):
            print('  %-70s calls %4s avg %8.1f us total %8.1f' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
