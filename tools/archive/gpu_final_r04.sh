#!/bin/bash
# end-of-round session on the GPU box (one gpurun call): the bench lines (driver protocol 5 + 20 with the CPU leg, survey protocol
# 5 + 50, the other configurations, 8 x 10^6 particles, the decomposed path with one rank and with two ranks sharing the GPU), the
# smoke entry, the PCIe probe, then the rocprofv3 summaries of the three single-GPU configurations and of the 8M-particle scene
# (tools/profile_r04.sh), and the device timeline report.  (The whole GPU suite runs in its own call: tools/gpu_r04_s18.sh.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final
mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_5_20.json 2> $OUT/bench_5_20.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_5_50.json 2> $OUT/bench_5_50.err
timeout 600 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_5_20.json 2> $OUT/bench_cfg3.err
timeout 600 python bench.py --config 3 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_5_50.json 2>> $OUT/bench_cfg3.err
timeout 600 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4_5_20.json 2> $OUT/bench_cfg4.err
timeout 600 python bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4_5_50.json 2>> $OUT/bench_cfg4.err
timeout 600 python bench.py --side 200 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_8m_5_20.json 2> $OUT/bench_8m.err
timeout 600 python bench.py --steps 20 --warmup 5 --force-slabs --no-cpu-baseline 2> $OUT/bench_force_slabs.err | grep "^{" > $OUT/bench_force_slabs_5_20.json
timeout 600 python bench.py --steps 20 --warmup 5 --force-slabs --transport peer --no-cpu-baseline 2> $OUT/bench_force_slabs_peer.err | grep "^{" > $OUT/bench_force_slabs_peer_5_20.json
timeout 600 python bench.py --gpus 2 --transport peer --share-devices --steps 20 --warmup 5 2> $OUT/bench_2ranks_shared.err | grep "^{" > $OUT/bench_2ranks_shared_5_20.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python tools/pcie_probe.py > $OUT/pcie.log 2>&1; cat $OUT/pcie.log
timeout 600 python tools/big_probe.py 200 300 > $OUT/big_probe.log 2>&1; cat $OUT/big_probe.log
for f in bench_5_20 bench_5_50 bench_cfg3_5_20 bench_cfg3_5_50 bench_cfg4_5_20 bench_cfg4_5_50 bench_8m_5_20 bench_force_slabs_5_20 bench_force_slabs_peer_5_20 bench_2ranks_shared_5_20; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$OUT/$f.json').read().strip().splitlines() if l.startswith('{')][-1])
    print('$f', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'us %.1f'%d['roofline']['kernel_us'], d.get('regimes'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
if [ -z "$SKIP_PROFILES" ]; then
bash tools/profile_r04.sh r04_cfg2 2>&1 | tail -25
bash tools/profile_r04.sh r04_cfg3 --config 3 2>&1 | tail -22
bash tools/profile_r04.sh r04_cfg4 --config 4 2>&1 | tail -22
bash tools/profile_r04.sh r04_8m --side 200 2>&1 | tail -22
fi
STEPS=25 bash tools/gap_trace.sh > /dev/null 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 3 8 > $OUT/gap_report_free_fall.txt 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv > $OUT/gap_report_last_steps.txt 2>&1
head -4 $OUT/gap_report_free_fall.txt
