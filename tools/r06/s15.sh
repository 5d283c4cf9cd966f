# round 6, session 15: worlds with four masses on the plane layouts against the general kernels, 10^6 particles; config 4 (two masses)
# before / after (SALVA_HIP_LIB_VARIANT=bug: a library built from commit 12fb8e3, before this change) -> gpurun_out/r06_masses
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r06_masses; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "layout or masses or mass" 2>&1 | grep -E "passed|failed|FAILED" | tee $O/tests.log
for rep in 1 2; do for v in "SALVA_HIP_MAX_MASSES=4" "SALVA_HIP_NO_TWO_MASS=1"; do
  echo "== four masses, 12 steps of free fall [$v]"; env $v STEPS=12 python tools/r06/multi_mass_probe.py
  echo "== four masses, 40 steps (the impact from step 18 on) [$v]"; env $v python tools/r06/multi_mass_probe.py
done; done 2>&1 | tee $O/four_masses.log
for rep in 1 2; do for lib in bug ""; do echo "== config 4, lib [$lib]"; SALVA_HIP_LIB_VARIANT=$lib python bench.py --steps 20 --warmup 5 --config 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f' % d['ms_per_step'], r['kernel'], '%.2f us' % r['kernel_us'], ' '.join('%s %.2f' % (k, v['kernel_us']) for k, v in r['other_kernels'].items()))"; done; done 2>&1 | tee $O/config4.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o four -- env SALVA_HIP_MAX_MASSES=4 python $GRAFT_REPO_ROOT/tools/r06/multi_mass_probe.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r06_masses/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open("gpurun_out/r06_masses/four_masses_kernel_stats.txt", "w") as out:
        for r in rows[:24]:
            out.write("%-90s calls %6s avg %10.1f ns  %5s %%\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
    print(open("gpurun_out/r06_masses/four_masses_kernel_stats.txt").read()[:2500])
PY
rm -rf $O/prof
