# round 6, session 15b: rocprofv3 kernel statistics of the four-mass probe on the segment kernels (SALVA_HIP_MAX_MASSES=4)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_masses; mkdir -p $O
cd /tmp
STEPS=12 SALVA_HIP_MAX_MASSES=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o four -- python $GRAFT_REPO_ROOT/tools/r06/multi_mass_probe.py > $O/prof.log 2>&1
tail -3 $O/prof.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r06_masses/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open("gpurun_out/r06_masses/four_masses_kernel_stats.txt", "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats, STEPS=12 SALVA_HIP_MAX_MASSES=4 python tools/r06/multi_mass_probe.py (10^6 particles, four masses)\n")
        for r in rows[:24]:
            out.write("%-100s calls %6s avg %10.1f ns  %5s %%\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
    print(open("gpurun_out/r06_masses/four_masses_kernel_stats.txt").read()[:3500])
PY
rm -rf $O/prof
