# end-of-round session B: rocprofv3 summaries of the three single-GPU configurations and of the 8M-particle scene
# (tools/profile_r05.sh: kernel trace + stats, three separate PMC passes), and the device timeline report.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
bash tools/profile_r05.sh r05_cfg2 2>&1 | tail -25
bash tools/profile_r05.sh r05_cfg3 --config 3 2>&1 | tail -22
bash tools/profile_r05.sh r05_cfg4 --config 4 2>&1 | tail -22
bash tools/profile_r05.sh r05_8m --side 200 2>&1 | tail -22
OUT=$R/gpurun_out/r05_final; mkdir -p $OUT
STEPS=25 bash tools/gap_trace.sh > /dev/null 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 3 8 > $OUT/gap_report_free_fall.txt 2>&1; python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv > $OUT/gap_report_last_steps.txt 2>&1
head -4 $OUT/gap_report_free_fall.txt
