set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
STEPS=25 bash tools/gap_trace.sh > /dev/null 2>&1
python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv 3 8 > $O/gap_report_free_fall.txt 2>&1
python tools/gap_tsv_report.py gpurun_out/gaps/kernels.tsv > $O/gap_report_last_steps.txt 2>&1
head -70 $O/gap_report_free_fall.txt
