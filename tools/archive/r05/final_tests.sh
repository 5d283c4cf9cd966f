# the whole GPU suite three times in a row from cold processes, on the final code
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_final
mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for k in 1 2 3; do
  timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests_cold_$k.log 2>&1; echo "tests rc=$?" >> $OUT/tests_cold_$k.log
  grep -E "passed|failed" $OUT/tests_cold_$k.log | tail -1; tail -n 1 $OUT/tests_cold_$k.log
done
