# round 6, the very last session: one comment changed in a hashed source after final_a/b/c — the suite from a cold process, the
# rocprofv3 summaries and the bench lines once more on the tree that is committed -> gpurun_out/r06_final, r06_cfg*, r06_8m
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
timeout 2000 python -m pytest tests -q -m gpu > $O/tests_cold_2.log 2>&1; grep -E "passed|failed" $O/tests_cold_2.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/r06/final_b.sh > $O/final_b.log 2>&1; tail -2 $O/final_b.log
for d in r06_cfg2 r06_cfg3 r06_cfg4 r06_8m; do rm -rf profiles/$d; cp -r gpurun_out/$d profiles/$d; done   # (bench.py reads profiles/: on this box only)
bash tools/r06/final_c.sh
