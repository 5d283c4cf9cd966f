cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/quick_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/quick_bench20.json 2> gpurun_out/quick_bench.err
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/quick_bench50.json 2>> gpurun_out/quick_bench.err
