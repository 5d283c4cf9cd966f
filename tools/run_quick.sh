cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/quick_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/quick_bench20.json 2> gpurun_out/quick_bench.err
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/quick_bench50.json 2>> gpurun_out/quick_bench.err
SALVA_HIP_TILE_TIMING=1 timeout 300 python bench.py --steps 8 --warmup 0 --no-cpu-baseline 2>&1 | grep "tile timing" > gpurun_out/quick_timing.log
