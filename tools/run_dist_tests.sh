timeout 900 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/dist_test.log
timeout 600 python bench.py --side 40 --steps 10 --warmup 2 --force-slabs --no-cpu-baseline > gpurun_out/bench_forceslab.log 2>&1
timeout 600 python bench.py --side 40 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1
