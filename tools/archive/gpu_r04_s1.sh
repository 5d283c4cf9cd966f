set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s1
timeout 600 python -m pytest tests/test_host_shape_gpu.py tests/test_numpy_reading_gpu.py -x -q > gpurun_out/s1/tests_a.log 2>&1; echo "rc tests_a $?" >> gpurun_out/s1/rc.log
SALVA_CONFIG5_SIDE=40 timeout 600 python -m pytest tests/test_config5_processes_gpu.py -x -q -s > gpurun_out/s1/config5_side40.log 2>&1; echo "rc c5_40 $?" >> gpurun_out/s1/rc.log
timeout 400 python bench.py --gpus 2 --transport peer --share-devices --steps 20 --warmup 5 > gpurun_out/s1/bench_2ranks_shared.log 2>&1; echo "rc bench2 $?" >> gpurun_out/s1/rc.log
timeout 100 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/s1/bench_2ranks_refused.log 2>&1; echo "rc bench2refused $?" >> gpurun_out/s1/rc.log
timeout 300 python tools/ab_probe.py --steps 25 > gpurun_out/s1/ab_base.log 2>&1; echo "rc ab $?" >> gpurun_out/s1/rc.log
timeout 300 python tools/ab_probe.py --steps 60 > gpurun_out/s1/ab_base60.log 2>&1; echo "rc ab60 $?" >> gpurun_out/s1/rc.log
timeout 900 python -m pytest tests/test_config5_processes_gpu.py -x -q -s > gpurun_out/s1/config5_full.log 2>&1; echo "rc c5_full $?" >> gpurun_out/s1/rc.log
cat gpurun_out/s1/rc.log
