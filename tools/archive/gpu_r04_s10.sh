set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
timeout 900 python tools/pcie_probe.py > $O/pcie.log 2>&1
timeout 600 python tools/big_probe.py 200 > $O/big200.log 2>&1
timeout 600 python bench.py --side 200 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_8m_5_20.json 2> $O/bench_8m.err
cat $O/pcie.log $O/big200.log; tail -c 1500 $O/bench_8m_5_20.json
