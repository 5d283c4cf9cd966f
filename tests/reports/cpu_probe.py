"""How many host threads does the oracle want on this box?  (run via gpurun; informs bench.py's cpu_baseline)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
import bench
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
print("loadavg", open("/proc/loadavg").read().strip())
side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fluid, shell = bench.build_scene(side)
for T in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,32,64,128,256").split(",")]:
    w = O.OracleWorld(bench.R, 2.0, O.DFSPH, threads=T)
    fid = w.add_fluid(fluid, 1000.0); w.add_xsph(fid, 0.5, 0.0); w.add_boundary(shell)
    w.step(bench.DT, bench.GRAVITY)
    t0 = time.perf_counter(); n = 2
    for _ in range(n): st = w.step(bench.DT, bench.GRAVITY)
    dt = (time.perf_counter() - t0) / n
    print(f"side {side} threads {T}: {dt:.3f} s/step  {len(fluid)/dt/1e3:.1f} k particle-steps/s  n_div {st.n_div_iters}", flush=True)
    del w
