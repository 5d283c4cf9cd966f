"""Where the time between two steps goes: Python mirror vs the C call vs the GPU-timed part of the step (1M-particle bench scene)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from salva_amd import _lib as L  # noqa: E402

fluid, shell = bench.build_scene(100)
w, f = bench.make_world(fluid, shell, 0)
w.counters.enable()  # (step_ms / grid_ms come from the stage timers, off by default)
for _ in range(5):
    w.step(bench.DT, bench.GRAVITY)
N = 20
t_py, t_c, t_gpu = [], [], []
g = (C.c_float * 3)(*bench.GRAVITY)
for _ in range(N):
    t0 = time.perf_counter()
    st = w.step(bench.DT, bench.GRAVITY)
    t_py.append(time.perf_counter() - t0)
    t_gpu.append(st.step_ms)
for _ in range(N):
    st = L.StepStats()
    t0 = time.perf_counter()
    L.check(w._L.salva_hip_step(w._h, bench.DT, g, C.byref(st)))
    t_c.append(time.perf_counter() - t0)
    t_gpu.append(st.step_ms)
print(f"python step {np.mean(t_py) * 1e3:.3f} ms | raw C call {np.mean(t_c) * 1e3:.3f} ms | GPU-timed (events) first loop {np.mean(t_gpu[:N]):.3f} ms, second {np.mean(t_gpu[N:]):.3f} ms")
