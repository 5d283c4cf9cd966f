"""Where the time of a free-fall step goes on the host: the Python mirror's `step` against the raw C call, each on a fresh world over the
same steps (6..13 of the bench scene: one divergence and one pressure iteration each, ~0.63 ms)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from salva_amd import _lib as L  # noqa: E402

fluid, shell = bench.build_scene(100)
N = 8


def run(raw):
    w, f = bench.make_world(fluid, shell, 0)
    for _ in range(5):
        w.step(bench.DT, bench.GRAVITY)
    g = (C.c_float * 3)(*bench.GRAVITY)
    st = L.StepStats()
    ts = []
    for _ in range(N):
        t0 = time.perf_counter()
        if raw:
            L.check(w._L.salva_hip_step(w._h, bench.DT, g, C.byref(st)))
        else:
            w.step(bench.DT, bench.GRAVITY)
        ts.append(time.perf_counter() - t0)
    return np.asarray(ts) * 1e3


for rep in range(2):
    py, raw = run(False), run(True)
    print(f"rep {rep}: python mirror step {py.mean():.4f} ms (min {py.min():.4f}) | raw C call {raw.mean():.4f} ms (min {raw.min():.4f}) | "
          f"mirror overhead {1e3 * (py.mean() - raw.mean()):.1f} us per step", flush=True)
