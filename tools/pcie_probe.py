"""PCIe-inclusive rates: what the step costs when the host reads positions + velocities back after every step, or
re-uploads them before every step (the boundary hands over host arrays; bench.py's `value` keeps state resident)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
fl, sh = bench.build_scene(100)
w, f = bench.make_world(fl, sh, 0)
for _ in range(5): w.step(bench.DT, bench.GRAVITY)
def run(mode, steps=20):
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step(bench.DT, bench.GRAVITY)
        if mode >= 1:
            p = f.positions; v = f.velocities           # download (lazy: one D2H of each array)
        if mode >= 2:
            f.positions = p; f.velocities = v           # mark dirty: uploaded by the next step
    return (time.perf_counter() - t0) / steps * 1e3
for name, mode in (("resident", 0), ("download pos+vel every step", 1), ("download + re-upload every step", 2)):
    print(f"{name}: {run(mode):.2f} ms/step", flush=True)
