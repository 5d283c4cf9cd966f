#!/usr/bin/env python
"""Pin the oracle against real salva: compare a dump written by bench/rust_ref (`--dump DIR`) with the oracle run on the
same scene.  Usage:  python tests/golden/compare_rust_dump.py DIR --side S --steps K --warmup W
(The image this work was done in has no cargo; whoever has runs bench/rust_ref first.)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--side", type=int, default=20)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
a = ap.parse_args()
fluid, shell = bench.build_scene(a.side)
w = O.OracleWorld(bench.R, 2.0, O.DFSPH, threads=os.cpu_count() or 1)
f = w.add_fluid(fluid, 1000.0)
w.add_xsph(f, 0.5, 0.0)
w.add_boundary(shell)
for _ in range(a.warmup + a.steps):
    w.step(bench.DT, bench.GRAVITY)
ref_p = np.fromfile(os.path.join(a.dir, "positions.f32"), np.float32).reshape(-1, 3)
ref_v = np.fromfile(os.path.join(a.dir, "velocities.f32"), np.float32).reshape(-1, 3)
p, v = w.fluid_vec(f, "positions"), w.fluid_vec(f, "velocities")
dp = np.abs(p - ref_p).max() / bench.R
dv = np.abs(v - ref_v).max()
print(f"max |dx| = {dp:.3e} r   max |dv| = {dv:.3e} m/s over {len(p)} particles after {a.warmup + a.steps} steps")
sys.exit(0 if dp < 1e-3 and dv < 1e-2 else 1)
