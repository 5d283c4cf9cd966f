"""Tall column whose floor leaks (as in the reference): how does the step cost grow as a few particles fall away?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, scenes
fl, sh = scenes.tank(16, 100, 16, bench.R)
w = LiquidWorld(DFSPHSolver(), bench.R, 2.0)
w.counters.enable()  # (step_ms / grid_ms come from the stage timers, off by default)
f = Fluid(scenes.jitter(fl, 0.1 * bench.R, 42), bench.R, 1000.0)
f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
w.add_fluid(f); w.add_boundary(Boundary(sh))
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1200):
    st = w.step(bench.DT, bench.GRAVITY)
    if (k + 1) % 200 == 0:
        p = f.positions
        ext = (p.max(0) - p.min(0)) / 0.1
        print(f"step {k+1}: ms {st.step_ms:.2f} (grid {st.grid_ms:.2f} solver {st.solver_ms:.2f}) cells bbox {ext.astype(int).tolist()} below floor {(p[:,1] < sh[:,1].min()-0.01).sum()} dev MB {w.device_bytes()/1e6:.0f}", flush=True)
