import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, scenes
R, DT, G = 0.025, 1.0 / 200.0, (0.0, -9.81, 0.0)
gap = float(os.environ.get("GAP", "0"))
fluid = scenes.jitter(scenes.cube_fluid_positions(100, 100, 100, R), 0.1 * R, seed=42)
cx, cz = np.median(fluid[:, 0]), np.median(fluid[:, 2])
w = LiquidWorld(DFSPHSolver(), R, 2.0)
for (sx, sz), rho0 in zip(((0, 0), (1, 0), (0, 1), (1, 1)), (1000.0, 800.0, 600.0, 400.0)):
    sel = ((fluid[:, 0] > cx) == bool(sx)) & ((fluid[:, 2] > cz) == bool(sz))
    p = np.ascontiguousarray(fluid[sel]); p[:, 0] += np.float32((sx - 0.5) * gap); p[:, 2] += np.float32((sz - 0.5) * gap)
    f = Fluid(p, R, rho0); f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0)); w.add_fluid(f)
ms = []
for k in range(10):
    t0 = time.perf_counter(); st = w.step(DT, G); ms.append((time.perf_counter() - t0) * 1e3)
print("gap %.2f: %.3f ms/step, k_nbr_tile %.1f us, k_pred_density %.1f us, k_divergence_apply %.1f us" % (gap, float(np.mean(ms[4:])), w.time_kernel(4, 20), w.time_kernel(0, 20), w.time_kernel(6, 20)))
