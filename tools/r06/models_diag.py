"""How much do the neighbour-sum kernels pay for a second fluid MODEL (same mass: the uniform plane kernels either way)?
A free block of 10^6 particles as one fluid / as two halves of the same density0 / of different density0 (two masses)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from salva_amd import DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, scenes
R, DT, G = 0.025, 1.0 / 200.0, (0.0, -9.81, 0.0)
fluid = scenes.jitter(scenes.cube_fluid_positions(100, 100, 100, R), 0.1 * R, seed=42)
cx = np.median(fluid[:, 0])
for tag, parts in (("one fluid", [(np.ones(len(fluid), bool), 1000.0)]),
                   ("two fluids, one mass", [(fluid[:, 0] <= cx, 1000.0), (fluid[:, 0] > cx, 1000.0)]),
                   ("two fluids, two masses", [(fluid[:, 0] <= cx, 1000.0), (fluid[:, 0] > cx, 500.0)])):
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    for sel, rho0 in parts:
        f = Fluid(np.ascontiguousarray(fluid[sel]), R, rho0); f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0)); w.add_fluid(f)
    ms = []
    for k in range(10):
        t0 = time.perf_counter(); st = w.step(DT, G); ms.append((time.perf_counter() - t0) * 1e3)
    print("%-24s %.3f ms/step, k_nbr_tile %.1f us, k_pred_density %.1f, k_divergence %.1f, k_divergence_apply %.1f" % (tag, float(np.mean(ms[4:])), w.time_kernel(4, 20), w.time_kernel(0, 20), w.time_kernel(1, 20), w.time_kernel(6, 20)))
    del w
