"""Round 6: 10^6 particles in four fluids of different density0 (2 x 2 columns in a tank, BASELINE config 4 with four masses instead of
two): ms per step over STEPS steps.  Run once as is and once with SALVA_HIP_NO_TWO_MASS=1 (the general kernels)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, scenes
R, DT, G = 0.025, 1.0 / 200.0, (0.0, -9.81, 0.0)
steps = int(os.environ.get("STEPS", "40"))
side = int(os.environ.get("SIDE", "100"))
fluid, shell = scenes.tank(side, side, side, R)
fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
cx, cz = np.median(fluid[:, 0]), np.median(fluid[:, 2])
w = LiquidWorld(DFSPHSolver(), R, 2.0)
fls = []
for (sx, sz), rho0 in zip(((0, 0), (1, 0), (0, 1), (1, 1)), (1000.0, 800.0, 600.0, 400.0)):
    sel = ((fluid[:, 0] > cx) == bool(sx)) & ((fluid[:, 2] > cz) == bool(sz))
    f = Fluid(np.ascontiguousarray(fluid[sel]), R, rho0)
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    fls.append(w.add_fluid(f))
w.add_boundary(Boundary(shell))
ms, it = [], []
for k in range(steps):
    t0 = time.perf_counter(); st = w.step(DT, G); ms.append((time.perf_counter() - t0) * 1e3)
    it.append((st.n_divergence_iters, st.n_pressure_iters))
print("steps 5..%d: %.3f ms/step; iterations %s; contacts %d" % (steps - 1, float(np.mean(ms[5:])), it[5::8], int(st.ncontacts)))
for kid, name in ((1, "k_divergence"), (6, "k_divergence_apply"), (0, "k_pred_density"), (4, "k_nbr_tile")):
    print(name, "%.1f us" % w.time_kernel(kid, 20))
