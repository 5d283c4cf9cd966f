"""Long runs: stability of capacities / LDS sizing / iteration control over hundreds of steps (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from salva_amd import Boundary, DFSPHSolver, Fluid, IISPHSolver, LiquidWorld, XSPHViscosity, ArtificialViscosity, scenes

def soak(name, w, f, steps, every):
    t0 = time.perf_counter()
    for k in range(steps):
        st = w.step(bench.DT, bench.GRAVITY)
        if (k + 1) % every == 0:
            p = f.positions
            print(f"{name} step {k+1}: ms {st.step_ms:.2f} n_d {st.n_divergence_iters} n_p {st.n_pressure_iters} halo {int(st.reserved[0])} "
                  f"contacts/particle {st.ncontacts/len(p):.1f} y[min,max] {p[:,1].min():.3f} {p[:,1].max():.3f} finite {np.isfinite(p).all()} "
                  f"dev MB {w.device_bytes()/1e6:.0f}", flush=True)
    print(f"{name}: {steps} steps in {time.perf_counter()-t0:.1f} s", flush=True)

# 1M bench scene, 400 steps
fl, sh = bench.build_scene(100)
w, f = bench.make_world(fl, sh, 0)
w.counters.enable()  # (step_ms / grid_ms come from the stage timers, off by default)
soak("tank1M", w, f, 300, 100)
del w
# dam break: 60x60x40 column in a tank 3x as long, artificial viscosity with boundary term, 1500 steps
fluid, shell = scenes.tank(60, 60, 40, bench.R, wall_cells=120)
w = LiquidWorld(DFSPHSolver(), bench.R, 2.0)
w.counters.enable()  # (step_ms / grid_ms come from the stage timers, off by default)
f = Fluid(scenes.jitter(fluid, 0.05 * bench.R), bench.R, 1000.0)
f.nonpressure_forces.append(ArtificialViscosity(0.5, 0.2))
w.add_fluid(f); w.add_boundary(Boundary(shell))
soak("dambreak144k", w, f, 1250, 250)
del w
# IISPH tank 200k, 600 steps
fluid, shell = scenes.tank(60, 60, 60, bench.R)
w = LiquidWorld(IISPHSolver(), bench.R, 2.0)
w.counters.enable()  # (step_ms / grid_ms come from the stage timers, off by default)
f = Fluid(scenes.jitter(fluid, 0.05 * bench.R), bench.R, 1000.0)
f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
w.add_fluid(f); w.add_boundary(Boundary(shell))
soak("iisph216k", w, f, 300, 100)
