# long runs of the bench scene (config 2, 10^6 particles): ms per step and iterations per window of 100 steps, tile statistics every
# 100 steps.  LIB=variant STEPS=n; env switches pass through.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, scenes
R, DT, G = 0.025, 1.0 / 200.0, (0.0, -9.81, 0.0)
steps = int(os.environ.get("STEPS", "1000"))
fluid, shell = scenes.tank(100, 100, 100, R)
fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
w = LiquidWorld(DFSPHSolver(), R, 2.0)
f = Fluid(fluid, R, 1000.0); f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
w.add_fluid(f); w.add_boundary(Boundary(shell))
ms, it, halo = [], [], []
for k in range(steps):
    t0 = time.perf_counter(); st = w.step(DT, G); ms.append((time.perf_counter() - t0) * 1e3)
    it.append(st.n_divergence_iters); halo.append(int(st.reserved[0]))
    if (k + 1) % 100 == 0:
        a = slice(k - 99, k + 1)
        print(f"steps {k-99:4d}..{k:4d}: {np.mean(ms[a]):.3f} ms/step, div iters {np.mean(it[a]):.1f}, max halo {min(halo[a])}..{max(halo[a])}, threads {int(st.reserved[2])}", flush=True)
print(f"whole {steps}: {np.mean(ms):.3f} ms/step = {1e6 / (np.mean(ms) * 1e-3):.3e} particle-steps/s")
c = w.counters
print("counters: light", c.light_class_passes, "sparse", c.sparse_class_passes, "chained", c.chained_passes, "breaks", c.chain_breaks, "pregrid", c.pregrid_adopted, "dropped", c.pregrid_dropped, "discarded", c.discarded_passes)
for kid, name in ((1, "k_divergence"), (6, "k_divergence_apply"), (0, "k_pred_density"), (4, "k_nbr_tile")):
    print(name, "%.1f us" % w.time_kernel(kid, 20))
PY
