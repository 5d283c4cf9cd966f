"""Where does the asynchronous read-back's 0.28 ms per step go?  Per-step times of the bench scene resident and with
positions + velocities read back after every step (bench.py's `with_download` leg), per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, scenes

R, DT, G = 0.025, 1.0 / 200.0, (0.0, -9.81, 0.0)
fluid, shell = scenes.tank(100, 100, 100, R)
fluid = scenes.jitter(fluid, 0.1 * R, seed=42)


def run(download, steps=20):
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    f = Fluid(fluid, R, 1000.0)
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    h = w.add_fluid(f)
    w.add_boundary(Boundary(shell))
    for _ in range(5):
        w.step(DT, G)
    ms, t_wait, t_enq = [], [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        w.step(DT, G)
        t1 = time.perf_counter()
        if download:
            w.wait_download()
            t2 = time.perf_counter()
            w.download_async(h)
            t3 = time.perf_counter()
            t_wait.append((t2 - t1) * 1e3); t_enq.append((t3 - t2) * 1e3)
        ms.append((time.perf_counter() - t0) * 1e3)
    w.wait_download()
    return np.array(ms), np.array(t_wait), np.array(t_enq)

a, _, _ = run(False)
b, tw, te = run(True)
print("env", {k: v for k, v in os.environ.items() if k.startswith("HSA_") or k.startswith("SALVA_HIP_DL")})
print("resident  per step ms:", np.round(a, 3).tolist(), "mean %.3f" % a.mean())
print("download  per step ms:", np.round(b, 3).tolist(), "mean %.3f" % b.mean())
print("  of which wait_download:", np.round(tw, 3).tolist())
print("  of which download_async (enqueue):", np.round(te, 3).tolist())
print("  step() itself with a copy in flight:", np.round(b - tw - te, 3).tolist())
