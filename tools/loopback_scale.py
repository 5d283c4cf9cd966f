"""Decomposition overhead at bench size, on ONE GPU: the 10^6-particle bench scene as 1 domain vs 2 / 4 x-slabs driven by
host threads over the loopback transport (same World code RCCL drives).  The slabs share the GPU, so this measures the
extra work and synchronisation the decomposition adds (ghost planes, refreshes, lock-step solves) — not a speed-up."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, dist

H = 0.1
fluid, shell = bench.build_scene(100)
STEPS, WARM = 20, 3

def single():
    w, f = bench.make_world(fluid, shell, 0)
    for _ in range(WARM): w.step(bench.DT, bench.GRAVITY)
    t0 = time.perf_counter()
    its = []
    for _ in range(STEPS):
        st = w.step(bench.DT, bench.GRAVITY); its.append(st.n_divergence_iters)
    return (time.perf_counter() - t0) / STEPS * 1e3, its

def slabs(nr):
    cx = dist.cell_x(fluid, H)
    sl = dist.split_slabs(cx, nr)
    owner = dist.owner_of(cx, sl)
    comms = dist.Comm.loopback(nr)
    out = [None] * nr
    bar = threading.Barrier(nr)
    def main(r):
        mine = np.nonzero(owner == r)[0]
        w = LiquidWorld(DFSPHSolver(), bench.R, 2.0)
        f = Fluid(fluid[mine], bench.R, 1000.0); f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
        w.add_fluid(f); w.add_boundary(Boundary(shell[dist.boundary_subset(shell, H, sl[r], r, nr)]))
        w.set_domain(comms[r], sl[r][0], sl[r][1], int((owner < r).sum()))
        for _ in range(WARM): w.step(bench.DT, bench.GRAVITY)
        bar.wait(); t0 = time.perf_counter(); its = []; gh = 0
        for _ in range(STEPS):
            st = w.step(bench.DT, bench.GRAVITY); its.append(st.n_divergence_iters); gh = int(st.reserved[4])
        bar.wait()
        out[r] = ((time.perf_counter() - t0) / STEPS * 1e3, its, len(mine), gh)
    ts = [threading.Thread(target=main, args=(r,)) for r in range(nr)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return out

ms, its = single()
print(f"1 domain: {ms:.2f} ms/step, divergence iterations {its[-5:]}", flush=True)
for nr in (2, 4):
    o = slabs(nr)
    print(f"{nr} slabs on one GPU: {max(x[0] for x in o):.2f} ms/step (all slabs serialised on the GPU), particles/ghosts per slab "
          f"{[(x[2], x[3]) for x in o]}, iterations agree with the single domain: {all(x[1] == its for x in o)}", flush=True)
