"""Histogram of particles per 4x4x4-cell tile (an absolute tile lattice: close enough for a histogram) after STEPS steps of the bench scene."""
import os, sys
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
for k in range(steps):
    w.step(DT, G)
p = np.asarray(f.positions, dtype=np.float64)
cells = np.floor(p / 0.1).astype(np.int64)
def hist(keys, label, edges):
    _, cnt = np.unique(keys, axis=0, return_counts=True)
    h = np.histogram(cnt, bins=edges)[0]
    print(label, "n =", len(cnt), dict(zip([f"{a}..{b-1}" for a, b in zip(edges[:-1], edges[1:])], h.tolist())))
    return cnt
tiles = cells // 4
hist(tiles, "particles per tile:", [1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 100000])
# halo population of the sparse tiles: particles in the 6x6x6 cells around tiles that own <= 64
tk, inv, cnt = np.unique(tiles, axis=0, return_inverse=True, return_counts=True)
from collections import Counter
cellcount = Counter(map(tuple, cells.tolist()))
sparse = tk[cnt <= 64]
hal = []
for t in sparse[:3000]:
    base = t * 4 - 1
    s = 0
    for dx in range(6):
        for dy in range(6):
            for dz in range(6):
                s += cellcount.get((base[0] + dx, base[1] + dy, base[2] + dz), 0)
    hal.append(s)
hal = np.asarray(hal)
print("halo of tiles owning <= 64:", dict(zip(["<=64", "<=192", "<=512", "<=1024", ">1024"], [int((hal <= 64).sum()), int(((hal > 64) & (hal <= 192)).sum()), int(((hal > 192) & (hal <= 512)).sum()), int(((hal > 512) & (hal <= 1024)).sum()), int((hal > 1024).sum())])))
