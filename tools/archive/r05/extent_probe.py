"""Why a long run slows down: every 100 steps of the bench scene, the particles' bounding box in cells, how many particles have
left the tank, and the stage timers of one step (grid / solver).  tools/r05/soak.sh found config 2 going from 2.0 to 4.1 ms per
step between steps 300 and 1000 at a constant iteration count."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fluids, shell = bench.build_config(config, 100)
w, handles = bench.make_config_world(config, fluids, shell, 0)
h = 4.0 * bench.R
lo, hi = shell.min(axis=0), shell.max(axis=0)
t0 = time.perf_counter()
for s in range(nsteps + 1):
    if s % 100 == 0:
        pos = np.concatenate([np.asarray(f.positions) for f in handles])
        mn, mx = pos.min(axis=0), pos.max(axis=0)
        cells = np.floor(mx / h) - np.floor(mn / h) + 1
        out = int(((pos < lo - h) | (pos > hi + h)).any(axis=1).sum())
        below = int((pos[:, 1] < lo[1] - h).sum())
        w.counters.enable()
        st = w.step(bench.DT, bench.GRAVITY)
        w.counters.disable()
        print(f"step {s:5d}: bbox {cells.astype(int).tolist()} cells = {np.prod(cells):.3e}, y in [{mn[1]:9.2f}, {mx[1]:7.2f}], "
              f"{out} particles outside the tank ({below} below the floor); grid {st.grid_ms:.3f} ms solver {st.solver_ms:.3f} ms, "
              f"iters ({st.n_divergence_iters}, {st.n_pressure_iters}), wall {time.perf_counter() - t0:.1f} s", flush=True)
    else:
        w.step(bench.DT, bench.GRAVITY)
