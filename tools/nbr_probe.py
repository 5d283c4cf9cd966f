"""Time the neighbour-list builder variants on the state a few steps into a bench configuration (run once per variant:
SALVA_HIP_NBR_VARIANT / SALVA_HIP_NBR_THREADS are read once per process)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fluids, shell = bench.build_config(cfg, 100)
w, handles = bench.make_config_world(cfg, fluids, shell, 0)
for _ in range(nsteps):
    st = w.step(bench.DT, bench.GRAVITY)
us = w.time_kernel(4, 20)
cnt = np.concatenate([w.contact_counts(h) for h in handles])
print(f"config {cfg} after {nsteps} steps: variant {os.environ.get('SALVA_HIP_NBR_VARIANT', '2')} threads {os.environ.get('SALVA_HIP_NBR_THREADS', '-')}: "
      f"k_nbr_tile + k_list_stats {us:.1f} us; ncontacts {st.ncontacts}, list checksum {int(cnt.astype(np.uint64).sum())} {int((cnt.astype(np.uint64) * np.arange(len(cnt), dtype=np.uint64) % 1000003).sum())}, step {st.step_ms:.3f} ms")
