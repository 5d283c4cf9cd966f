"""Which step, which rank: the strays scene of tests/test_dist_gpu.py with and without forced launch classes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import test_dist_gpu as T

pos, vel, bpos = T.make_scene(nx=48, ny=10, nz=10, seed=9)
rng = np.random.default_rng(3)
stray = np.zeros((50, 3), np.float32)
stray[:, 0] = rng.uniform(pos[:, 0].min(), pos[:, 0].max(), 50)
stray[:25, 1] = rng.uniform(-60.0, -20.0, 25)
stray[25:, 1] = rng.uniform(0.2, 0.6, 25)
stray[25:, 2] = rng.uniform(15.0, 40.0, 25)
stray[:25, 2] = rng.uniform(pos[:, 2].min(), pos[:, 2].max(), 25)
svel = np.zeros((50, 3), np.float32); svel[:, 1] = -3.0
pos2, vel2 = np.concatenate([pos, stray]), np.concatenate([vel, svel])
nsteps = 10
import salva_amd
for tag, env in (("plain", {}), ("classes", {"SALVA_HIP_CLASSES": "1"}), ("nofold", {"SALVA_HIP_NO_FOLD": "1"}), ("nofold-classes", {"SALVA_HIP_NO_FOLD": "1", "SALVA_HIP_CLASSES": "1"}),
                 ("nofold-classes-nochain", {"SALVA_HIP_NO_FOLD": "1", "SALVA_HIP_CLASSES": "1", "SALVA_HIP_NO_CHAIN": "1", "SALVA_HIP_NO_PREGRID": "1"}),
                 ("nofold-classes-compact", {"SALVA_HIP_NO_FOLD": "1", "SALVA_HIP_CLASSES": "1", "SALVA_HIP_COMPACT_HALO": "1"}),
                 ("nofold-classes-noplanes", {"SALVA_HIP_NO_FOLD": "1", "SALVA_HIP_CLASSES": "1", "SALVA_HIP_NO_PLANES": "1"})):
    for k in ("SALVA_HIP_CLASSES", "SALVA_HIP_NO_FOLD", "SALVA_HIP_NO_CHAIN", "SALVA_HIP_NO_PREGRID", "SALVA_HIP_COMPACT_HALO", "SALVA_HIP_NO_PLANES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p, v, st = T.run_single(pos2, vel2, bpos, nsteps, False)
    print(tag, [int(s.ncontacts) for s in st], [(s.n_divergence_iters, s.n_pressure_iters) for s in st], [int(s.reserved[0]) for s in st])
