"""Large single-GPU runs: 200^3 = 8 M (BASELINE config 5's total, here on one GPU) and 300^3 = 27 M particles, DFSPH + XSPH in the
tank: table sizes, memory, time per step, invariants (sum of list lengths = reported contacts, finite state)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
for side in [int(x) for x in (sys.argv[1:] or ["200", "300"])]:
    t0 = time.perf_counter()
    fl, sh = bench.build_scene(side)
    w, f = bench.make_world(fl, sh, 0)
    w.counters.enable()  # (step_ms comes from the stage timers, off by default)
    t1 = time.perf_counter()
    ms = []
    for k in range(8):
        st = w.step(bench.DT, bench.GRAVITY); ms.append(st.step_ms)
    n = len(fl)
    cnt = int(w.contact_counts(f).astype(np.int64).sum() + w.contact_counts(f, True).astype(np.int64).sum())
    p = f.positions
    kus = w.time_pred_density(20)
    kbar = cnt / n
    frac = n * (4.0 * kbar + 52.0) / (kus * 1e-6) / 8e12
    print(f"   k_pred_density {kus:.1f} us = {frac:.3f} of the 8 TB/s roofline (algorithmic bytes N(4K+52), K = {kbar:.2f})")
    print(f"{side}^3 = {n} particles (+{len(sh)} boundary): scene {t1-t0:.1f} s, step ms {['%.1f' % m for m in ms]}, "
          f"{n/ (np.mean(ms[3:])*1e-3)/1e6:.0f} M particle-steps/s, device {w.device_bytes()/2**30:.2f} GiB, "
          f"list entries {cnt} <= contacts {st.ncontacts}, finite {bool(np.isfinite(p).all())}, halo {int(st.reserved[0])}", flush=True)
    del w, f
