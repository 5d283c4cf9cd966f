"""Per-step wall time and stage times of the bench scene (diagnostics)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
fl, sh = bench.build_scene(100)
w, f = bench.make_world(fl, sh, 0)
w.counters.enable()  # (step_ms / grid_ms come from the stage timers, off by default)
rows = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    t0 = time.perf_counter()
    st = w.step(bench.DT, bench.GRAVITY)
    t1 = time.perf_counter()
    c = w.counters
    rows.append((k, (t1 - t0) * 1e3, st.grid_ms, st.solver_ms, st.n_divergence_iters, st.n_pressure_iters, c.speculative_passes, c.discarded_passes))
for r in rows:
    print("step %2d wall %.3f ms grid %.3f solver %.3f iters %d/%d spec %d discarded %d" % r)
