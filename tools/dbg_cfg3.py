import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from parity import DT, Scene, max_norm_diff
from salva_amd import scenes
R = 0.025
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R)
vel = scenes.random_velocities(len(pos), 0.1)
s = Scene(R, 2.0, "iisph")
s.add_fluid(pos, vel, 1000.0, forces=[("akinci", 1.0, 0.0)])
w, (fl,), _ = s.make_hip()
o = s.make_oracle(threads=16)
o64 = s.make_oracle(threads=16, f64=True)
for k in range(3):
    st = w.step(DT, (0, 0, 0)); so = o.step(DT, (0, 0, 0)); s6 = o64.step(DT, (0, 0, 0))
    v, vo, v6 = fl.velocities, o.fluid_vec(0, "velocities"), o64.fluid_vec(0, "velocities")
    print(f"step {k}: iters gpu {st.n_pressure_iters} oracle {so.n_press_iters} f64 {s6.n_press_iters}  err gpu {st.density_error:.3e} oracle {so.density_error:.3e} f64 {s6.density_error:.3e}"
          f"  |dv| gpu-o32 {max_norm_diff(v, vo):.2e}  o32-o64 {max_norm_diff(vo, v6):.2e}  gpu-o64 {max_norm_diff(v, v6):.2e}  ms {st.step_ms:.2f}", flush=True)
