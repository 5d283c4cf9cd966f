"""k_pred_density ablations (run on the GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys; sys.path.insert(0, %r)
import bench
from bench import *
fluid, shell = build_scene(100)
w, f = make_world(fluid, shell, 0)
for _ in range(8): st = w.step(DT, GRAVITY)
print("threads", int(st.reserved[2]), "halo", int(st.reserved[0]), "pred_density us", round(w.time_pred_density(100), 2), "ms/step", round(st.step_ms,3), flush=True)
''' % ROOT
for pad in ("0",):
  for mode in ("", "3", "4"):
    env = dict(os.environ, SALVA_HIP_LDS_PAD=pad)
    if mode: env["SALVA_HIP_EXP_MODE"] = mode
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"PAD={pad} MODE={mode or 'product'}:", p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:])
