"""Parity report of SURVEY.md §8c: GPU vs oracle-f32 vs oracle-f64 after N in {1, 10, 100} steps — max and RMS of |dx| / r and
of |d(v + dv)| / v_ref, and the iteration-count traces.  Run on the GPU box; the output is committed under profiles/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from parity import DT, GRAVITY, Scene
from salva_amd import scenes

R = 0.025

def tank_scene(solver, forces, n=20):
    s = Scene(R, 2.0, solver)
    fluid, shell = scenes.tank(n, n, n, R)
    s.add_fluid(scenes.jitter(fluid, 0.1 * R, 42), None, 1000.0, forces=forces)
    s.add_boundary(shell)
    return s

def free_scene(solver, forces, n=20):
    s = Scene(R, 2.0, solver)
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, 42)
    s.add_fluid(pos, scenes.random_velocities(len(pos), 0.1, 12345), 1000.0, forces=forces)
    return s

CASES = [("config 2(A) 20^3 tank, DFSPH + XSPH(0.5,0)", lambda: tank_scene("dfsph", [("xsph", 0.5, 0.0)])),
         ("config 2(B) 20^3 free block, DFSPH + XSPH(0.5,0)", lambda: free_scene("dfsph", [("xsph", 0.5, 0.0)])),
         ("config 3 20^3 tank, IISPH + Akinci(1,10)", lambda: tank_scene("iisph", [("akinci", 1.0, 10.0)]))]

def stats(a, b, scale):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=1) / scale
    return d.max(), np.sqrt((d * d).mean())

for name, build in CASES:
    sc = build()
    w, (fl,), _ = sc.make_hip()
    o32, o64 = sc.make_oracle(threads=16), sc.make_oracle(threads=16, f64=True)
    print(f"== {name}: {len(sc.fluids[0]['pos'])} particles")
    print("   N | GPU vs oracle-f32: max, rms |dx|/r ; max, rms |dw|/v_ref | oracle-f32 vs f64: max |dx|/r, max |dw|/v_ref | iterations (div, press) GPU / f32 / f64")
    for k in range(1, 101):
        st = w.step(DT, GRAVITY); s32 = o32.step(DT, GRAVITY); s64 = o64.step(DT, GRAVITY)
        if k in (1, 10, 100):
            wg = fl.velocities.astype(np.float64) + w.velocity_changes(fl)
            w32 = o32.fluid_vec(0, "velocities") + o32.fluid_vec(0, "velocity_changes")
            w64 = o64.fluid_vec(0, "velocities") + o64.fluid_vec(0, "velocity_changes")
            vref = max(np.abs(w64).max(), 2 * R / DT * 1e-2)
            gx = stats(fl.positions, o32.fluid_vec(0, "positions"), R); gw = stats(wg, w32, vref)
            ox = stats(o32.fluid_vec(0, "positions"), o64.fluid_vec(0, "positions"), R); ow = stats(w32, w64, vref)
            print(f" {k:3d} | {gx[0]:.2e} {gx[1]:.2e} ; {gw[0]:.2e} {gw[1]:.2e} | {ox[0]:.2e} {ow[0]:.2e} | "
                  f"({st.n_divergence_iters},{st.n_pressure_iters}) / ({s32.n_div_iters},{s32.n_press_iters}) / ({s64.n_div_iters},{s64.n_press_iters})", flush=True)
