#!/usr/bin/env python
"""Writes every scene of tests/golden_scenes.py (and, with --bench SIDE, the bench.py tank) as a `.scene` file that
bench/rust_ref — the REAL salva3d, for whoever has cargo; this image has none — loads with `--scene FILE --dump DIR`.
tests/golden/compare_rust_dump.py then checks the dump against tests/golden/<name>.npz, which pins the oracle that produced
those fixtures against salva itself.

Format (little endian): magic "SLVSCN01"; f32 radius, smoothing; u32 solver (0 DFSPH, 1 IISPH); u32 min/max pressure iter,
f32 max density error, u32 min/max divergence iter, f32 max divergence error; u32 nsteps; f32 dt; f32 gravity[3];
u32 nfluids, per fluid: u32 n, f32 density0, u32 memberships, filter, has_vel, has_vol, nforces; per force u32 kind
(1 XSPH, 2 Artificial, 3 Akinci2013, 4 DFSPHViscosity, 5 He2014, 6 WCSPH), u32 nparams, f32 params[]; f32 pos[3n], vel[3n]?, vol[n]?;
u32 nboundaries, per boundary: u32 n, memberships, filter, wants_forces, has_vel; f32 pos[3n], vel[3n]?."""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_scenes import SCENES  # noqa: E402
from parity import DT, GRAVITY, Scene  # noqa: E402

KINDS = {"xsph": 1, "artificial": 2, "akinci": 3, "dfsph_viscosity": 4, "he2014": 5, "wcsph_tension": 6}
DEFAULTS = {"artificial": [None, None, 1.0, 0.0, 10.0], "dfsph_viscosity": [None, 1, 50, 0.01]}


def write_scene(scene: Scene, nsteps: int, path: str):
    sp = scene.solver_params
    with open(path, "wb") as fh:
        fh.write(b"SLVSCN01")
        fh.write(struct.pack("<ffI", scene.radius, scene.smoothing, 0 if scene.solver == "dfsph" else 1))
        fh.write(struct.pack("<IIfIIf", sp["min_pressure_iter"], sp["max_pressure_iter"], sp["max_density_error"],
                             sp["min_divergence_iter"], sp["max_divergence_iter"], sp["max_divergence_error"]))
        fh.write(struct.pack("<If3f", nsteps, DT, *GRAVITY))
        fh.write(struct.pack("<I", len(scene.fluids)))
        for f in scene.fluids:
            n = len(f["pos"])
            fh.write(struct.pack("<IfIIIII", n, f["density0"], f["groups"][0], f["groups"][1], int(f["vel"] is not None),
                                 int(f["volumes"] is not None), len(f["forces"])))
            for frc in f["forces"]:
                params = list(frc[1:])
                full = DEFAULTS.get(frc[0])
                if full is not None:
                    params = params + full[len(params):]
                fh.write(struct.pack("<II", KINDS[frc[0]], len(params)))
                fh.write(np.asarray(params, "<f4").tobytes())
            fh.write(np.ascontiguousarray(f["pos"], "<f4").tobytes())
            if f["vel"] is not None:
                fh.write(np.ascontiguousarray(f["vel"], "<f4").tobytes())
            if f["volumes"] is not None:
                fh.write(np.ascontiguousarray(f["volumes"], "<f4").tobytes())
        fh.write(struct.pack("<I", len(scene.boundaries)))
        for b in scene.boundaries:
            fh.write(struct.pack("<IIIII", len(b["pos"]), b["groups"][0], b["groups"][1], int(b["wants_forces"]), int(b["vel"] is not None)))
            fh.write(np.ascontiguousarray(b["pos"], "<f4").tobytes())
            if b["vel"] is not None:
                fh.write(np.ascontiguousarray(b["vel"], "<f4").tobytes())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "bench", "rust_ref", "scenes"))
    ap.add_argument("--bench", type=int, default=0, help="also write the bench.py tank with this block side (100 = BASELINE config 2)")
    ap.add_argument("--bench-steps", type=int, default=55)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for name, (builder, nsteps) in SCENES.items():
        path = os.path.join(a.out, f"{name}.scene")
        write_scene(builder(), nsteps, path)
        print(name, os.path.getsize(path), "bytes")
    if a.bench:
        import bench

        fluid, shell = bench.build_scene(a.bench)
        s = Scene(bench.R, 2.0, "dfsph")
        s.add_fluid(fluid, None, 1000.0, forces=[("xsph", 0.5, 0.0)])
        s.add_boundary(shell)
        path = os.path.join(a.out, f"bench_tank_{a.bench}.scene")
        write_scene(s, a.bench_steps, path)
        print("bench tank", os.path.getsize(path), "bytes")
