#!/usr/bin/env python
"""Pin the oracle against real salva: compare a dump written by bench/rust_ref with what the oracle produced.

  python tests/golden/compare_rust_dump.py --scene NAME DIR     DIR = `cargo run --release -- --scene scenes/NAME.scene --dump DIR`;
                                                                compared with tests/golden/NAME.npz (all seven golden scenes)
  python tests/golden/compare_rust_dump.py --all ROOT           ROOT/NAME for every golden scene
  python tests/golden/compare_rust_dump.py DIR --side S --steps K --warmup W     the bench.py tank (oracle re-run here)

(The image this work was done in has no cargo; whoever has runs bench/rust_ref first.)  Tolerances: salva's own summation
order is unspecified (hash-bucket / rayon order), so trajectories are compared like the GPU path is compared with the oracle:
positions to 1e-3 r, velocities to 1e-2 m/s (x tol_scale of the scene), boundary volumes to 1e-5 relative, accumulated
boundary forces to 1 % of their largest entry, contact counts per step exactly for the first step and to 1e-4 afterwards."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, cols=None):
    a = np.fromfile(path, np.float32)
    return a.reshape(-1, cols) if cols else a


def compare_scene(name, d):
    from golden_scenes import R, SCENES

    gold = np.load(os.path.join(HERE, f"{name}.npz"))
    scale = SCENES[name][0]().tol_scale
    ok = True

    def report(what, err, tol):
        nonlocal ok
        good = err <= tol
        ok &= bool(good)
        print(f"  {what:<28s} {err:.3e}  (tolerance {tol:.1e})  {'ok' if good else 'MISMATCH'}")

    nf = sum(1 for k in gold.files if k.startswith("pos_"))
    for prefix, tol_p, tol_v in (("s1_", 1e-5, 1e-4), ("", 1e-3, 1e-2)):
        for f in range(nf):
            p = load(os.path.join(d, f"{prefix}pos_{f}.f32"), 3)
            v = load(os.path.join(d, f"{prefix}vel_{f}.f32"), 3)
            report(f"{prefix}pos_{f} [r]", np.abs(p - gold[f"{prefix}pos_{f}"]).max() / R, tol_p * scale)
            report(f"{prefix}vel_{f} [m/s]", np.abs(v - gold[f"{prefix}vel_{f}"]).max(), tol_v * scale)
    for k in sorted(k for k in gold.files if k.startswith("s1_bvol_")):
        b = k.rsplit("_", 1)[1]
        vol = load(os.path.join(d, f"s1_bvol_{b}.f32"))
        report(f"s1_bvol_{b} [rel]", float(np.max(np.abs(vol - gold[k]) / np.maximum(np.abs(gold[k]), 1e-30))), 1e-5)
    for k in sorted(k for k in gold.files if k.startswith("bforce_")):
        b = k.rsplit("_", 1)[1]
        fr = load(os.path.join(d, f"bforce_{b}.f32"), 3)
        report(f"bforce_{b} [rel max]", float(np.abs(fr - gold[k]).max() / max(np.abs(gold[k]).max(), 1e-30)), 1e-2)
    nc = np.loadtxt(os.path.join(d, "ncontacts.txt"), dtype=np.int64).reshape(-1)
    gc = gold["iters"][:, 2]
    report("ncontacts, first step", abs(int(nc[0]) - int(gc[0])), 0)
    report("ncontacts, later [rel]", float(np.max(np.abs(nc - gc) / np.maximum(gc, 1))), 1e-4)
    return ok


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--side", type=int, default=20)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    if a.all or a.scene:
        from golden_scenes import SCENES

        names = list(SCENES) if a.all else [a.scene]
        good = True
        for name in names:
            print(name)
            good &= compare_scene(name, os.path.join(a.dir, name) if a.all else a.dir)
        print("oracle pinned against salva on", ", ".join(names) if good else "MISMATCH")
        sys.exit(0 if good else 1)
    import bench  # noqa: E402
    from oracle import oracle as O  # noqa: E402

    fluid, shell = bench.build_scene(a.side)
    w = O.OracleWorld(bench.R, 2.0, O.DFSPH, threads=os.cpu_count() or 1)
    f = w.add_fluid(fluid, 1000.0)
    w.add_xsph(f, 0.5, 0.0)
    w.add_boundary(shell)
    for _ in range(a.warmup + a.steps):
        w.step(bench.DT, bench.GRAVITY)
    ref_p = load(os.path.join(a.dir, "positions.f32"), 3)
    ref_v = load(os.path.join(a.dir, "velocities.f32"), 3)
    p, v = w.fluid_vec(f, "positions"), w.fluid_vec(f, "velocities")
    dp = np.abs(p - ref_p).max() / bench.R
    dv = np.abs(v - ref_v).max()
    print(f"max |dx| = {dp:.3e} r   max |dv| = {dv:.3e} m/s over {len(p)} particles after {a.warmup + a.steps} steps")
    sys.exit(0 if dp < 1e-3 and dv < 1e-2 else 1)
