#!/usr/bin/env python
"""A/B probe for kernel work: run the bench scene on ONE library build and print one line of timings + checksums.

  SALVA_HIP_LIB_VARIANT=<name> python tools/ab_probe.py [--config 2] [--side 100] [--steps 25] [--reps 30]

`make -C salva_amd/csrc VARIANT=<name> VDEFS="-D..."` builds libsalva_hip_<name>.so; run this once per build (separate
processes) and compare the lines.  Kernel ids: salva_hip_time_kernel (0 k_pred_density, 1 k_divergence, 2/3 IISPH, 4 k_nbr_tile).
The checksum is over the final positions and the iteration trace, so two builds whose arithmetic is meant to be identical
can be checked for it, and builds whose arithmetic differs show how far apart they end up (max |dx| / r vs a reference
.npy via --ref / --save).
"""
import argparse
import faulthandler
import hashlib
import os
import sys
import time

faulthandler.dump_traceback_later(int(os.environ.get("AB_PROBE_WATCHDOG", "100")), exit=True)  # a hang prints where, then exits

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from salva_amd import (Akinci2013SurfaceTension, Boundary, DFSPHSolver, Fluid, IISPHSolver, LiquidWorld, XSPHViscosity,  # noqa: E402
                       scenes)

R, DT, GRAVITY = 0.025, 1.0 / 200.0, (0.0, -9.81, 0.0)


def build_world(config, side):
    """bench.py's scenes for configs 2 and 3 (without importing bench: that pulls in torch, a minute on a fresh box)."""
    fluid, shell = scenes.tank(side, side, side, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
    w = LiquidWorld(IISPHSolver() if config == 3 else DFSPHSolver(), R, float(os.environ.get("AB_SMOOTHING", "2.0")))
    f = Fluid(fluid, R, 1000.0)
    f.nonpressure_forces.append(Akinci2013SurfaceTension(1.0, 10.0) if config == 3 else XSPHViscosity(0.5, 0.0))
    h = w.add_fluid(f)
    w.add_boundary(Boundary(shell))
    return w, [h]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--side", type=int, default=100)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--kernels", default="")
    ap.add_argument("--sched", default="", help="comma list of SALVA_HIP_SCHED values: one world per value, same process")
    ap.add_argument("--save", default="")
    ap.add_argument("--ref", default="")
    a = ap.parse_args()
    for sched in (a.sched.split(",") if a.sched else [None]):
        if sched is not None:
            os.environ["SALVA_HIP_SCHED"] = sched
        run_one(a, sched)


def run_one(a, sched):
    assert a.config in (2, 3)
    w, handles = build_world(a.config, a.side)
    iters = []
    ms = []
    print("AB-progress world built", flush=True)
    for _ in range(a.steps):
        t0 = time.perf_counter()
        st = w.step(DT, GRAVITY)
        ms.append((time.perf_counter() - t0) * 1e3)
        iters.append((int(st.n_divergence_iters), int(st.n_pressure_iters)))
    kernels = [int(k) for k in a.kernels.split(",")] if a.kernels else ([2, 3, 4] if a.config == 3 else [0, 1, 4])
    print("AB-progress stepped", iters[-1], flush=True)
    us = {}
    for k in kernels:
        print("AB-progress timing kernel", k, flush=True)
        try:
            us[k] = w.time_kernel(k, a.reps)
        except Exception as e:  # noqa: BLE001
            us[k] = "ERR %s" % e
    pos = np.concatenate([np.asarray(h.positions, dtype=np.float32) for h in handles])
    hsh = hashlib.sha1(pos.tobytes() + repr(iters).encode()).hexdigest()[:12]
    extra = ""
    if a.ref and os.path.exists(a.ref):
        ref = np.load(a.ref)
        extra = " max|dx|/r vs ref = %.3e" % (float(np.abs(pos - ref).max()) / R)
    if a.save and not os.path.exists(a.save):  # (the first world of the first process is the reference)
        np.save(a.save, pos)
    tag = os.environ.get("SALVA_HIP_LIB_VARIANT", "product") + ("" if sched is None else "/sched=" + sched)
    print("AB lib=%s config=%d side=%d steps=%d | kernel us %s | step ms first5 %.3f last5 %.3f | iters last %s | halo %d bhalo %d threads %d | sha %s%s" % (
        tag, a.config, a.side, a.steps, " ".join("%d:%s" % (k, ("%.2f" % v) if isinstance(v, float) else v) for k, v in us.items()),
        float(np.mean(ms[1:6])), float(np.mean(ms[-5:])), iters[-3:], int(st.reserved[0]), int(st.reserved[1]), int(st.reserved[2]), hsh, extra), flush=True)


if __name__ == "__main__":
    main()
