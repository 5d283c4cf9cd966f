#!/usr/bin/env python
"""Kernel experiments: time the execution variants of k_pred_density on the bench scene and check that they write the
same bits.  (salva_hip_time_variant; results in DESIGN.md §3.3.)

  python tools/variant_probe.py [--side 100] [--steps 30] [--jitter 0.1] [--variants 0,2,3,1:64,1:128]
  SALVA_HIP_LIB_VARIANT=t3 python tools/variant_probe.py ...      # 3x4x4-cell tiles
  SALVA_HIP_PIPE_WAVES=8 ...                                      # waves per pipeline workgroup
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from salva_amd import DFSPHSolver, Fluid, Boundary, LiquidWorld, XSPHViscosity, scenes  # noqa: E402

R = 0.025


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=100)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--jitter", type=float, default=0.1)
    ap.add_argument("--variants", default="0,2,3,1:64,1:128,1:192")
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    fluid, shell = scenes.tank(a.side, a.side, a.side, R)
    if a.jitter > 0:
        fluid = scenes.jitter(fluid, a.jitter * R, seed=42)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    f = Fluid(fluid, R, 1000.0)
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(shell))
    for _ in range(a.steps):
        st = w.step(1.0 / 200.0, (0.0, -9.81, 0.0))
    tag = "lib=%s waves=%s side=%d jitter=%g steps=%d halo=%d bhalo=%d threads=%d" % (
        os.environ.get("SALVA_HIP_LIB_VARIANT", "t4"), os.environ.get("SALVA_HIP_PIPE_WAVES", "auto"), a.side, a.jitter, a.steps,
        int(st.reserved[0]), int(st.reserved[1]), int(st.reserved[2]))
    ref = None
    for v in a.variants.split(","):
        var, _, par = v.partition(":")
        try:
            us, cs = w.time_variant(int(var), int(par or 0), a.reps)
        except Exception as e:  # noqa: BLE001
            print(f"{tag} variant={v} FAILED {e}", flush=True)
            continue
        if ref is None:
            ref = cs
        print(f"{tag} variant={v} us={us:.2f} same_bits={'yes' if cs == ref else 'NO'}", flush=True)


if __name__ == "__main__":
    main()
