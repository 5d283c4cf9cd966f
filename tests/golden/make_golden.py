"""Generates tests/golden/*.npz with the CPU oracle (oracle/salva_oracle.cpp, f32 build).

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot be built here (pure Rust, no
cargo), so these fixtures pin the *oracle* (and through it the HIP path) against regressions; they are not
reference outputs.  Re-run with `python tests/golden/make_golden.py` after an intentional oracle change.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_scenes import SCENES, run_oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    for name, (builder, nsteps) in SCENES.items():
        out = run_oracle(builder(), nsteps)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")
