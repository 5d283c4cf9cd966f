"""tools/mirror_sweep.py — tests/test_mirrors_gpu.py over further seeds (C++ vs Python mirror, bit-identical replay)."""
import os, sys, pathlib, tempfile, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mirrors_gpu as T
bad = 0
for seed in range(100, 112):
    for solver in ("dfsph", "iisph"):
        with tempfile.TemporaryDirectory() as d:
            try:
                T.test_cpp_and_python_mirrors_replay_identically(solver, seed, pathlib.Path(d))
            except BaseException as e:  # noqa: BLE001
                bad += 1
                print("FAIL", solver, seed, repr(e)[:300], flush=True)
print("failures:", bad, "of 24")
