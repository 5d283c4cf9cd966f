"""tools/fuzz_sweep.py — runs tests/test_fuzz_gpu.py's random-operation differential test over 100 further seeds (50 per solver) and
reports where each failing sequence stopped; the sweep behind the fixes listed in DESIGN.md §4 (about 4 s on the GPU box)."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_gpu as T
bad = []
for seed in range(20, 70):
    for solver in ("dfsph", "iisph"):
        try:
            T.test_random_operation_sequences_match_oracle(solver, seed)
        except BaseException as e:  # noqa: BLE001
            tb = traceback.extract_tb(sys.exc_info()[2])
            fr = [f for f in tb if f.filename.endswith("test_fuzz_gpu.py")][-1]
            args = e.args[0] if e.args else None
            log = args[0] if isinstance(args, tuple) else args
            extra = args[1:] if isinstance(args, tuple) else ()
            nlog = len(log) if isinstance(log, list) else -1
            bad.append((solver, seed))
            print("FAIL", solver, seed, "line", fr.lineno, "|", fr.line.strip()[:110], "| after", nlog, "ops, last", [str(x) for x in (log[-4:] if isinstance(log, list) else [])], extra, type(e).__name__, flush=True)
print("failures:", len(bad), "of 100")
