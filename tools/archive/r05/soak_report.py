"""Summarise long bench runs (tools/r05/soak.sh): per window of 100 timed steps the mean step time, the throughput and the mean
(divergence, pressure) iteration counts; flags a non-finite or non-positive step time."""
import json
import sys

for path in sys.argv[1:]:
    line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    ms, it = d["per_step_ms"], d["iters"]
    n = d["config"]["particles_per_gpu"]
    assert all(m == m and m > 0 for m in ms), "a step time is not a positive number"
    print(f"{path}: {d['config']['workload']}")
    print(f"  {len(ms)} timed steps, whole run {d['ms_per_step']:.3f} ms/step = {d['value']:.3e} {d['unit']}")
    for a in range(0, len(ms), 100):
        w, wi = ms[a:a + 100], it[a:a + 100]
        m = sum(w) / len(w)
        print(f"  steps {a:4d}..{a + len(w) - 1:4d}: {m:7.3f} ms/step  {n / m * 1e3:10.3e} particle-steps/s   iterations (div, press) mean "
              f"({sum(x[0] for x in wi) / len(wi):5.1f}, {sum(x[1] for x in wi) / len(wi):4.1f})  max ({max(x[0] for x in wi)}, {max(x[1] for x in wi)})")
