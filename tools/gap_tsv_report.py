"""tools/gap_tsv_report.py kernels.tsv [first_step last_step] — device timeline of a few steps from the compact kernel list
tools/gap_trace.sh leaves behind (name, start ns, end ns): span, busy and idle time per step, the largest idle gaps by the pair of
kernels around them, and the kernel time per step."""
import sys
from collections import defaultdict

rows = []
for line in open(sys.argv[1]):
    name, s, e = line.rstrip("\n").split("\t")
    rows.append((int(s), int(e), name.replace("salva::", "").replace("void ", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_cell_keys")]
print(f"{len(rows)} kernels, {len(starts)} steps")
a = int(sys.argv[2]) if len(sys.argv) > 2 else max(len(starts) - 5, 0)
b = int(sys.argv[3]) if len(sys.argv) > 3 else len(starts) - 1
seg = rows[starts[a]:starts[b]]
n = b - a
span = seg[-1][1] - seg[0][0]
# the idle time after the last kernel of a step belongs to the step as well: measure to the next step's first kernel
span = rows[starts[b]][0] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print(f"steps {a}..{b - 1}: per step span {span / n / 1e3:.1f} us, kernels busy {busy / n / 1e3:.1f} us, idle {(span - busy) / n / 1e3:.1f} us, {len(seg) / n:.0f} launches")
gaps = defaultdict(lambda: [0, 0])
for x, y in zip(seg, seg[1:] + [rows[starts[b]]]):
    g = y[0] - x[1]
    if g > 0:
        gaps[(x[2][:36], y[2][:36])][0] += g
        gaps[(x[2][:36], y[2][:36])][1] += 1
print("largest idle gaps per step (after -> before: us per step, count per step, mean us):")
for (x, y), (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {x:36s} -> {y:36s} {g / n / 1e3:7.1f} {c / n:5.1f} {g / c / 1e3:7.2f}")
byk = defaultdict(lambda: [0, 0])
for s, e, k in seg:
    byk[k[:48]][0] += e - s
    byk[k[:48]][1] += 1
print("kernel time per step (us, launches per step):")
for k, (t, c) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {k:48s} {t / n / 1e3:8.1f} {c / n:5.1f}")
