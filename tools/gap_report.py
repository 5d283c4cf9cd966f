"""Attribute the GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV to the kernel that ended before the gap."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("salva::", "")))
rows.sort()
# steps are delimited by k_cell_keys launches on the fluid (first kernel of a step)
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_cell_keys")]
print(f"{len(rows)} kernels, {len(starts)} k_cell_keys launches")
if len(starts) < 4:
    sys.exit(0)
lo, hi = starts[-4], starts[-1]  # three full steps near the end
seg = rows[lo:hi]
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
gaps = defaultdict(lambda: [0, 0])
for a, b in zip(seg[:-1], seg[1:]):
    g = b[0] - a[1]
    if g > 0:
        gaps[(a[2][:40], b[2][:40])][0] += g
        gaps[(a[2][:40], b[2][:40])][1] += 1
nsteps = 3
print(f"per step: span {span / nsteps / 1e3:.1f} us, kernels busy {busy / nsteps / 1e3:.1f} us, idle {(span - busy) / nsteps / 1e3:.1f} us, {len(seg) / nsteps:.0f} launches")
print("largest idle gaps per step (after kernel -> before kernel: total us, count, mean us):")
for (a, b), (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {a:40s} -> {b:40s} {g / nsteps / 1e3:8.1f} {c / nsteps:6.1f} {g / c / 1e3:7.2f}")
bykern = defaultdict(lambda: [0, 0])
for s, e, k in seg:
    bykern[k[:50]][0] += e - s
    bykern[k[:50]][1] += 1
print("kernel time per step (us, launches):")
for k, (t, c) in sorted(bykern.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {k:50s} {t / nsteps / 1e3:8.1f} {c / nsteps:6.1f}")
