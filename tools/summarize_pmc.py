#!/usr/bin/env python
"""tools/summarize_pmc.py DIR — per-kernel means of the rocprofv3 --pmc passes written by tools/profile.sh.

Writes DIR/pmc_summary.csv (kernel, counter, launches, mean per launch) and DIR/hbm_traffic.json: HBM bytes per launch
of the neighbour-sum kernels, corrected as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes for gfx950
(FETCH_SIZE counts 64 B per 128-B request: doubled; FETCH_SIZE / WRITE_SIZE are reported in KiB)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(os.path.join(d, "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("salva::", "").replace("void ", "")
        name = re.sub(r"<[\du, ]+>$", "", name)  # (layout / variant instantiations of one kernel are summed under its name)
        name = re.sub(r"_p[23](_two|_multi)?$", "", name)   # (the plane-layout forms of the DFSPH solver kernels, uniform and two-mass: the same pass)
        k = (name, r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
rows = sorted(acc.items())
with open(os.path.join(d, "pmc_summary.csv"), "w") as out:
    out.write("kernel,counter,launches,mean_per_launch\n")
    for (name, ctr), (n, tot) in rows:
        out.write(f"{name},{ctr},{n},{tot / n:.6g}\n")
traffic = {}
for (name, ctr), (n, tot) in rows:
    if ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        traffic.setdefault(name, {})[ctr] = tot / n
res = {}
for name, t in traffic.items():
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        rd = 2.0 * t["FETCH_SIZE"] * 1024.0
        wr = t["WRITE_SIZE"] * 1024.0
        res[name] = {"read_bytes": rd, "write_bytes": wr, "bytes": rd + wr}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salva_amd import kernel_source_sha  # noqa: E402

json.dump({"note": "per launch; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB (uncalibrated)",
           "kernel_src_sha": kernel_source_sha(),  # bench.py refuses the figure when the kernel sources have changed since
           "kernels": res}, open(os.path.join(d, "hbm_traffic.json"), "w"), indent=1)
for name, v in sorted(res.items()):
    print(f"{name:28s} read {v['read_bytes'] / 1e6:8.1f} MB  write {v['write_bytes'] / 1e6:7.1f} MB")
