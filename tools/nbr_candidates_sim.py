#!/usr/bin/env python
"""tools/nbr_candidates_sim.py — CPU study for the neighbour-list kernel (k_nbr_tile): how many exact
distance tests per particle do different candidate-culling schemes need on the bench scene's packing?  No GPU involved; the
output is the quantitative basis for the next redesign of that kernel (profiles/r03_experiments/nbr_candidates_sim.log).

Scene: a jittered lattice at the bench's packing (spacing 2r = h/2, jitter 0.1 r), interior particles only.  Schemes:
  cells27      today: every particle of the 3x3x3 cells around the particle's cell (9 rows of 3 z-adjacent cells);
  quads        today, as executed: rows are walked in aligned groups of four slots (one ds_read_b128 per axis plane), so a row
               costs ceil((lead + n) / 4) * 4 tests;
  rowskip      skip a whole row when the particle's distance to that row's cell column (in the two axes across the row)
               exceeds h — needs nothing sorted;
  rowwindow    particles sorted along the row axis inside a row; test only those whose coordinate along the row lies within
               sqrt(h^2 - gap^2) of the particle's, gap = distance to the row's column as above (exact per-row chord);
  rowwindow/4  the same at the granularity of aligned groups of four;
  halfcells    cells of edge h/2: the 5x5x5 half-cells around the particle's half-cell, minus those farther than h (125 -> fewer
               cells, 25 rows of 5).
  rowbins4     rowwindow without a search: particles ordered by quarter-cell bins along the row inside each cell (2 more sort-key
               bits), a per-row table of bin starts; the window is rounded outwards to bins.
Reported: mean tests per particle, and the ratio to the true neighbour count (the useful work).  Second table: the same at the
granularity the kernel executes — a wave of 64 consecutive own particles walks its nine rows in lock-step, four candidates per
trip, and pays the LONGEST trip count among its lanes for every row."""
import numpy as np


def main(n=28, seed=3):
    rng = np.random.default_rng(seed)
    r = 0.025
    h = 4 * r
    g = (np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), axis=-1).reshape(-1, 3) + 0.5) * 2 * r
    x = g + rng.uniform(-0.1 * r, 0.1 * r, g.shape)
    cell = np.floor(x / h).astype(np.int64)
    nc = cell.max(axis=0) + 1
    interior = ((cell >= 2) & (cell <= nc - 3)).all(axis=1)
    # sort by (cx, cy, cz, z): rows run along z like the kernel's (3 z-adjacent cells are contiguous)
    key = ((cell[:, 0] * nc[1] + cell[:, 1]) * nc[2] + cell[:, 2])
    order = np.lexsort((x[:, 2], key))
    x, cell, key, interior = x[order], cell[order], key[order], interior[order]
    start = np.searchsorted(key, np.arange(nc.prod() + 1))
    idx = np.nonzero(interior)[0]
    first = int(rng.integers(0, len(idx) - 64 * 64))
    idx = idx[first:first + 64 * 64]  # 64 waves of 64 consecutive particles (sorted order = the kernel's slices)
    tot = dict(true=0, cells27=0, quads=0, rowskip=0, rowwindow=0, rowwindow4=0, rowbins4=0, halfcells=0)
    per_lane = {}  # particle -> [(quads today, quads with bin windows) per row]
    hx = x / (h / 2)
    hcell = np.floor(hx).astype(np.int64)
    for i in idx:
        p, c = x[i], cell[i]
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                b = start[((c[0] + dx) * nc[1] + (c[1] + dy)) * nc[2] + (c[2] - 1)]
                e = start[((c[0] + dx) * nc[1] + (c[1] + dy)) * nc[2] + (c[2] + 1) + 1]
                q = x[b:e]
                d2 = ((q - p) ** 2).sum(axis=1)
                tot["true"] += int((d2 <= h * h).sum())
                tot["cells27"] += e - b
                tot["quads"] += ((e - (b & ~3)) + 3) // 4 * 4
                # gap of the particle to the row's column in x and y
                lo = np.array([(c[0] + dx) * h, (c[1] + dy) * h])
                gap = np.maximum(0.0, np.maximum(lo - p[:2], p[:2] - (lo + h)))
                g2 = float((gap ** 2).sum())
                quads_today = ((e - (b & ~3)) + 3) // 4
                quads_bins = 0
                if g2 <= h * h:
                    wz0 = np.sqrt(h * h - g2)
                    zlo, zhi = np.floor((p[2] - wz0) / (h / 4)) * (h / 4), (np.floor((p[2] + wz0) / (h / 4)) + 1) * (h / 4)
                    lo_b = b + int(np.searchsorted(q[:, 2], zlo, side="left"))
                    hi_b = b + int(np.searchsorted(q[:, 2], zhi, side="left"))
                    tot["rowbins4"] += hi_b - lo_b
                    quads_bins = ((hi_b - (lo_b & ~3)) + 3) // 4 if hi_b > lo_b else 0
                per_lane.setdefault(int(i), []).append((quads_today, quads_bins))
                if g2 <= h * h:
                    tot["rowskip"] += e - b
                    wz = np.sqrt(h * h - g2)
                    lo_i = b + int(np.searchsorted(q[:, 2], p[2] - wz, side="left"))
                    hi_i = b + int(np.searchsorted(q[:, 2], p[2] + wz, side="right"))
                    tot["rowwindow"] += hi_i - lo_i
                    if hi_i > lo_i:
                        tot["rowwindow4"] += ((hi_i - (lo_i & ~3)) + 3) // 4 * 4
        # half cells: count particles in half-cells whose box is within h of the particle
        hc = hcell[i]
        near = (np.abs(hcell - hc) <= 2).all(axis=1)
        cand = np.nonzero(near)[0]
        lo = hcell[cand] * (h / 2)
        gap = np.maximum(0.0, np.maximum(lo - p, p - (lo + h / 2)))
        tot["halfcells"] += int(((gap ** 2).sum(axis=1) <= h * h).sum())
    m = len(idx)
    print(f"{m} interior particles of a {n}^3 lattice at the bench packing (h = {h}, spacing h/2, jitter 0.1 r)")
    t = tot["true"] / m
    for k in ("true", "cells27", "quads", "rowskip", "rowwindow", "rowwindow4", "rowbins4", "halfcells"):
        v = tot[k] / m
        print(f"  {k:12s} {v:7.1f} tests per particle   {v / t:5.2f} x the true neighbour count   {v / (tot['quads'] / m):5.2f} of today's executed tests")
    # wave granularity
    today = bins = 0
    for wv in range(0, m, 64):
        lanes = [per_lane[int(i)] for i in idx[wv:wv + 64]]
        for row in range(9):
            today += max(l[row][0] for l in lanes)
            bins += max(l[row][1] for l in lanes)
    nw = m // 64
    print(f"per wave of 64 consecutive particles, trips of four candidates summed over the nine rows (longest lane per row):")
    print(f"  today        {today / nw:6.1f} trips")
    print(f"  rowbins4     {bins / nw:6.1f} trips   {bins / today:5.2f} of today's")


if __name__ == "__main__":
    main()
