#!/usr/bin/env python
"""LDS bank-conflict model of the plane-layout (p3 / p2) neighbour-sum loops: ds_read_b64, lane groups {0-31} {32-63}, bank pair
= slot mod 32 (8-byte planes; MI355X_MICROARCH.md LDS table).  A jittered 2r lattice, one interior 4x4x4-cell tile + halo, lists in
k_nbr_tile's order (9 rows of 3 z-adjacent cells, ascending slot within a row), padded per 64-particle slice with self contacts.
Reports LDS cycles per 32-lane group access (1 = conflict free) for candidate list orders / slot maps."""
import numpy as np, sys, itertools
rng = np.random.default_rng(1)
r = 0.025; d = 2*r; h = 4*r
JIT = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
n = 14
g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij'), -1).reshape(-1,3).astype(np.float64)
pos = (g + 0.5) * d + rng.uniform(-JIT*r, JIT*r, size=g.shape) - d
cell = np.floor(pos / h).astype(int)
inh = ((cell >= 0) & (cell <= 5)).all(1)
P = pos[inh]; C = cell[inh]
perm0 = rng.permutation(len(P))   # arbitrary order within a cell
P = P[perm0]; C = C[perm0]

def build(halo_key, row_axis_fast='z'):
    """halo_key(C)->sort key of halo cells ; returns slots arrays"""
    hc = halo_key(C)
    order = np.lexsort((np.arange(len(P)), hc))
    return order

def make(halo_order='xyz'):
    # halo slot order: cell index with given axis order (last = fastest)
    ax = {'x':0,'y':1,'z':2}
    a,b,c_ = [ax[ch] for ch in halo_order]
    hc = (C[:,a]*6 + C[:,b])*6 + C[:,c_]
    order = np.lexsort((np.arange(len(P)), hc))
    Ps = P[order]; Cs = C[order]; hcs = hc[order]
    return Ps, Cs, hcs, (a,b,c_)

def lists_for(Ps, Cs, hcs, axes, own_order='xyz'):
    a,b,c_ = axes
    own = ((Cs >= 1) & (Cs <= 4)).all(1)
    own_idx = np.nonzero(own)[0]
    ax = {'x':0,'y':1,'z':2}
    oa,ob,oc = [ax[ch] for ch in own_order]
    key = ((Cs[own_idx,oa]-1)*4 + (Cs[own_idx,ob]-1))*4 + (Cs[own_idx,oc]-1)
    own_idx = own_idx[np.lexsort((own_idx, key))]
    cstart = np.searchsorted(hcs, np.arange(217))
    rows = []   # per particle: list of 9 lists (hits per row), rows ordered (da, db)
    for i in own_idx:
        c = Cs[i]; R = []
        for da in (-1,0,1):
            for db in (-1,0,1):
                row = ((c[a]+da)*6 + (c[b]+db))*6 + (c[c_]-1)
                bb, e = cstart[row], cstart[row+3]
                cand = np.arange(bb, e)
                d2 = ((Ps[cand] - Ps[i])**2).sum(1)
                R.append(cand[d2 <= h*h].tolist())
        rows.append(R)
    return own_idx, rows

def cycles(seqs, selfs, nb=32, sigma=lambda s: s):
    tot = 0; ideal = 0
    for s0 in range(0, len(seqs), 64):
        sl = [seqs[s0+l] for l in range(min(64, len(seqs)-s0))]
        K = max(len(x) for x in sl); K += K & 1
        sl = [x + [selfs[s0+l]]*(K-len(x)) for l, x in enumerate(sl)]
        for k in range(K):
            for G in (range(0,32), range(32,64)):
                slots = set(sigma(sl[l][k]) for l in G if l < len(sl))
                if not slots: continue
                tot += np.bincount([s % nb for s in slots], minlength=nb).max(); ideal += 1
    return tot/ideal

def flat(rows, rowperm=lambda l: range(9), within=lambda L,l: L):
    out = []
    for l, R in enumerate(rows):
        s = []
        for k in rowperm(l % 64):
            s.extend(within(R[k], l % 64))
        out.append(s)
    return out

for halo_order, own_order in (('xyz','xyz'),):
    Ps, Cs, hcs, axes = make(halo_order)
    own_idx, rows = lists_for(Ps, Cs, hcs, axes, own_order)
    selfs = list(own_idx)
    lens = np.array([sum(len(x) for x in R) for R in rows])
    print(f"halo {halo_order} own {own_order}: S={len(Ps)} own={len(own_idx)} mean contacts {lens.mean():.2f} max {lens.max()}")
    base = flat(rows)
    print("  baseline                          ", round(cycles(base, selfs),3), " (b128-style 16-lane groups n/a)")
    print("  rows rotated by lane              ", round(cycles(flat(rows, lambda l: [(k+l)%9 for k in range(9)]), selfs),3))
    print("  rows rotated by lane>>3 (own cell)", round(cycles(flat(rows, lambda l: [(k+(l>>3))%9 for k in range(9)]), selfs),3))
    print("  rows reversed for odd own cell    ", round(cycles(flat(rows, lambda l: list(range(9)) if ((l>>3)&1)==0 else list(range(8,-1,-1))), selfs),3))
    print("  within-row descending for odd lane", round(cycles(flat(rows, within=lambda L,l: L if (l&1)==0 else L[::-1]), selfs),3))
    # fully sorted by (slot mod 32 - lane) mod 32
    def rotsort(R, l):
        allx = [x for row in R for x in row]
        return sorted(allx, key=lambda x: ((x - l) % 32, x))
    print("  sort by (slot-lane) mod 32        ", round(cycles([rotsort(R, l%64) for l,R in enumerate(rows)], selfs),3))
    def rotsort2(R, l):
        allx = [x for row in R for x in row]
        return sorted(allx, key=lambda x: ((x % 32 - (l%32)) % 32, x))
    # random order (reference point)
    rs = []
    for R in rows:
        allx = [x for row in R for x in row]; rng.shuffle(allx); rs.append(list(allx))
    print("  random order                      ", round(cycles(rs, selfs),3))
    # slot swizzles
    for name, sg in (("xor s>>5", lambda s: (s & ~31) | ((s ^ (s >> 5)) & 31)),
                     ("add 5*(s>>5)", lambda s: (s & ~31) | ((s + 5*(s >> 5)) & 31)),
                     ("add 11*(s>>5)", lambda s: (s & ~31) | ((s + 11*(s >> 5)) & 31)),):
        print(f"  swizzle {name:24s}", round(cycles(base, selfs, sigma=sg),3))
    # lockstep by row (pad each row to the slice max) -> trip count inflation
    def lockstep(rows, selfs):
        out = []; 
        for s0 in range(0, len(rows), 64):
            nl = min(64, len(rows)-s0)
            seq = [[] for _ in range(nl)]
            for k in range(9):
                m = max(len(rows[s0+l][k]) for l in range(nl))
                for l in range(nl):
                    x = rows[s0+l][k]
                    seq[l].extend(x + [selfs[s0+l]]*(m-len(x)))
            out.extend(seq)
        return out
    ls = lockstep(rows, selfs)
    print("  lockstep by row                   ", round(cycles(ls, selfs),3), " trip inflation", round(np.mean([len(x) for x in ls])/ (np.mean([max(lens[s0:s0+64]) for s0 in range(0,len(lens),64)])),3))
