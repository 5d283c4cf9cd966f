"""Host side of the multi-GPU slab decomposition, on CPU: the partition helpers (numpy) and the two-rank exchange
protocol over torch.distributed's gloo backend (tests/dist_worker.py).  The device side is tests/test_dist_gpu.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from salva_amd import dist, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, H = 0.025, 0.1


def test_split_slabs_covers_and_balances():
    rng = np.random.default_rng(3)
    cx = rng.integers(-7, 40, size=20000)
    for nranks in (1, 2, 3, 5, 8):
        slabs = dist.split_slabs(cx, nranks)
        assert len(slabs) == nranks
        assert slabs[0][0] == cx.min() and slabs[-1][1] == cx.max()
        for (lo, hi), (lo2, _hi2) in zip(slabs, slabs[1:]):
            assert lo2 == hi + 1
        assert all(hi - lo + 1 >= 4 for lo, hi in slabs)
        owner = dist.owner_of(cx, slabs)
        for r, (lo, hi) in enumerate(slabs):
            assert ((cx >= lo) & (cx <= hi) == (owner == r)).all()
        counts = np.bincount(owner, minlength=nranks)
        assert counts.max() <= 1.35 * len(cx) / nranks + 1


def test_split_slabs_skewed_and_degenerate():
    # almost everything in one plane: the cuts still leave every slab four planes (two mirrored planes per face)
    cx = np.concatenate([np.full(1000, 3), np.arange(0, 16)])
    slabs = dist.split_slabs(cx, 3)
    assert all(hi - lo + 1 >= 4 for lo, hi in slabs) and slabs[0][0] == 0 and slabs[-1][1] == 15
    with pytest.raises(ValueError):
        dist.split_slabs(np.arange(11), 3)  # 11 planes cannot hold 3 slabs of 4
    with pytest.raises(ValueError):
        dist.split_slabs(np.array([], int), 2)
    # beyond the ends: open-ended ownership
    assert dist.owner_of(np.array([-100, 100]), slabs).tolist() == [0, 2]


def test_cell_x_is_f32_floor():
    x = np.array([[-0.1, 0, 0], [-1e-9, 0, 0], [0.0, 0, 0], [0.099999994, 0, 0], [0.1, 0, 0], [0.30000001, 0, 0]], np.float32)
    ref = np.floor(x[:, 0] / np.float32(0.1)).astype(int)
    assert dist.cell_x(x, 0.1).tolist() == ref.tolist()
    assert dist.cell_x(x, 0.1)[0] in (-1, -2) and dist.cell_x(x, 0.1)[2] == 0


def test_emitted_table_sections_add_up_to_the_concatenation():
    """World::dist_gather_emitted's packing (dist.pack_emitted / unpack_emitted): zero-padded sections of (point, source id) rows and
    32-bit fluid indices, added as 64-bit integers, are the concatenation in rank order — also when two ranks share a 64-bit word
    of the fluid indices (odd counts) and when a rank has nothing."""
    rng = np.random.default_rng(5)
    counts = [3, 0, 5, 1]
    total = sum(counts)
    parts, acc = [], np.zeros(dist.emitted_table_words(total), np.uint64)
    for r, k in enumerate(counts):
        pts = rng.normal(size=(k, 3)).astype(np.float32)
        pts[:1] *= -1e-30  # (sign bits and tiny exponents survive the integer addition)
        ids = rng.integers(0, 2 ** 32, size=k, dtype=np.uint64).astype(np.uint32)
        fl = rng.integers(0, 2 ** 32, size=k, dtype=np.uint64).astype(np.uint32)
        parts.append((pts, ids, fl))
        acc = acc + dist.pack_emitted(pts, ids, fl, sum(counts[:r]), total)  # uint64 addition wraps like the device's
    pts, ids, fl = dist.unpack_emitted(acc, total)
    assert np.array_equal(pts.view(np.uint32), np.concatenate([p[0] for p in parts]).view(np.uint32))
    assert np.array_equal(ids, np.concatenate([p[1] for p in parts])) and np.array_equal(fl, np.concatenate([p[2] for p in parts]))


def test_selection_rules():
    cx = np.arange(-2, 12)
    slab = (3, 6)
    keep, lo, hi = dist.select_migration(cx, slab, True, True)
    assert cx[keep].tolist() == [3, 4, 5, 6] and cx[lo].max() == 2 and cx[hi].min() == 7
    keep, lo, hi = dist.select_migration(cx, slab, False, True)   # first rank: open towards -x
    assert cx[keep].min() == -2 and not lo.any()
    glo, ghi = dist.select_ghost_planes(np.array([3, 4, 5, 6]), slab, True, True)
    assert glo.tolist() == [True, True, False, False] and ghi.tolist() == [False, False, True, True]
    glo, ghi = dist.select_ghost_planes(np.array([1, 3, 6, 9]), slab, False, True)  # open end holds strays, mirrors none there
    assert not glo.any() and ghi.tolist() == [False, False, True, True]


def test_boundary_subset_closes_the_neighbourhoods():
    """Every boundary particle an owned fluid particle can touch is in the rank's subset, and so is every boundary
    neighbour of such a particle (so that its volume is the undivided domain's)."""
    pos, bpos = scenes.tank(40, 5, 5, R, wall_cells=3)
    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, 3)
    owner = dist.owner_of(cx, slabs)
    for r in range(3):
        sub = np.zeros(len(bpos), bool)
        sub[dist.boundary_subset(bpos, H, slabs[r], r, 3)] = True
        mine = pos[owner == r]
        d2 = ((mine[:, None, :] - bpos[None, :, :]) ** 2).sum(-1)
        touched = (d2 <= H * H).any(0)
        assert sub[touched].all()
        tb = bpos[touched]
        d2b = ((tb[:, None, :] - bpos[None, :, :]) ** 2).sum(-1)
        assert sub[(d2b <= H * H).any(0)].all()
        if 0 < r < 2:
            assert sub.sum() < len(bpos)  # an inner rank holds a strict subset


def test_bench_slab_scene_is_consistent():
    sys.path.insert(0, ROOT)
    import bench

    world, side = 3, 12
    seen = []
    for r in range(world):
        fluid, shell, slab, nshell = bench.build_slab_scene(side, r, world)
        cx = dist.cell_x(fluid, H)
        # everything a rank uploads is inside its slab or at most one plane outside (first-step migration handles that)
        assert cx.min() >= slab[0] - 1 and cx.max() <= slab[1] + 1
        assert slab[1] - slab[0] + 1 >= 4
        seen.append(slab)
        assert len(shell) <= nshell
    for a, b in zip(seen, seen[1:]):
        assert b[0] == a[1] + 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_protocol_gloo(oracle_lib):
    """world_size-2 gloo run of the migration + ghost-plane protocol, checked against the oracle on the undivided domain."""
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    assert "OK 0" in p.stdout and "OK 1" in p.stdout


def test_peer_transport_protocol_model():
    """The ordering argument of salva_amd/csrc/comm_peer.hip, as a model: every rank runs the same sequence of operations in order
    on one stream — an exchange is put (write my message into the neighbour's slot of parity seq & 1, then raise its flag to seq)
    followed by get (wait for my flags >= seq, read my slots of that parity); an all-reduce writes my row of parity seq & 1 into
    every rank's mailbox, raises my flag there, waits for all flags in my own window and reads the rows.  Nothing acknowledges
    a read.  Under every interleaving of the ranks' streams a read must find exactly the message of its own sequence number —
    i.e. two slots per direction (and per mailbox row) are enough.  A random scheduler runs the ranks; a get / reduce whose flags
    are not up blocks its rank; writes land word by word with other ranks' steps in between (a torn message is visible to a reader
    that does not wait)."""
    rng = np.random.default_rng(11)
    for trial in range(60):
        nranks = int(rng.integers(2, 6))
        nops = int(rng.integers(5, 40))
        ops = [("x" if rng.random() < 0.6 else "r") for _ in range(nops)]  # the same program on every rank
        # windows: msg[rank][side][parity] = [seq written, words complete], flag[rank][side]; red[rank][parity][src], rflag[rank][src]
        msg = [[[[0, True] for _ in range(2)] for _ in range(2)] for _ in range(nranks)]
        flag = [[0, 0] for _ in range(nranks)]
        red = [[[[0, True] for _ in range(nranks)] for _ in range(2)] for _ in range(nranks)]
        rflag = [[0] * nranks for _ in range(nranks)]

        def program(r):
            xseq = [0, 0]  # per link: [with the lower, with the upper neighbour]
            rseq = 0
            for op in ops:
                if op == "x":
                    links = [(0, r - 1)] * (r > 0) + [(1, r + 1)] * (r + 1 < nranks)
                    for l, _nb in links:
                        xseq[l] += 1
                    for l, nb in links:  # put: the two halves of the message land at different times
                        slot = msg[nb][1 - l][xseq[l] & 1]
                        slot[0], slot[1] = xseq[l], False
                        yield
                        slot[1] = True
                        yield
                        flag[nb][1 - l] = xseq[l]
                    for l, _nb in links:  # get
                        while flag[r][l] < xseq[l]:
                            yield
                        slot = msg[r][l][xseq[l] & 1]
                        assert slot == [xseq[l], True], f"rank {r} link {l}: expected message {xseq[l]}, slot holds {slot}"
                        yield
                        assert slot == [xseq[l], True], f"rank {r} link {l}: message {xseq[l]} overwritten while being read ({slot})"
                else:
                    rseq += 1
                    for dst in range(nranks):
                        row = red[dst][rseq & 1][r]
                        row[0], row[1] = rseq, False
                        yield
                        row[1] = True
                    yield
                    for dst in range(nranks):
                        rflag[dst][r] = rseq
                    for src in range(nranks):
                        while rflag[r][src] < rseq:
                            yield
                    for src in range(nranks):
                        assert red[r][rseq & 1][src] == [rseq, True], f"rank {r}: all-reduce {rseq}, row of rank {src} holds {red[r][rseq & 1][src]}"
                        yield
                        assert red[r][rseq & 1][src] == [rseq, True], f"rank {r}: all-reduce {rseq}, row of rank {src} overwritten while being read"

        live = {r: program(r) for r in range(nranks)}
        spins = 0
        while live:
            r = int(rng.choice(sorted(live)))
            if rng.random() < 0.3:  # bursts: one rank runs far ahead when it can
                steps = int(rng.integers(1, 200))
            else:
                steps = 1
            for _ in range(steps):
                try:
                    next(live[r])
                except StopIteration:
                    del live[r]
                    break
            spins += 1
            assert spins < 2_000_000, "deadlock in the model"
