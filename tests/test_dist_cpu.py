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
