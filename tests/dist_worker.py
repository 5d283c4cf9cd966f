"""Worker of tests/test_dist_cpu.py::test_two_rank_protocol_gloo — run under torch.distributed.run with the gloo
backend, world_size 2, on CPU.  Each rank plays one slab: it runs the host mirror of the migration / ghost-plane
protocol (salva_amd/dist.py, the rules of csrc/dist.hip) with real point-to-point messages, then checks with the CPU
oracle that its local set (owned + ghosts) gives every owned particle exactly the contacts the undivided domain gives
it, and that the all-reduced convergence sums are identical on both ranks.  Prints "OK <rank>" on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from salva_amd import dist, scenes  # noqa: E402

R, SF = 0.025, 2.0
H = R * SF * 2


def sendrecv_rows(rows_lo, rows_hi, rank, world):
    """Exchange float64 row blocks with rank-1 / rank+1 (sizes first, then payload); returns (from_lo, from_hi)."""
    out = [np.zeros((0, rows_lo.shape[1])), np.zeros((0, rows_lo.shape[1]))]
    for side, peer, rows in ((0, rank - 1, rows_lo), (1, rank + 1, rows_hi)):
        if peer < 0 or peer >= world:
            continue
        cnt_out = torch.tensor([len(rows)], dtype=torch.int64)
        cnt_in = torch.zeros(1, dtype=torch.int64)
        payload = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float64))
        # lower rank sends first: no deadlock with blocking gloo send/recv
        if rank < peer:
            td.send(cnt_out, peer); td.recv(cnt_in, peer)
            if len(rows): td.send(payload, peer)
            buf = torch.zeros((int(cnt_in), rows.shape[1]), dtype=torch.float64)
            if int(cnt_in): td.recv(buf, peer)
        else:
            td.recv(cnt_in, peer); td.send(cnt_out, peer)
            buf = torch.zeros((int(cnt_in), rows.shape[1]), dtype=torch.float64)
            if int(cnt_in): td.recv(buf, peer)
            if len(rows): td.send(payload, peer)
        out[side] = buf.numpy()
    return out


def main():
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    assert world == 2

    # the same global scene on both ranks; a deliberately stale assignment so that phase 1 has something to migrate
    pos, bpos = scenes.tank(24, 6, 6, R, wall_cells=2)
    pos = scenes.jitter(pos, 0.3 * R, seed=9)
    gid = np.arange(len(pos))
    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, world)
    stale_owner = dist.owner_of(dist.cell_x(pos - np.array([0.6 * H, 0, 0], np.float32), H), slabs)
    mine = stale_owner == rank
    rows = np.column_stack([gid[mine], pos[mine].astype(np.float64)])

    # phase 1: migration
    has_lo, has_hi = rank > 0, rank + 1 < world
    keep, to_lo, to_hi = dist.select_migration(dist.cell_x(rows[:, 1:4], H), slabs[rank], has_lo, has_hi)
    from_lo, from_hi = sendrecv_rows(rows[to_lo], rows[to_hi], rank, world)
    owned = np.vstack([rows[keep], from_lo, from_hi])
    n_moved = torch.tensor([int(to_lo.sum() + to_hi.sum())])
    td.all_reduce(n_moved)
    assert int(n_moved) > 0, "the stale assignment was meant to force migration"
    ocx = dist.cell_x(owned[:, 1:4], H)
    assert (dist.owner_of(ocx, slabs) == rank).all(), "after phase 1 every particle sits on its owner"
    total = torch.tensor([len(owned)])
    td.all_reduce(total)
    assert int(total) == len(pos)

    # phase 2: ghost planes
    g_lo, g_hi = dist.select_ghost_planes(ocx, slabs[rank], has_lo, has_hi)
    assert not (g_lo & g_hi).any(), "a particle is mirrored to at most one neighbour (slabs are >= 4 planes thick)"
    ghosts_lo, ghosts_hi = sendrecv_rows(owned[g_lo], owned[g_hi], rank, world)
    local = np.vstack([owned, ghosts_lo, ghosts_hi])

    # the local set reproduces the undivided domain's contacts for every owned particle
    bsub = dist.boundary_subset(bpos, H, slabs[rank], rank, world)

    def counts(fluid_pos, boundary_pos):
        w = O.OracleWorld(R, SF, O.DFSPH)
        f = w.add_fluid(np.asarray(fluid_pos, np.float32), 1000.0)
        w.add_boundary(np.asarray(boundary_pos, np.float32))
        w.step(1e-4, (0.0, 0.0, 0.0))
        return (w.contact_counts(f), w.contact_counts(f, True), w.fluid_scalar(f, "densities"), w.boundary_volumes(0))

    gff, gfb, grho, gvol = counts(pos, bpos)
    lff, lfb, lrho, lvol = counts(local[:, 1:4], bpos[bsub])
    og = owned[:, 0].astype(int)
    no = len(owned)
    assert (lff[:no] == gff[og]).all(), "fluid-fluid contact counts differ from the undivided domain"
    assert (lfb[:no] == gfb[og]).all(), "fluid-boundary contact counts differ from the undivided domain"
    np.testing.assert_allclose(lrho[:no], grho[og], rtol=1e-5)
    # two planes are mirrored although the interaction range is one: the INNER ghost plane then has its whole
    # neighbourhood here too, so whatever a pass computes for it locally is what its owner computes
    ghosts = local[no:]
    gcx = dist.cell_x(ghosts[:, 1:4], H)
    inner = (gcx == slabs[rank][0] - 1) | (gcx == slabs[rank][1] + 1)
    assert inner.any() and (~inner).any()
    gi = ghosts[inner, 0].astype(int)
    li = no + np.nonzero(inner)[0]
    assert (lff[li] == gff[gi]).all() and (lfb[li] == gfb[gi]).all(), "inner ghost plane: incomplete neighbourhoods"
    np.testing.assert_allclose(lrho[li], grho[gi], rtol=1e-5)
    # boundary particles an owned fluid particle can touch (<= 1 plane away) have the undivided domain's volume
    bcx = dist.cell_x(bpos[bsub], H)
    lo = -(1 << 60) if rank == 0 else slabs[rank][0] - 1
    hi = (1 << 60) if rank == world - 1 else slabs[rank][1] + 1
    near = (bcx >= lo) & (bcx <= hi)
    np.testing.assert_allclose(lvol[near], gvol[bsub][near], rtol=1e-5)

    # the convergence test is global: same all-reduced sums, hence the same decision, on both ranks
    err = np.maximum(lrho[:no] / 1000.0 - 1.0, 0.0).sum()
    s = torch.tensor([err, float(no)], dtype=torch.float64)
    td.all_reduce(s)
    mean_err = float(s[0] / s[1])
    gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    td.all_gather(gathered, torch.tensor([mean_err], dtype=torch.float64))
    assert all(float(g) == mean_err for g in gathered)
    np.testing.assert_allclose(mean_err, np.maximum(grho / 1000.0 - 1.0, 0.0).mean(), rtol=1e-5, atol=1e-12)

    # creation and removal in a running decomposed world are collective (salva_hip_add_particles / salva_hip_delete_owned): the
    # new ids continue after the largest id, rank by rank; every rank drops the listed ids it owns; the per-fluid count the error
    # averages divide by is all-reduced, so both ranks end up with the same total and no id exists twice
    n_add = 3 + 2 * rank
    adds = torch.zeros(world, dtype=torch.int64)
    adds[rank] = n_add
    td.all_reduce(adds)
    first, gid_next = dist.allocate_new_ids(adds.tolist(), rank, len(pos))
    my_ids = np.concatenate([owned[:, 0].astype(np.int64), first + np.arange(n_add)])
    doomed = np.concatenate([np.arange(0, len(pos), 7), [len(pos) + 1, len(pos) + 4]])  # old ids of both ranks + two new ones
    keep_mask = dist.apply_owned_deletions(my_ids, doomed)
    delta = torch.tensor([n_add - int((~keep_mask).sum())], dtype=torch.int64)
    td.all_reduce(delta)
    my_ids = my_ids[keep_mask]
    total_after = torch.tensor([len(my_ids)], dtype=torch.int64)
    td.all_reduce(total_after)
    expect = len(pos) + int(adds.sum()) - len(doomed)
    assert int(total_after) == expect == len(pos) + int(delta), (int(total_after), expect, int(delta))
    assert gid_next == len(pos) + int(adds.sum())
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    td.all_gather(sizes, torch.tensor([len(my_ids)], dtype=torch.int64))
    padded = torch.full((max(int(x) for x in sizes),), -1, dtype=torch.int64)
    padded[:len(my_ids)] = torch.from_numpy(my_ids)
    everyone = [torch.zeros_like(padded) for _ in range(world)]
    td.all_gather(everyone, padded)
    allids = np.concatenate([e.numpy()[:int(n)] for e, n in zip(everyone, sizes)])
    assert len(np.unique(allids)) == len(allids) == expect, "an id exists on two ranks or was lost"
    assert not np.isin(doomed, allids).any()

    # DynamicContactSampling is collective (World::dist_gather_emitted): a rank projects everything it holds — a ghost is pushed out
    # of the collider exactly like its owner — emits for the particles it OWNS, and the ranks assemble one table, the same on
    # every rank, by adding zero-padded sections as 64-bit integers.  Here: a ball on the cut; the oracle's DynamicContactSampling
    # arm on each rank's local set and on the undivided domain; the summed table must be the concatenation in rank order, and as a
    # set of (source particle, point) it must be the undivided domain's, bit for bit.
    ball_t = np.float32([(slabs[0][1] + 1) * H, 0.05, 0.0])
    ball_r, dt_prev = 0.12, 0.004

    def emitted(fluid_pos, fluid_vel):
        w = O.OracleWorld(R, SF, O.DFSPH)
        f = w.add_fluid(np.asarray(fluid_pos, np.float32), 1000.0, np.asarray(fluid_vel, np.float32))
        b = w.add_boundary(np.zeros((0, 3), np.float32))
        w.set_boundary_dynamic_sampling(b, 1, [ball_r])
        w.update_boundary_pose(b, ball_t, np.float32([0, 0, 0, 1]), np.float32([0.2, 0.0, 0.1]), np.float32([0.0, 1.0, 0.0]), ball_t, True, False)
        w.set_timestep(dt_prev, 1.0 / dt_prev)
        w.step(1e-4, (0.0, 0.0, 0.0))
        _, src = w.boundary_sources(b)
        return np.asarray(src, np.int64), w.boundary_vec(b, "positions").astype(np.float32)

    vel_all = scenes.random_velocities(len(pos), 1.5, seed=4)  # (the prediction x + v dt decides who emits)
    gsrc, gpts = emitted(pos, vel_all)
    assert len(gsrc) > 30, "the ball was meant to sit in the fluid"
    lsrc, lpts = emitted(local[:, 1:4], vel_all[local[:, 0].astype(int)])
    mine_e = lsrc < no  # (ghosts emit on their owner's rank)
    my_gids = owned[lsrc[mine_e], 0].astype(np.uint32)
    counts_e = torch.zeros(world, dtype=torch.int64)
    counts_e[rank] = int(mine_e.sum())
    td.all_reduce(counts_e)
    total_e, before_e = int(counts_e.sum()), int(counts_e[:rank].sum())
    assert int(counts_e.min()) > 0, "the ball was meant to straddle the cut"
    words = dist.pack_emitted(lpts[mine_e], my_gids, np.zeros(len(my_gids), np.uint32), before_e, total_e)
    assert len(words) == dist.emitted_table_words(total_e)
    tw = torch.from_numpy(words.view(np.int64).copy())
    td.all_reduce(tw)  # (two's-complement addition: the same bits as the device's unsigned sum)
    pts_all, src_all, fl_all = dist.unpack_emitted(tw.numpy().view(np.uint64), total_e)
    assert np.array_equal(src_all[before_e:before_e + len(my_gids)], my_gids) and not fl_all.any()
    assert np.array_equal(pts_all[before_e:before_e + len(my_gids)].view(np.uint32), lpts[mine_e].view(np.uint32))
    tables = [torch.zeros_like(tw) for _ in range(world)]
    td.all_gather(tables, tw)
    assert all(torch.equal(t, tw) for t in tables), "every rank holds the same table"
    assert len(np.unique(src_all)) == total_e == len(gsrc), (total_e, len(gsrc))
    go, ao = np.argsort(gsrc), np.argsort(src_all)
    assert np.array_equal(gsrc[go], src_all[ao].astype(np.int64))
    assert np.array_equal(gpts[go].view(np.uint32), pts_all[ao].view(np.uint32)), "an emitted point differs from the undivided domain's"

    td.barrier()
    print(f"OK {rank}", flush=True)
    td.destroy_process_group()


if __name__ == "__main__":
    main()
