"""Worker of tests/test_peer_transport_gpu.py — run under torch.distributed.run with the gloo backend (rendezvous and result
gathering only), one process per rank; the ranks share the box's GPUs round-robin, so two ranks on ONE GPU is a valid run (the
peer-direct transport maps the other process's window through hipIpc whichever device it lives on).  Every rank first runs the
transport's collective self-test with slots shorter than the messages, then steps its x-slab of one tank scene through the
transport; rank 0 also steps the undivided domain and compares.  Prints "PEER_OK <rank>" on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, dist, scenes  # noqa: E402

R, SF = 0.025, 2.0
H = R * SF * 2
DT = 1.0 / 200.0
G = (0.0, -9.81, 0.0)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    torch.cuda.set_device(device)
    td.init_process_group("gloo")

    def gather(handle):
        out = [None] * world
        td.all_gather_object(out, handle)
        return out

    # ---- the transport by itself: 64 KB slots, messages up to 300 000 bytes (five rounds per exchange), odd and empty lengths
    small = dist.Comm.peer(rank, world, device, gather, slot_bytes=64 << 10)
    small.selftest(300_000, 6)
    td.barrier()
    for nbytes in (4 << 10, 64 << 10):  # a refresh of 1 000 / 16 000 ghosts' scalar field
        us_x, us_r = small.time(nbytes, 200)
        print(f"PEER_TIMING rank {rank}/{world} on device {device}: exchange of {nbytes} B each way {us_x:.1f} us, all-reduce {us_r:.1f} us", flush=True)
    td.barrier()
    small.destroy()

    # ---- a decomposed run over it
    nsteps = 8
    pos, bpos = scenes.tank(40 * world, 10, 10, R, wall_cells=4)
    pos = scenes.jitter(pos, 0.1 * R, 5).astype(np.float32)
    vel = scenes.random_velocities(len(pos), 0.5, 6).astype(np.float32)
    vel[:, 0] += 2.0  # drifting towards +x: particles change owner during the run
    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, world)
    owner = dist.owner_of(cx, slabs)
    offsets = np.concatenate([[0], np.cumsum([(owner == r).sum() for r in range(world)])])
    comm = dist.Comm.peer(rank, world, device, gather)

    w = LiquidWorld(DFSPHSolver(), R, SF, device=device)
    mine = np.nonzero(owner == rank)[0]
    f = Fluid(pos[mine], R, 1000.0)
    f.velocities = vel[mine]
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[rank], rank, world)]))
    w.set_domain(comm, slabs[rank][0], slabs[rank][1], int(offsets[rank]))
    td.barrier()
    stats = [w.step(DT, G) for _ in range(nsteps)]
    gid, p, v, _slot = w.owned()
    iters = [(s.n_divergence_iters, s.n_pressure_iters) for s in stats]

    gathered = [None] * world
    td.all_gather_object(gathered, (gid, p, v, iters))
    assert all(g[3] == iters for g in gathered), [g[3] for g in gathered]  # the convergence test is global
    if rank == 0:
        ref = LiquidWorld(DFSPHSolver(), R, SF, device=device)
        fr = Fluid(pos, R, 1000.0)
        fr.velocities = vel
        fr.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
        ref.add_fluid(fr)
        ref.add_boundary(Boundary(bpos))
        ref_stats = [ref.step(DT, G) for _ in range(nsteps)]
        order = np.concatenate([np.nonzero(owner == r)[0] for r in range(world)])  # global id = rank-major upload order
        got_p = np.full_like(pos, np.nan)
        seen = np.zeros(len(pos), int)
        for g, pp, vv, _ in gathered:
            got_p[order[g]] = pp
            np.add.at(seen, order[g], 1)
        assert (seen == 1).all(), f"{(seen != 1).sum()} particles lost or duplicated"
        moved = (dist.owner_of(dist.cell_x(got_p, H), slabs) != owner).sum()
        assert moved > 0, "the scene was meant to exercise migration"
        dp = np.abs(got_p - fr.positions).max()
        assert dp < 2e-4 * H, f"positions differ from the undivided domain by {dp / H:.2e} h"
        same = sum((s.n_divergence_iters, s.n_pressure_iters) == it for s, it in zip(ref_stats, iters))
        assert same >= nsteps - 2, "iteration counts drifted from the undivided domain"
    td.barrier()
    del w
    comm.destroy()
    print(f"PEER_OK {rank}", flush=True)
    td.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:  # noqa: BLE001 - die at once so that the launcher tears the other ranks down
        import traceback

        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)
