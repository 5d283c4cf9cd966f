"""Worker of tests/test_rccl_multi_gpu.py — run under torch.distributed.run, one rank per GPU, backend nccl (= RCCL on ROCm).
Every rank steps its x-slab of one tank scene through the RCCL transport (comm.hip: grouped ncclSend/ncclRecv with the two
neighbours + the all-reduced convergence test); rank 0 also steps the undivided domain on its own GPU and compares.  Prints
"RCCL_OK <rank>" on success."""
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
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.single_node_rccl_env()
    td.init_process_group("nccl", device_id=torch.device("cuda", local))
    nsteps = 8
    # a block 20 cells long per rank, drifting towards +x: particles change owner during the run
    pos, bpos = scenes.tank(40 * world, 10, 10, R, wall_cells=4)
    pos = scenes.jitter(pos, 0.1 * R, 5).astype(np.float32)
    vel = scenes.random_velocities(len(pos), 0.5, 6).astype(np.float32)
    vel[:, 0] += 2.0
    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, world)
    owner = dist.owner_of(cx, slabs)
    offsets = np.concatenate([[0], np.cumsum([(owner == r).sum() for r in range(world)])])

    idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(dist.Comm.unique_id()), dtype=torch.uint8))
    td.broadcast(idt, 0)
    comm = dist.Comm.rccl(rank, world, bytes(idt.cpu().numpy().tobytes()), local)

    w = LiquidWorld(DFSPHSolver(), R, SF, device=local)
    mine = np.nonzero(owner == rank)[0]
    f = Fluid(pos[mine], R, 1000.0)
    f.velocities = vel[mine]
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[rank], rank, world)]))
    w.set_domain(comm, slabs[rank][0], slabs[rank][1], int(offsets[rank]))
    stats = [w.step(DT, G) for _ in range(nsteps)]
    gid, p, v, _slot = w.owned()
    iters = [(s.n_divergence_iters, s.n_pressure_iters) for s in stats]

    gathered = [None] * world
    td.all_gather_object(gathered, (gid, p, v, iters))
    # the convergence test is global: every rank took the same iterations
    assert all(g[3] == iters for g in gathered), [g[3] for g in gathered]
    if rank == 0:
        ref = LiquidWorld(DFSPHSolver(), R, SF, device=local)
        fr = Fluid(pos, R, 1000.0)
        fr.velocities = vel
        fr.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
        ref.add_fluid(fr)
        ref.add_boundary(Boundary(bpos))
        ref_stats = [ref.step(DT, G) for _ in range(nsteps)]
        # global id = rank-major upload order
        order = np.concatenate([np.nonzero(owner == r)[0] for r in range(world)])
        got_p = np.full_like(pos, np.nan)
        seen = np.zeros(len(pos), int)
        for g, pp, vv, _ in gathered:
            got_p[order[g]] = pp
            np.add.at(seen, order[g], 1)
        assert (seen == 1).all(), f"{(seen != 1).sum()} particles lost or duplicated"
        moved = (dist.owner_of(dist.cell_x(got_p, H), slabs) != owner).sum()
        assert moved > 0 or world == 1, "the scene was meant to exercise migration over RCCL"
        dp = np.abs(got_p - fr.positions).max()
        assert dp < 2e-4 * H, f"positions differ from the undivided domain by {dp / H:.2e} h"
        same = sum((s.n_divergence_iters, s.n_pressure_iters) == it for s, it in zip(ref_stats, iters))
        assert same >= nsteps - 2, "iteration counts drifted from the undivided domain"
    td.barrier()
    del w
    comm.destroy()
    print(f"RCCL_OK {rank}", flush=True)
    td.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:  # noqa: BLE001 - die at once so that the launcher tears the other ranks down
        import traceback

        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)
