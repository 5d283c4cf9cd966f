"""Worker of tests/test_config5_processes_gpu.py — BASELINE config 5 in its literal geometry between REAL processes: run under
torch.distributed.run with the gloo backend (rendezvous and result gathering only), eight ranks, one block of SIDE^3 particles
each (bench.slab_block), the ranks sharing the box's GPUs round-robin and exchanging through the xGMI peer-direct transport
(comm_peer.hip).  Rank 0 then steps the undivided world of 8 x SIDE^3 particles and compares as tests/test_config5_gpu.py does
for the loopback threads.  Prints "CONFIG5_OK" from rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, dist  # noqa: E402

SIDE = int(os.environ.get("SALVA_CONFIG5_SIDE", "100"))
NSTEPS = int(os.environ.get("SALVA_CONFIG5_STEPS", "24"))
H = 4.0 * bench.R


def _fluid(pos):
    f = Fluid(pos, bench.R, 1000.0)
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    return f


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    torch.cuda.set_device(device)
    td.init_process_group("gloo")

    def gather(handle):
        out = [None] * world
        td.all_gather_object(out, handle)
        return out

    shell, slabs = bench.slab_scene_geometry(SIDE, world)
    block = bench.slab_block(SIDE, rank)
    sizes = [len(bench.slab_block(SIDE, r)) for r in range(world)] if rank == 0 else None
    comm = dist.Comm.peer(rank, world, device, gather)
    w = LiquidWorld(DFSPHSolver(), bench.R, 2.0, device=device)
    w.add_fluid(_fluid(block))
    w.add_boundary(Boundary(shell[dist.boundary_subset(shell, H, slabs[rank], rank, world)]))
    w.set_domain(comm, slabs[rank][0], slabs[rank][1], rank * len(block))
    td.barrier()
    st = [w.step(bench.DT, bench.GRAVITY) for _ in range(NSTEPS)]
    stats = [(int(s.n_divergence_iters), int(s.n_pressure_iters), int(s.ncontacts), int(s.nparticles)) for s in st]
    gid, p, v, _slot = w.owned()
    gathered = [None] * world
    td.all_gather_object(gathered, (gid, p, v, stats))
    td.barrier()
    del w
    comm.destroy()

    if rank == 0:
        n = sum(sizes)
        assert all(s == sizes[0] for s in sizes)
        ref = LiquidWorld(DFSPHSolver(), bench.R, 2.0, device=device)
        f = ref.add_fluid(_fluid(np.concatenate([bench.slab_block(SIDE, r) for r in range(world)])))
        ref.add_boundary(Boundary(shell))
        ref_stats = [ref.step(bench.DT, bench.GRAVITY) for _ in range(NSTEPS)]
        ref_stats = [(int(s.n_divergence_iters), int(s.n_pressure_iters), int(s.ncontacts)) for s in ref_stats]
        ref_p = np.array(f.positions, dtype=np.float32)
        ref_v = np.array(f.velocities, dtype=np.float32)
        got_p = np.full_like(ref_p, np.nan)
        got_v = np.full_like(ref_v, np.nan)
        seen = np.zeros(n, np.int32)
        for g, pp, vv, _ in gathered:
            got_p[g] = pp
            got_v[g] = vv
            np.add.at(seen, g, 1)
        assert (seen == 1).all(), f"{int((seen != 1).sum())} particles lost or duplicated"
        same = 0
        for k in range(NSTEPS):
            its = {(g[3][k][0], g[3][k][1]) for g in gathered}
            assert len(its) == 1, f"step {k}: ranks disagree on iteration counts {its}"
            assert sum(g[3][k][3] for g in gathered) == n, f"step {k}: owned particle counts do not add up"
            same += next(iter(its)) == ref_stats[k][:2]
            tot = sum(g[3][k][2] for g in gathered)
            slack = 0 if k == 0 else max(4, int(1e-6 * ref_stats[k][2]))
            assert abs(tot - ref_stats[k][2]) <= slack, f"step {k}: contacts {tot} over the slabs vs {ref_stats[k][2]} undivided"
        assert same >= NSTEPS - 1, f"iteration counts: {[s[:2] for s in gathered[0][3]]} vs {[s[:2] for s in ref_stats]}"
        dp = float(np.abs(got_p - ref_p).max())
        dv = float(np.abs(got_v - ref_v).max())
        print(f"config 5 between {world} processes over the peer-direct transport: {n} particles, {NSTEPS} steps: max |dx| = {dp / H:.2e} h, "
              f"max |dv| = {dv:.2e} m/s, iterations identical in {same}/{NSTEPS} steps", flush=True)
        assert dp < 2e-4 * H, f"positions differ by {dp / H:.2e} h"
        assert dv < 5e-3, f"velocities differ by {dv:.2e} m/s"
        print("CONFIG5_OK", flush=True)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:  # noqa: BLE001 - die at once so that the launcher tears the other ranks down
        import traceback

        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)
