#!/usr/bin/env python
"""bench.py — particle-steps/s of the LiquidWorld::step hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one `LiquidWorld::step(dt = 1/200, g = -9.81 y)` over the synthetic scene of BASELINE config[1]
("3D DFSPH 1M particles, single fluid, XSPH viscosity, 1xMI355X", concretised in SURVEY.md §8d config 2 (A)):
a 100^3 lattice block (spacing 2r, r = 0.025, h = 0.1, jitter +-0.1 r with LCG seed 42) resting in an open lattice
tank (floor + 4 walls), rho0 = 1000, XSPHViscosity(0.5, 0), DFSPH defaults.  State is resident in HBM when the timed
region starts; the timed region contains everything a step does (cell sort, neighbour lists, all solver passes,
convergence read-backs) and nothing else.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from salva_amd import DFSPHSolver, Fluid, Boundary, LiquidWorld, XSPHViscosity, scenes  # noqa: E402
from salva_amd import dist as slab  # noqa: E402

R = 0.025
DT = 1.0 / 200.0
GRAVITY = (0.0, -9.81, 0.0)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def build_scene(side: int):
    fluid, shell = scenes.tank(side, side, side, R)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
    return fluid, shell


def build_slab_scene(side: int, rank: int, world: int):
    """Weak scaling: `world` copies of the N=1 block side by side along x in one long tank; rank r uploads block r and the
    tank particles near its slab.  The cut between blocks r-1 and r is the cell plane under block r's lower face."""
    d = 2.0 * R
    h = R * 2.0 * 2.0
    fluid = scenes.cube_fluid_positions(side, side, side, R)
    fluid[:, 0] += np.float32(rank * side * d)
    fluid = scenes.jitter(fluid, 0.1 * R, seed=42 + rank)
    fmin = np.array([-side * R + R] * 3, dtype=np.float64)
    fmax = fmin + np.array([world * side - 1, side - 1, side - 1], dtype=np.float64) * d
    mins, maxs = fmin - d, fmax + d
    maxs[1] += max(side // 2, 4) * d
    shell = scenes.box_shell(mins, maxs, R, faces="xXyzZ")
    cuts = [int(np.floor((fmin[0] - 0.5 * d + r * side * d) / h)) for r in range(world + 1)]
    slabs = [(cuts[r], cuts[r + 1] - 1) for r in range(world)]
    mine = slab.boundary_subset(shell, h, slabs[rank], rank, world)
    return fluid, shell[mine], slabs[rank], len(shell)


def make_world(fluid, shell, device: int):
    w = LiquidWorld(DFSPHSolver(), R, 2.0, device=device)
    f = Fluid(fluid, R, 1000.0)
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(shell))
    return w, f


def committed_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*/hbm_traffic.json, written by
    tools/summarize_pmc.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same scene, with the
    gfx950 correction of MI355X_MICROARCH.md).  Counters cannot be read from inside this process, so this is the
    measured figure of the committed profile, not of this run; None if there is none."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "hbm_traffic.json")), reverse=True):
        try:
            k = json.load(open(f))["kernels"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if k:
            return float(k["bytes"]), os.path.relpath(f, ROOT)
    return None, None


def flush_c_stdio():
    """fflush(NULL): push whatever native libraries printf'ed (RCCL's banner) out before Python prints."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def effective_cores() -> int:
    """Host cores this process may actually use: min(visible CPUs, scheduler affinity, cgroup CPU quota).  (The GPU
    boxes show 256 CPUs but cap the container at 16; 256 OpenMP threads on a 16-CPU quota run 10x slower than 16.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(n, 1)


def cpu_baseline(side: int, steps: int, warmup: int):
    """The CPU oracle (C++ restatement of salva's CPU path, kind = "port") on a scaled-down copy of the same scene
    (same spacing, tank, forces, dt, and the same warm-up + step count, so it goes through the same free-fall ->
    impact regimes and iteration counts), all host cores."""
    from oracle import oracle as O

    fluid, shell = build_scene(side)
    cores = effective_cores()
    w = O.OracleWorld(R, 2.0, O.DFSPH, threads=cores)
    fid = w.add_fluid(fluid, 1000.0)
    w.add_xsph(fid, 0.5, 0.0)
    w.add_boundary(shell)
    nd = []
    for _ in range(warmup):
        w.step(DT, GRAVITY)
    t0 = time.perf_counter()
    for _ in range(steps):
        st = w.step(DT, GRAVITY)
        nd.append(st.n_div_iters)
    dt = time.perf_counter() - t0
    return {"value": len(fluid) * steps / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"{warmup}+{steps} steps of the same scene scaled to {side}^3 = {len(fluid)} fluid particles "
                      f"(+{len(shell)} boundary), oracle/salva_oracle.cpp f32, OpenMP {cores} threads, "
                      f"{dt:.1f} s, mean divergence iterations {float(np.mean(nd)):.1f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 5 warm-up + 50 timed steps
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--side", type=int, default=100, help="particles per edge of the fluid block (100 -> 1M)")
    ap.add_argument("--cpu-side", type=int, default=64, help="edge of the scaled-down block the CPU baseline runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-slabs", action="store_true",
                    help="take the decomposed (RCCL transport) code path even with one rank; a self-test aid, not a bench mode")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to measure")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        slab.single_node_rccl_env()  # one node by contract: keep RCCL's bootstrap off the (absent) network
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    comm = None
    decomposed = world > 1 or args.force_slabs
    if not decomposed:
        fluid, shell = build_scene(args.side)
        nshell_total = len(shell)
        w, f = make_world(fluid, shell, local_rank)
    else:
        # slab decomposition along x: RCCL point-to-point with the two neighbours + one tiny all-reduce per convergence test
        fluid, shell, my_slab, nshell_total = build_slab_scene(args.side, rank, world)
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(slab.Comm.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(idt, 0)
        comm = slab.Comm.rccl(rank, world, bytes(idt.cpu().numpy().tobytes()), local_rank)
        w, f = make_world(fluid, shell, local_rank)
        w.set_domain(comm, my_slab[0], my_slab[1], rank * len(fluid))
    n = len(fluid)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    iters = []
    for _ in range(args.warmup):
        w.step(DT, GRAVITY)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = w.step(DT, GRAVITY)
        iters.append((st.n_divergence_iters, st.n_pressure_iters, st.ncontacts, st.grid_ms, st.solver_ms))
    tile_stats = {"max_halo_fluid": int(st.reserved[0]), "max_halo_boundary": int(st.reserved[1]), "tile_threads": int(st.reserved[2]),
                  "ghost_particles": int(st.reserved[4])}
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant neighbour-sum kernel, timed live with HIP events on the world's own stream
    it = np.asarray(iters, dtype=np.float64)
    K = float(it[-1, 2]) / n  # mean directed contacts per fluid particle (ff + fb + bb) / N  ~ list entries per particle
    kernel_us = w.time_pred_density(50)
    if not decomposed:
        kbar = float(w.contact_counts(f).mean() + w.contact_counts(f, True).mean())
    else:  # host-order fields do not exist in a decomposed run: list entries per local particle, from the step report
        kbar = float(st.reserved[3])
    algo_bytes = n * (4.0 * kbar + 52.0)  # k_pred_density moves N (4K + 52) bytes per launch
    achieved = algo_bytes / (kernel_us * 1e-6) / 1e9
    traffic, traffic_src = committed_traffic("k_pred_density") if n == 1000000 else (None, None)
    roofline = {"bound": "hbm", "kernel": "k_pred_density", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel_us": kernel_us,
                "algorithmic_bytes": algo_bytes, "mean_contacts": kbar}

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args.cpu_side, args.steps, args.warmup)
        value = n * world * args.steps / elapsed
        out = {
            "metric": "particle-steps/sec (3D DFSPH)",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"3D DFSPH {n * world} fluid particles (+{nshell_total} boundary), single fluid, XSPH viscosity, "
                            f"lattice block in open tank, r=0.025 h=0.1 dt=1/200",
                "particles_per_gpu": n,
                "parallelism": "single domain" if not decomposed else
                f"{world} x-slabs, one per GPU: RCCL send/recv of two ghost planes per face with the 2 neighbours (one exchange per "
                f"solver iteration) + all-reduced convergence test",
                "mean_divergence_iters": float(it[:, 0].mean()),
                "mean_pressure_iters": float(it[:, 1].mean()),
                "mean_contacts_per_particle": kbar,
                "grid_ms": float(it[:, 3].mean()),
                "solver_ms": float(it[:, 4].mean()),
                "tiles": tile_stats,
                "speculative_passes": int(w.counters.speculative_passes), "discarded_passes": int(w.counters.discarded_passes),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    # Tear down first and print last: RCCL writes a version banner through C stdio (flushed when it pleases, typically at
    # exit); the JSON line must be the last thing on stdout, after every rank has emptied its C buffers.
    try:
        if world > 1:
            dist.barrier()
        if comm is not None:
            del w
            comm.destroy()
        flush_c_stdio()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        flush_c_stdio()
    finally:
        if rank == 0:
            if world > 1:
                time.sleep(0.5)  # the other ranks are past their last flush by now (no process group left to sync on)
            sys.stdout.write(json.dumps(out) + "\n")
            sys.stdout.flush()


if __name__ == "__main__":
    try:
        main()
    except BaseException:  # noqa: BLE001
        # in a multi-rank run the other ranks are blocked in a neighbour exchange: die at once so that the launcher
        # tears the job down instead of waiting for a timeout
        import traceback

        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
