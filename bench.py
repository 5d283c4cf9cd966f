#!/usr/bin/env python
"""bench.py — particle-steps/s of the LiquidWorld::step hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W [--config {2,3,4}]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no launcher: bench.py starts torch.distributed.run itself, N ranks, one per GPU)
`--gpus` decides the number of ranks: it must equal WORLD_SIZE under a launcher, and the node must show N GPUs (ranks sharing
devices — a functional run only, flagged `ranks_sharing_gpus` — needs `--transport peer --share-devices`).

A "step" is one `LiquidWorld::step(dt = 1/200, g = -9.81 y)`.  Default workload = BASELINE config[1] ("3D DFSPH 1M particles,
single fluid, XSPH viscosity, 1xMI355X", concretised in SURVEY.md §8d config 2 (A)): a 100^3 lattice block (spacing 2r,
r = 0.025, h = 0.1, jitter +-0.1 r with LCG seed 42) resting in an open lattice tank (floor + 4 walls), rho0 = 1000,
XSPHViscosity(0.5, 0), DFSPH defaults.  `--config 3` (IISPH + Akinci2013(1.0, 10.0), same tank) and `--config 4` (two stacked
10^6-particle fluids, rho0 1000 / 500, XSPH, DFSPH) emit the same line for the other single-GPU configurations; they are
profiling aids, not the headline (N=1 default stays config 2).  State is resident in HBM when the timed region starts; the
timed region contains everything a step does (cell sort, neighbour lists, all solver passes, convergence read-backs) and
nothing else.  One JSON line is printed by rank 0.

The scene changes regime while it runs: free fall and impact (divergence solve converges in ~4 iterations), then — from
about step 24 — a compressed column whose divergence solve no longer converges within its 50-iteration cap (the oracle does
the same).  `value` is the whole-run mean the contract asks for and therefore depends on --steps; `regimes` and
`per_step_ms` / `iters` make that explicit: `first20` = the first 20 timed steps, `settled` = the timed steps whose
divergence solve hit the cap.
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


def slab_scene_geometry(side: int, world: int):
    """BASELINE config 5's geometry: `world` blocks of side^3 particles side by side along x in one long open tank.
    Returns (full boundary shell, [(first, last) cell plane of each slab]); block r is cut from block r-1 at the cell plane
    under its lower face."""
    d = 2.0 * R
    h = R * 2.0 * 2.0
    fmin = np.array([-side * R + R] * 3, dtype=np.float64)
    fmax = fmin + np.array([world * side - 1, side - 1, side - 1], dtype=np.float64) * d
    mins, maxs = fmin - d, fmax + d
    maxs[1] += max(side // 2, 4) * d
    shell = scenes.box_shell(mins, maxs, R, faces="xXyzZ")
    cuts = [int(np.floor((fmin[0] - 0.5 * d + r * side * d) / h)) for r in range(world + 1)]
    return shell, [(cuts[r], cuts[r + 1] - 1) for r in range(world)]


def slab_block(side: int, rank: int):
    """The fluid block of rank `rank` in the long tank (jitter seed 42 + rank)."""
    fluid = scenes.cube_fluid_positions(side, side, side, R)
    fluid[:, 0] += np.float32(rank * side * 2.0 * R)
    return scenes.jitter(fluid, 0.1 * R, seed=42 + rank)


def build_slab_scene(side: int, rank: int, world: int):
    """Weak scaling: `world` copies of the N=1 block side by side along x in one long tank; rank r uploads block r and the
    tank particles near its slab."""
    shell, slabs = slab_scene_geometry(side, world)
    mine = slab.boundary_subset(shell, R * 4.0, slabs[rank], rank, world)
    return slab_block(side, rank), shell[mine], slabs[rank], len(shell)


def make_world(fluid, shell, device: int):
    w = LiquidWorld(DFSPHSolver(), R, 2.0, device=device)
    f = Fluid(fluid, R, 1000.0)
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(shell))
    return w, f


CONFIGS = {
    2: dict(metric="particle-steps/sec (3D DFSPH)", solver="dfsph", kernel=(0, "k_pred_density", 52.0),
            what="single fluid, XSPH viscosity, lattice block in open tank"),
    3: dict(metric="particle-steps/sec (3D IISPH)", solver="iisph", kernel=(2, "k_iisph_next_pressure", 60.0),
            what="single fluid, IISPH + Akinci2013 surface tension (1.0, 10.0), lattice block in open tank"),
    4: dict(metric="particle-steps/sec (3D DFSPH, two-phase)", solver="dfsph", kernel=(0, "k_pred_density", 52.0),
            what="two stacked fluids (rho0 1000 below, 500 above), XSPH viscosity each, in one open tank"),
}


def build_config(config: int, side: int):
    """(list of (positions, density0), boundary positions) of a single-GPU configuration."""
    if config == 4:
        fluid, shell = scenes.tank(side, 2 * side, side, R)
        fluid = scenes.jitter(fluid, 0.1 * R, seed=42)
        mid = 0.5 * (float(fluid[:, 1].min()) + float(fluid[:, 1].max()))
        return [(np.ascontiguousarray(fluid[fluid[:, 1] < mid]), 1000.0), (np.ascontiguousarray(fluid[fluid[:, 1] >= mid]), 500.0)], shell
    fluid, shell = build_scene(side)
    return [(fluid, 1000.0)], shell


def make_config_world(config: int, fluids, shell, device: int):
    from salva_amd import Akinci2013SurfaceTension, IISPHSolver

    w = LiquidWorld(IISPHSolver() if CONFIGS[config]["solver"] == "iisph" else DFSPHSolver(), R, 2.0, device=device)
    handles = []
    for pos, rho0 in fluids:
        f = Fluid(pos, R, rho0)
        if config == 3:
            f.nonpressure_forces.append(Akinci2013SurfaceTension(1.0, 10.0))
        else:
            f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
        handles.append(w.add_fluid(f))
    w.add_boundary(Boundary(shell))
    return w, handles


def make_config_oracle(config: int, fluids, shell, threads: int):
    from oracle import oracle as O

    w = O.OracleWorld(R, 2.0, O.IISPH if CONFIGS[config]["solver"] == "iisph" else O.DFSPH, threads=threads, native=True)
    for pos, rho0 in fluids:
        fid = w.add_fluid(pos, rho0)
        if config == 3:
            w.add_akinci2013(fid, 1.0, 10.0)
        else:
            w.add_xsph(fid, 0.5, 0.0)
    w.add_boundary(shell)
    return w


def committed_traffic(kernel: str, config: int, side: int):
    """HBM bytes per launch of `kernel` from the committed PMC summary of THIS workload (profiles/r*_cfg<config>/ for the
    10^6-per-fluid configurations, profiles/r*_8m/ for side 200; newest round first), written by tools/summarize_pmc.py
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction of MI355X_MICROARCH.md.
    Counters cannot be read from inside this process, so this is the measured figure of a COMMITTED PROFILE, not of this run,
    and the JSON line says so (`traffic_source` = "committed_profile:<file>").  The profile records the hash of the kernel
    sources it was taken with (salva_amd.kernel_source_sha): when the sources have changed since, the figure is withheld
    (traffic = null, traffic_source = "STALE ...") and a warning goes to stderr, instead of going stale silently.
    Returns (bytes or None, source string or None)."""
    import glob

    from salva_amd import kernel_source_sha
    tag = {100: f"cfg{config}", 200: "8m" if config == 2 else None}.get(side)
    if tag is None:
        return None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}", "hbm_traffic.json")), reverse=True):
        try:
            j = json.load(open(f))
            k = j["kernels"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if k:
            rel = os.path.relpath(f, ROOT)
            sha = j.get("kernel_src_sha")
            if sha != kernel_source_sha():
                sys.stderr.write(f"bench.py: WARNING: {rel} was taken with other kernel sources (profile {sha}, tree {kernel_source_sha()}): "
                                 f"roofline.traffic withheld — re-run tools/profile_r03.sh and commit the summary\n")
                return None, f"STALE committed_profile:{rel} (kernel sources changed since it was taken)"
            return float(k["bytes"]), f"committed_profile:{rel}"
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


def cpu_baseline(config: int, side: int, steps: int, warmup: int, gpu_step_ms, budget_s: float = 25.0):
    """The CPU oracle (C++ restatement of salva's CPU path, kind = "port") on THE SAME scene at the same size: the same
    warm-up steps, then as many of the same timed steps as fit the time budget (at least two), on all usable host cores, in the
    -O3 -march=native timing build compiled on this machine (oracle/Makefile; same -ffp-contract=off, same results); then a
    few more steps on ONE thread — the reference's default build is serial (`parallel` is opt-in, build/salva3d/Cargo.toml:17-20).
    `gpu_same_steps` is the device's rate over exactly the steps the CPU timed, so the two can be compared like for like."""
    fluids, shell = build_config(config, side)
    n = sum(len(p) for p, _ in fluids)
    cores = effective_cores()
    w = make_config_oracle(config, fluids, shell, cores)
    for _ in range(warmup):
        w.step(DT, GRAVITY)
    t0 = time.perf_counter()
    done, iters = 0, []
    while done < steps and (done < 2 or time.perf_counter() - t0 < budget_s):
        st = w.step(DT, GRAVITY)
        iters.append((st.n_div_iters, st.n_press_iters))
        done += 1
    dt_all = time.perf_counter() - t0
    w.set_threads(1)
    t1 = time.perf_counter()
    done1 = 0
    while done1 < 1 or (time.perf_counter() - t1 < 0.4 * budget_s and done1 < 3):
        w.step(DT, GRAVITY)
        done1 += 1
    dt_one = time.perf_counter() - t1
    gpu_same = n * done / (sum(gpu_step_ms[:done]) * 1e-3) if len(gpu_step_ms) >= done and done else None
    return {"value": n * done / dt_all, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"the bench scene itself ({n} fluid particles + {len(shell)} boundary): {warmup} warm-up steps, then the first "
                      f"{done} of the {steps} timed steps in {dt_all:.1f} s, oracle/salva_oracle.cpp f32, -O3 -march=native, "
                      f"OpenMP {cores} threads; (divergence, pressure) iterations {iters[0]} .. {iters[-1]}",
            "steps": done, "gpu_same_steps": gpu_same,
            "single_thread": {"value": n * done1 / dt_one, "unit": "particle-steps/s", "cores": 1,
                              "sample": f"the {done1} step(s) after those, same build, 1 thread, {dt_one:.1f} s"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 5 warm-up + 50 timed steps
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE configuration (SURVEY.md §8d): 2 DFSPH + XSPH (headline, default), 3 IISPH + Akinci2013, 4 two-phase DFSPH")
    ap.add_argument("--side", type=int, default=100, help="particles per edge of the fluid block (100 -> 1M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-download-leg", action="store_true",
                    help="skip the second run of the same steps with positions + velocities read back after every step (`with_download`)")
    ap.add_argument("--no-big-leg", action="store_true",
                    help="skip the roofline of the dominant kernel at 8 x 10^6 particles (`roofline.at_8m`; default run of config 2 only)")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of host time for the multi-thread CPU leg")
    ap.add_argument("--transport", choices=("rccl", "peer"), default="rccl",
                    help="slab exchange of a multi-GPU run: RCCL send/recv + all-reduce (default, what BASELINE.json names), or the "
                         "xGMI peer-direct transport (flagged stores into hipIpc-mapped windows, salva_amd/csrc/comm_peer.hip)")
    ap.add_argument("--force-slabs", action="store_true",
                    help="take the decomposed (RCCL transport) code path even with one rank; a self-test aid, not a bench mode")
    ap.add_argument("--share-devices", action="store_true",
                    help="allow more ranks than GPUs (ranks share the devices round-robin; needs --transport peer): a functional run of "
                         "the multi-process path on a single-GPU box, never a scaling measurement")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to measure")
    ndev = torch.cuda.device_count()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > ndev and not (args.share_devices and args.transport == "peer"):
        raise SystemExit(f"--gpus {args.gpus} but this node shows {ndev} GPU(s): one rank per GPU over RCCL needs {args.gpus} devices "
                         f"(a functional run with ranks sharing devices: --transport peer --share-devices)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1), so that
        # the flag alone decides how many ranks step — the line's n_gpus is then the ranks that actually ran
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): the two must agree (n_gpus in the line = ranks that stepped)")
    # More ranks than GPUs (a self-test aid on the single-GPU boxes, not a scaling measurement): the ranks share the devices
    # round-robin.  RCCL refuses two ranks on one device, so such a run needs --transport peer, and torch.distributed (used here
    # for the barrier and the max-over-ranks of the elapsed time only) runs over gloo.
    shared = world > ndev
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            if args.transport != "peer":
                raise SystemExit(f"{world} ranks on {ndev} GPU(s): RCCL needs one GPU per rank; --transport peer accepts ranks sharing a device")
            dist.init_process_group("gloo")
        else:
            slab.single_node_rccl_env()  # one node by contract: keep RCCL's bootstrap off the (absent) network
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    comm = None
    decomposed = world > 1 or args.force_slabs
    cfg = CONFIGS[args.config]
    if decomposed and args.config != 2:
        raise SystemExit("the decomposed run is BASELINE config 5 = config 2 per GPU: use --config 2")
    if not decomposed:
        fluids, shell = build_config(args.config, args.side)
        nshell_total = len(shell)
        w, handles = make_config_world(args.config, fluids, shell, local_rank)
        n = sum(len(p) for p, _ in fluids)
    else:
        # slab decomposition along x: RCCL point-to-point with the two neighbours + one tiny all-reduce per convergence test
        fluid, shell, my_slab, nshell_total = build_slab_scene(args.side, rank, world)
        if args.transport == "peer":
            def gather(handle):
                out = [None] * world
                if world > 1:
                    dist.all_gather_object(out, handle)
                else:
                    out[0] = handle
                return out
            comm = slab.Comm.peer(rank, world, local_rank, gather)
        else:
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(slab.Comm.unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(idt, 0)
            comm = slab.Comm.rccl(rank, world, bytes(idt.cpu().numpy().tobytes()), local_rank)
        w, f = make_world(fluid, shell, local_rank)
        handles = [f]
        w.set_domain(comm, my_slab[0], my_slab[1], rank * len(fluid))
        n = len(fluid)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    iters, step_ms = [], []
    # The stage timers (Counters, off by default as in the reference) cost ~20 us of host time per step: they run during the
    # warm-up steps only, which is where the grid / solver split reported in `config` comes from.
    w.counters.enable()
    warm = []
    for _ in range(args.warmup):
        stw = w.step(DT, GRAVITY)
        warm.append((stw.grid_ms, stw.solver_ms))
    # decomposed runs: what the exchanges cost inside the last warm-up step (HIP events around every ghost refresh and every
    # all-reduced convergence test, the wait for the neighbour included) — VERDICT r03, item 2
    exchange = None
    if decomposed and warm:
        t = w.dist_timing()
        exchange = {"what": "last warm-up step, this rank: ghost refresh = gather + exchange with the neighbours + scatter; test = error sum + "
                            "all-reduce + decision; HIP events on the world's stream, waiting for the neighbour included",
                    "refreshes_per_step": t["refreshes"], "us_per_refresh": (1e3 * t["refresh_ms"] / t["refreshes"]) if t["refreshes"] else None,
                    "tests_per_step": t["tests"], "us_per_test": (1e3 * t["test_ms"] / t["tests"]) if t["tests"] else None}
    w.counters.disable()
    barrier()
    t0 = time.perf_counter()
    tp = t0
    for _ in range(args.steps):
        st = w.step(DT, GRAVITY)  # returns when the step has completed on the device (its last read-back)
        tn = time.perf_counter()
        step_ms.append((tn - tp) * 1e3)
        tp = tn
        iters.append((st.n_divergence_iters, st.n_pressure_iters, st.ncontacts, st.grid_ms, st.solver_ms))
    tile_stats = {"max_halo_fluid": int(st.reserved[0]), "max_halo_boundary": int(st.reserved[1]), "tile_threads": int(st.reserved[2]),
                  "ghost_particles": int(st.reserved[4])}
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant neighbour-sum kernel, timed live with HIP events on the world's own stream
    it = np.asarray(iters, dtype=np.float64)
    kid, kname, sbytes = cfg["kernel"]
    kernel_us = w.time_kernel(kid, 50)
    if not decomposed:
        kbar = float(sum(w.contact_counts(h).sum() + w.contact_counts(h, True).sum() for h in handles)) / n
    else:  # host-order fields do not exist in a decomposed run: list entries per local particle, from the step report
        kbar = float(st.reserved[3])

    def kernel_roofline(name, us, s_bytes):
        """One kernel against the HBM roofline, both ways: `frac` / `frac_algorithmic` = SURVEY.md §8d's byte
        model N (4K + S) over the measured duration; `frac_measured_traffic` = the HBM bytes the kernel really moved (PMC
        FETCH_SIZE x 2 + WRITE_SIZE of the committed profile of this workload; list entries are 16-bit slots, so this is the
        smaller figure wherever the lists dominate) over the same duration."""
        algo = n * (4.0 * kbar + s_bytes)
        traffic, src = committed_traffic(name, args.config, args.side) if not decomposed else (None, None)
        ach = algo / (us * 1e-6) / 1e9
        return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "frac_algorithmic": ach / HBM_PEAK_GBS,
                "frac_measured_traffic": None if traffic is None else traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": src, "kernel_us": us, "algorithmic_bytes": algo, "mean_contacts": kbar}

    roofline = kernel_roofline(kname, kernel_us, sbytes)
    # the other kernels a step lives in (DFSPH: the divergence solve's two passes — 86 % of a settled step — and the list build)
    others = {}
    if cfg["solver"] == "dfsph":
        for oid, oname, os_ in ((1, "k_divergence", 48.0), (6, "k_divergence_apply", 44.0), (4, "k_nbr_tile", 12.0)):
            if oid == 4 and decomposed:
                continue
            try:
                others[oname] = kernel_roofline(oname, w.time_kernel(oid, 30), os_)
            except Exception as e:  # noqa: BLE001 - a timing aid must not cost the line
                others[oname] = {"error": str(e)}
    roofline["other_kernels"] = others
    spec_passes, disc_passes = int(w.counters.speculative_passes), int(w.counters.discarded_passes)
    div_cap = float(getattr(w.solver, "max_divergence_iter", 50))

    # ---- what a user pays who looks at the particles after every step (the reference's API leaves positions / velocities on the
    # host after `step`; testbed_plugin.rs:361-367 reads them each frame): the SAME warm-up + timed steps on a fresh world, with an
    # asynchronous read-back of positions and velocities into pinned host arrays after every step, one step late
    # (salva_hip_get_fluid_async / _wait_download).  Reported beside `value`, never as `value` (state resident: the contract).
    with_download = None
    if not decomposed and not args.no_download_leg and rank == 0:
        try:
            w = None  # (release the first world's device memory)
            w2, handles2 = make_config_world(args.config, fluids, shell, local_rank)
            for _ in range(args.warmup):  # (the warm-up steps read back too: the first read-back allocates the pinned arrays, 6-19 ms once)
                w2.step(DT, GRAVITY)
                w2.wait_download()
                for hd in handles2:
                    w2.download_async(hd)
            w2.wait_download()
            torch.cuda.synchronize()
            td = time.perf_counter()
            for _ in range(args.steps):
                w2.step(DT, GRAVITY)
                w2.wait_download()
                for hd in handles2:
                    w2.download_async(hd)
            w2.wait_download()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - td
            with_download = {"value": n * args.steps / el2, "unit": "particle-steps/s", "ms_per_step": el2 / args.steps * 1e3,
                             "what": "the same warm-up and timed steps on a fresh world, positions + velocities of every fluid read back after "
                                     "every step: asynchronously, into pinned host arrays, overlapping the next step (one step late)",
                             "bytes_per_step": int(24 * n)}
            w = w2
        except Exception as e:  # noqa: BLE001 - a reporting leg must not cost the line
            with_download = {"error": str(e)}

    # ---- the same kernel where the state does NOT fit the 256 MiB Infinity Cache: 200^3 = 8 x 10^6 particles (config 2 only, a few
    # steps of free fall; about 5 s with the scene build).  At 10^6 the launch is ~3 rounds of resident tiles deep; at 8 x 10^6, 23.
    at_8m = None
    if not decomposed and args.config == 2 and args.side == 100 and not args.no_big_leg and rank == 0:
        try:
            w = None
            fl8, sh8 = build_config(2, 200)
            w8, h8 = make_config_world(2, fl8, sh8, local_rank)
            for _ in range(4):
                w8.step(DT, GRAVITY)
            n8 = sum(len(p_) for p_, _ in fl8)
            k8 = float(sum(w8.contact_counts(h).sum() + w8.contact_counts(h, True).sum() for h in h8)) / n8
            us8 = w8.time_kernel(kid, 30)
            algo8 = n8 * (4.0 * k8 + sbytes)
            at_8m = {"kernel": kname, "particles": n8, "kernel_us": us8, "mean_contacts": k8, "algorithmic_bytes": algo8,
                     "achieved": algo8 / (us8 * 1e-6) / 1e9, "frac": algo8 / (us8 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "what": "the same kernel on a 200^3 block after 4 steps (free fall): state beyond the Infinity Cache"}
            w8 = None
        except Exception as e:  # noqa: BLE001
            at_8m = {"error": str(e)}
    roofline["at_8m"] = at_8m

    if rank == 0:
        # ---- the regimes the run went through (see the docstring): rates over the first 20 timed steps and over the timed
        # steps whose divergence solve ran into its iteration cap
        sm = np.asarray(step_ms)
        cap = div_cap
        first = sm[:20]
        settled = sm[it[:, 0] >= cap] if cfg["solver"] == "dfsph" else sm[:0]
        regimes = {
            "first20": {"steps": int(len(first)), "ms_per_step": float(first.mean()), "value": float(n * world / (first.mean() * 1e-3)),
                        "mean_divergence_iters": float(it[:20, 0].mean())},
            "settled": None if len(settled) == 0 else
            {"steps": int(len(settled)), "ms_per_step": float(settled.mean()), "value": float(n * world / (settled.mean() * 1e-3)),
             "what": f"timed steps whose divergence solve used all {int(cap)} iterations"},
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args.config, args.side, args.steps, args.warmup, step_ms, args.cpu_budget)
        value = n * world * args.steps / elapsed
        out = {
            "metric": cfg["metric"],
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": min(world, ndev),
            "ranks": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {args.config if not decomposed else 5}: 3D {cfg['solver'].upper()} {n * world} fluid particles "
                            f"(+{nshell_total} boundary), {cfg['what']}, r=0.025 h=0.1 dt=1/200",
                "particles_per_gpu": n * world // min(world, ndev),
                "ranks_sharing_gpus": bool(shared),  # true: a functional run of the multi-process path, not a scaling measurement
                "parallelism": "single domain" if not decomposed else
                (f"{world} x-slabs, one per GPU: RCCL send/recv of two ghost planes per face with the 2 neighbours (one exchange per "
                 f"solver iteration) + all-reduced convergence test" if args.transport == "rccl" else
                 f"{world} x-slabs, one per GPU: xGMI peer-direct exchange (flagged stores into the neighbours' hipIpc-mapped windows) of two "
                 f"ghost planes per face, one exchange per solver iteration, convergence sums through every rank's window"),
                "mean_divergence_iters": float(it[:, 0].mean()),
                "mean_pressure_iters": float(it[:, 1].mean()),
                "mean_contacts_per_particle": kbar,
                # (the last warm-up step: the first ones allocate)
                "warmup_grid_ms": float(warm[-1][0]) if warm else None,
                "warmup_solver_ms": float(warm[-1][1]) if warm else None,
                "tiles": tile_stats,
                "exchange": exchange,
                "speculative_passes": spec_passes, "discarded_passes": disc_passes,
            },
            "regimes": regimes,
            "with_download": with_download,
            "per_step_ms": [round(float(x), 4) for x in step_ms],
            "iters": [[int(a), int(b)] for a, b in it[:, :2]],
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    # Tear down first and print last: RCCL writes a version banner through C stdio (flushed when it pleases, typically at
    # exit); the JSON line must be the last thing on stdout, after every rank has emptied its C buffers.
    try:
        if world > 1:
            dist.barrier()
        if comm is not None:
            w = None
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
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001
        # in a multi-rank run the other ranks are blocked in a neighbour exchange: die at once so that the launcher
        # tears the job down instead of waiting for a timeout
        import traceback

        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
