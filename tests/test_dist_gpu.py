"""Multi-GPU slab decomposition on the GPU: N in-process ranks (one host thread and one world each, exchanging through
the loopback transport — the same World code path RCCL drives) must step to the particle states of one world that holds
the whole domain.  The only differences allowed are floating-point summation order (tile layouts differ per rank)."""
import threading

import numpy as np
import pytest

from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, dist, scenes

pytestmark = pytest.mark.gpu

R, SF = 0.025, 2.0
H = R * SF * 2
DT = 1.0 / 200.0
G = (0.0, -9.81, 0.0)


def make_scene(nx=40, ny=10, nz=10, seed=5):
    pos, bpos = scenes.tank(nx, ny, nz, R, wall_cells=4)
    pos = scenes.jitter(pos, 0.1 * R, seed)
    vel = scenes.random_velocities(len(pos), 0.5, seed + 1)
    vel[:, 0] += 2.0  # the whole block drifts towards +x: a full cell plane changes owner during the test
    return pos.astype(np.float32), vel.astype(np.float32), bpos.astype(np.float32)


SOLVER = {"kind": "dfsph"}  # switched by the IISPH test
STEP = {"dt": DT, "cfl": 0}  # switched by the CFL sub-stepping test
FORCES = {"make": lambda: [XSPHViscosity(0.5, 0.0)]}  # switched by the surface-tension test


def solver():
    from salva_amd import IISPHSolver

    return IISPHSolver() if SOLVER["kind"] == "iisph" else DFSPHSolver()


def run_single(pos, vel, bpos, nsteps, two_fluids):
    w = LiquidWorld(solver(), R, SF)
    fls = []
    for part in split_fluids(pos, two_fluids):
        f = Fluid(pos[part], R, 1000.0 if len(fls) == 0 else 800.0)
        f.velocities = vel[part]
        f.nonpressure_forces.extend(FORCES["make"]())
        fls.append((w.add_fluid(f), part))
    w.add_boundary(Boundary(bpos))
    if STEP["cfl"]:
        w.set_cfl_substepping(STEP["cfl"])
    stats = []
    for _ in range(nsteps):
        stats.append(w.step(STEP["dt"], G))
        stats[-1].substeps = w.substeps()
    out_p = np.zeros_like(pos)
    out_v = np.zeros_like(vel)
    for f, part in fls:
        out_p[part] = f.positions
        out_v[part] = f.velocities
    return out_p, out_v, stats


def split_fluids(pos, two_fluids):
    idx = np.arange(len(pos))
    if not two_fluids:
        return [idx]
    upper = pos[:, 1] > np.median(pos[:, 1])
    return [idx[~upper], idx[upper]]


def run_slabs(pos, vel, bpos, nsteps, nranks, two_fluids):
    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, nranks)
    owner = dist.owner_of(cx, slabs)
    comms = dist.Comm.loopback(nranks)
    parts = split_fluids(pos, two_fluids)
    results, errors, stats = [None] * nranks, [None] * nranks, [None] * nranks
    # global id = position in the concatenation (rank major, fluid, upload order)
    gid_to_global, offsets = [], []
    for r in range(nranks):
        offsets.append(len(gid_to_global))
        for part in parts:
            gid_to_global.extend(part[owner[part] == r].tolist())
    gid_to_global = np.array(gid_to_global)

    def rank_main(r):
        try:
            w = LiquidWorld(solver(), R, SF)
            for k, part in enumerate(parts):
                mine = part[owner[part] == r]
                f = Fluid(pos[mine], R, 1000.0 if k == 0 else 800.0)
                f.velocities = vel[mine]
                f.nonpressure_forces.extend(FORCES["make"]())
                w.add_fluid(f)
            w.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[r], r, nranks)]))
            w.set_domain(comms[r], slabs[r][0], slabs[r][1], offsets[r])
            if STEP["cfl"]:
                w.set_cfl_substepping(STEP["cfl"])
            stats[r] = []
            for _ in range(nsteps):
                stats[r].append(w.step(STEP["dt"], G))
                stats[r][-1].substeps = w.substeps()
            results[r] = w.owned()
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for e in errors:
        if e is not None:
            raise e
    assert not any(t.is_alive() for t in threads), "a rank hung"
    out_p = np.full_like(pos, np.nan)
    out_v = np.full_like(vel, np.nan)
    seen = np.zeros(len(pos), int)
    counts = []
    for r in range(nranks):
        gid, p, v, _slot = results[r]
        g = gid_to_global[gid]
        out_p[g] = p
        out_v[g] = v
        np.add.at(seen, g, 1)
        counts.append(len(gid))
    for c in comms:
        c.destroy()
    return out_p, out_v, stats, seen, counts, slabs


@pytest.mark.parametrize("nranks,two_fluids", [(2, False), (3, False), (2, True)])
def test_slabs_match_single_domain(hip_lib, nranks, two_fluids):
    pos, vel, bpos = make_scene()
    nsteps = 12
    ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, two_fluids)
    got_p, got_v, stats, seen, counts, slabs = run_slabs(pos, vel, bpos, nsteps, nranks, two_fluids)
    # every particle is owned by exactly one rank at the end, and some changed owner on the way
    assert (seen == 1).all(), f"{(seen != 1).sum()} particles lost or duplicated"
    final_owner = dist.owner_of(dist.cell_x(got_p, H), slabs)
    first_owner = dist.owner_of(dist.cell_x(pos, H), slabs)
    assert (final_owner != first_owner).sum() > 0, "the scene was meant to exercise migration"
    # the convergence test is global: all ranks take the same number of iterations, the same as the single domain
    for k in range(nsteps):
        it = {(s[k].n_divergence_iters, s[k].n_pressure_iters) for s in stats}
        assert len(it) == 1, f"step {k}: ranks disagree on iteration counts {it}"
        assert sum(int(s[k].nparticles) for s in stats) == len(pos)
    n_same = sum((stats[0][k].n_divergence_iters, stats[0][k].n_pressure_iters)
                 == (ref_stats[k].n_divergence_iters, ref_stats[k].n_pressure_iters) for k in range(nsteps))
    assert n_same >= nsteps - 2, "iteration counts drifted from the single-domain run"
    # total contacts: ff contacts are counted by the owner of the first particle; ghosts must not add any
    # states agree up to summation order
    dp = np.abs(got_p - ref_p).max()
    dv = np.abs(got_v - ref_v).max()
    assert dp < 2e-4 * H, f"positions differ by {dp / H:.2e} h"
    assert dv < 5e-3, f"velocities differ by {dv:.2e} m/s"


def test_cfl_substeps_in_a_decomposed_run(hip_lib):
    """Opt-in CFL sub-stepping (row f4) across slabs: max_i |v_i + a_i t| is a maximum over ALL particles, so the ranks all-reduce it
    (each rank's own maximum in its own slot of a summed row: exact) and every rank takes the substeps of the undivided world."""
    pos, vel, bpos = make_scene()
    nsteps = 5
    STEP.update(dt=1.0 / 60.0, cfl=2)
    try:
        ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, False)
        got_p, got_v, stats, seen, counts, slabs = run_slabs(pos, vel, bpos, nsteps, 2, False)
    finally:
        STEP.update(dt=DT, cfl=0)
    assert (seen == 1).all()
    assert max(len(s.substeps) for s in ref_stats) >= 3, "the scene was meant to sub-step"
    for k in range(nsteps):
        subs = [tuple(s[k].substeps) for s in stats]
        assert subs[0] == subs[1], f"step {k}: the ranks took different substeps {subs}"
        assert len(subs[0]) == len(ref_stats[k].substeps) and np.allclose(subs[0], ref_stats[k].substeps, rtol=1e-4), (k, subs[0], ref_stats[k].substeps)
    dp = np.abs(got_p - ref_p).max()
    dv = np.abs(got_v - ref_v).max()
    assert dp < 5e-4 * H, f"positions differ by {dp / H:.2e} h"
    assert dv < 1e-2, f"velocities differ by {dv:.2e} m/s"


class _HostField:
    """examples3d/custom_forces3.rs:67-90 (a pull towards a point) as a user NonPressureForce: per particle, no contacts."""

    def __init__(self, origin):
        from salva_amd import NonPressureForce  # noqa: F401 - (duck-typed: _desc comes from the base class below)
        self.origin = np.float32(origin)

    def solve(self, timestep, kernel_radius, ff, fb, fluid, boundaries, densities):
        d = self.origin - fluid.positions
        dist_ = np.sqrt((d * d).sum(1, dtype=np.float32))
        ok = dist_ > 0.1
        fluid.accelerations[ok] += d[ok] / dist_[ok, None] / dist_[ok, None]


class _HostXsph:
    """xsph_viscosity.rs:47-94 (fluid term) over the raw CSR arrays of the mirror's ParticlesContacts, vectorised: what a user
    plugin that needs the neighbours does.  In a decomposed run the arrays describe the rank's part of the fluid."""

    def __init__(self, coeff):
        self.c = np.float32(coeff)

    def solve(self, timestep, kernel_radius, ff, fb, fluid, boundaries, densities):
        n = fluid.num_particles()
        cnt = (ff.offsets[1:] - ff.offsets[:-1]).astype(np.int64)
        i = np.repeat(np.arange(n), cnt)
        j = ff.j.astype(np.int64)
        same = ff.j_model == ff.i_model
        i, j = i[same], j[same]
        h = np.float32(kernel_radius)
        d = fluid.positions[i] - fluid.positions[j]
        q = np.sqrt((d * d).sum(1, dtype=np.float32)) / h
        w = np.where(q <= 0.5, 1 + 6 * (q ** 3 - q ** 2), 2 * (1 - q) ** 3).astype(np.float32) * np.float32(8 / np.pi) / (h * h * h)
        w[q > 1] = 0
        m = (fluid.volumes * np.float32(fluid.density0))[j]
        term = (fluid.velocities[j] - fluid.velocities[i]) * (self.c * w * m / densities[j])[:, None]
        acc = np.zeros((n, 3), np.float32)
        np.add.at(acc, i, term)
        fluid.accelerations += acc * np.float32(timestep.inv_dt())


def _custom_forces():
    from salva_amd import NonPressureForce

    class Field(_HostField, NonPressureForce):
        pass

    class HXsph(_HostXsph, NonPressureForce):
        pass

    return [XSPHViscosity(0.3, 0.0), Field([1.2, 0.4, 0.1]), HXsph(0.4)]


def test_user_forces_run_on_every_rank_of_a_decomposed_world(hip_lib):
    """VERDICT r03, missing 4: a user-defined `NonPressureForce` (nonpressure_force.rs:10-30; custom_forces3.rs:67-90) in a
    decomposed run.  The callback runs on every rank at its place in the force list and sees the rank's part of the fluid — owned
    particles and ghosts, contacts as local indices (salva_hip_get_local / _get_local_contacts / _force_add_local_accelerations).
    Two slabs with a field force and a contact-based host XSPH step to the undivided world's states running the same two forces."""
    pos, vel, bpos = make_scene(nx=28, ny=8, nz=8)
    nsteps = 6
    FORCES["make"] = _custom_forces
    try:
        ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, False)
        got_p, got_v, stats, seen, counts, slabs = run_slabs(pos, vel, bpos, nsteps, 2, False)
    finally:
        FORCES["make"] = lambda: [XSPHViscosity(0.5, 0.0)]
    assert (seen == 1).all()
    for k in range(nsteps):
        it = {(s[k].n_divergence_iters, s[k].n_pressure_iters) for s in stats}
        assert len(it) == 1, f"step {k}: ranks disagree on iteration counts {it}"
    dp = np.abs(got_p - ref_p).max()
    dv = np.abs(got_v - ref_v).max()
    assert dp < 2e-4 * H, f"positions differ by {dp / H:.2e} h"
    assert dv < 5e-3, f"velocities differ by {dv:.2e} m/s"
    # the forces did something: without them the block would not be pulled towards the field's origin
    plain_p, _, _ = run_single(pos, vel, bpos, nsteps, False)
    assert np.abs(plain_p - ref_p).max() > 1e-3 * H


def test_local_view_and_local_contacts_describe_the_same_lists_as_the_host_order_export(hip_lib):
    """The per-rank contact export (salva_hip_get_local_contacts) on a single-domain world, where the host-order export exists to
    compare with: the same contact SETS once local indices are mapped to host indices through the ids of the local view."""
    pos, vel, bpos = make_scene(nx=12, ny=8, nz=8)
    w = LiquidWorld(DFSPHSolver(), R, SF)
    f = Fluid(pos, R, 1000.0)
    f.velocities = vel
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(bpos))
    for _ in range(3):
        w.step(DT, G)
    lv = w.local_view()
    assert len(lv["ids"]) == len(pos) and not lv["is_ghost"].any() and sorted(lv["ids"].tolist()) == list(range(len(pos)))
    assert np.array_equal(lv["positions"], np.asarray(f.positions)[lv["ids"]]) and np.array_equal(lv["velocities"], np.asarray(f.velocities)[lv["ids"]])
    assert np.allclose(lv["volumes"], f.volumes[0]) and (lv["densities"] > 0).all()
    for boundary in (False, True):
        off_l, jm_l, j_l = w.local_contacts(boundary)
        off_h, jm_h, j_h = w.fluid_contacts(f, boundary)
        ids = lv["ids"].astype(np.int64)
        for li in range(0, len(ids), 37):
            hi = ids[li]
            loc = j_l[int(off_l[li]):int(off_l[li + 1])].astype(np.int64)
            loc = ids[loc] if not boundary else loc
            host = j_h[int(off_h[hi]):int(off_h[hi + 1])].astype(np.int64)
            assert sorted(loc.tolist()) == sorted(host.tolist()), f"particle {hi}: local and host-order lists differ"
        assert int(off_l[-1]) == int(off_h[-1])


def test_slabs_match_single_domain_iisph(hip_lib):
    """The same comparison with the IISPH solver (its d_ii, sum d_ij p_j and pressure fields are refreshed per pass)."""
    SOLVER["kind"] = "iisph"
    try:
        pos, vel, bpos = make_scene()
        nsteps = 10
        ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, False)
        got_p, got_v, stats, seen, counts, slabs = run_slabs(pos, vel, bpos, nsteps, 2, False)
    finally:
        SOLVER["kind"] = "dfsph"
    assert (seen == 1).all()
    for k in range(nsteps):
        assert len({s[k].n_pressure_iters for s in stats}) == 1
    n_same = sum(stats[0][k].n_pressure_iters == ref_stats[k].n_pressure_iters for k in range(nsteps))
    assert n_same >= nsteps - 2
    assert np.abs(got_p - ref_p).max() < 2e-4 * H and np.abs(got_v - ref_v).max() < 5e-3


def test_slabs_with_surface_tensions(hip_lib):
    """He2014 (three dependent neighbour passes: the colours of the outer ghost plane are refreshed between them),
    WCSPH cohesion and Akinci2013 (normals of the inner ghost plane computed locally) across a slab boundary."""
    from salva_amd import Akinci2013SurfaceTension, He2014SurfaceTension, WCSPHSurfaceTension

    FORCES["make"] = lambda: [He2014SurfaceTension(1.0, 0.5), WCSPHSurfaceTension(0.2, 0.0), Akinci2013SurfaceTension(0.5, 2.0)]
    try:
        pos, vel, bpos = make_scene()
        nsteps = 8
        ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, True)
        got_p, got_v, stats, seen, counts, slabs = run_slabs(pos, vel, bpos, nsteps, 2, True)
    finally:
        FORCES["make"] = lambda: [XSPHViscosity(0.5, 0.0)]
    assert (seen == 1).all()
    assert np.isfinite(got_p).all()
    assert np.abs(got_p - ref_p).max() < 2e-4 * H and np.abs(got_v - ref_v).max() < 5e-3


def test_slabs_with_iterative_viscosity(hip_lib):
    """DFSPHViscosity adds a third globally-converged solve loop and three more ghost refreshes per iteration (v + a dt,
    u); two slabs must follow the single domain, iteration counts included.  (Barely perturbed lattice: the force
    diverges on rougher ones in the reference itself, tests/golden_scenes.py.)"""
    from salva_amd import DFSPHViscosity

    pos = scenes.jitter(scenes.cube_fluid_positions(24, 8, 8, R), 0.02 * R, 3).astype(np.float32)  # free block, no walls
    bpos = np.zeros((0, 3), np.float32)
    G = (0.0, 0.0, 0.0)
    vel = scenes.random_velocities(len(pos), 0.01, 4).astype(np.float32)
    vel[:, 0] += np.float32(2.0) * pos[:, 1]
    nsteps = 4

    def build(positions, velocities, boundary):
        w = LiquidWorld(solver(), R, SF)
        f = Fluid(positions, R, 1000.0)
        f.velocities = velocities
        f.nonpressure_forces.append(DFSPHViscosity(0.6))
        w.add_fluid(f)
        if len(boundary):
            w.add_boundary(Boundary(boundary))
        return w, f

    w, f = build(pos, vel, bpos)
    ref_iters = []
    for _ in range(nsteps):
        w.step(DT, G)
        ref_iters.append(f.nonpressure_forces[0].num_iterations)
    ref_p, ref_v = f.positions.copy(), f.velocities.copy()

    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, 2)
    owner = dist.owner_of(cx, slabs)
    comms = dist.Comm.loopback(2)
    order = np.concatenate([np.nonzero(owner == r)[0] for r in range(2)])
    offs = [0, int((owner == 0).sum())]
    res, errs, iters = [None, None], [None, None], [None, None]

    def rank_main(r):
        try:
            mine = np.nonzero(owner == r)[0]
            wr, fr = build(pos[mine], vel[mine], bpos[dist.boundary_subset(bpos, H, slabs[r], r, 2)])
            wr.set_domain(comms[r], slabs[r][0], slabs[r][1], offs[r])
            it = []
            for _ in range(nsteps):
                wr.step(DT, G)
                it.append(fr.nonpressure_forces[0].num_iterations)
            iters[r] = it
            res[r] = wr.owned()
        except BaseException as e:  # noqa: BLE001
            errs[r] = e

    ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a rank hung"
    for e in errs:
        if e is not None:
            raise e
    assert iters[0] == iters[1] == ref_iters
    got_p = np.full_like(pos, np.nan)
    got_v = np.full_like(vel, np.nan)
    for r in range(2):
        gid, p, v, _ = res[r]
        got_p[order[gid]] = p
        got_v[order[gid]] = v
    assert np.isfinite(got_p).all()
    assert np.abs(got_p - ref_p).max() < 2e-3 * H and np.abs(got_v - ref_v).max() < 2e-2
    for c in comms:
        c.destroy()


def test_single_rank_domain_is_the_plain_world(hip_lib):
    """A 1-rank communicator: no neighbours, no ghosts — must equal the plain world bit for bit in iteration counts and
    to rounding in state (the tile origin is the same, so this is in practice exact)."""
    pos, vel, bpos = make_scene(16, 8, 8)
    ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, 5, False)
    got_p, got_v, stats, seen, counts, _ = run_slabs(pos, vel, bpos, 5, 1, False)
    assert (seen == 1).all()
    assert [(s.n_divergence_iters, s.n_pressure_iters) for s in stats[0]] == \
           [(s.n_divergence_iters, s.n_pressure_iters) for s in ref_stats]
    np.testing.assert_allclose(got_p, ref_p, atol=1e-6)
    np.testing.assert_allclose(got_v, ref_v, atol=1e-5)


def test_rccl_transport_single_rank(hip_lib):
    """The RCCL transport itself (communicator creation, all-reduce, empty neighbour exchange) with a 1-rank
    communicator — all a 1-GPU box can run; the >1-rank protocol is the loopback-tested one above."""
    pos, vel, bpos = make_scene(16, 8, 8)
    ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, 4, False)
    comm = dist.Comm.rccl(0, 1, dist.Comm.unique_id(), 0)
    w = LiquidWorld(solver(), R, SF)
    f = Fluid(pos, R, 1000.0)
    f.velocities = vel
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    w.add_fluid(f)
    w.add_boundary(Boundary(bpos))
    cx = dist.cell_x(pos, H)
    w.set_domain(comm, int(cx.min()), int(cx.max()), 0)
    stats = [w.step(DT, G) for _ in range(4)]
    gid, p, v, _ = w.owned()
    assert (gid == np.arange(len(pos))).all()
    assert [(s.n_divergence_iters, s.n_pressure_iters) for s in stats] == \
           [(s.n_divergence_iters, s.n_pressure_iters) for s in ref_stats]
    np.testing.assert_allclose(p, ref_p, atol=1e-6)
    np.testing.assert_allclose(v, ref_v, atol=1e-5)
    with pytest.raises(Exception, match="multi-GPU"):
        _ = f.positions  # host-order access is replaced by owned()
    del w
    comm.destroy()


def test_boundary_forces_sum_to_the_single_domain_ones(hip_lib):
    """`boundary.forces` (Boundary::apply_force, boundary.rs:62-67) in a decomposed run: the tank's particles near the cut exist
    on both ranks, and ghosts run through every kernel — each reaction force must nevertheless be counted exactly once (by
    the rank that owns the fluid particle), so that the per-particle sums over the ranks equal the single-domain run's."""
    pos, vel, bpos = make_scene()
    nsteps, nranks = 6, 2

    w = LiquidWorld(solver(), R, SF)
    f = Fluid(pos, R, 1000.0)
    f.velocities = vel
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.3))
    w.add_fluid(f)
    b = w.add_boundary(Boundary(bpos, wants_forces=True))
    for _ in range(nsteps):
        w.step(DT, G)
    ref = b.forces.astype(np.float64)

    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, nranks)
    owner = dist.owner_of(cx, slabs)
    comms = dist.Comm.loopback(nranks)
    got = np.zeros_like(ref)
    errors = [None] * nranks
    lock = threading.Lock()
    offsets = np.concatenate([[0], np.cumsum([(owner == r).sum() for r in range(nranks)])])

    def rank_main(r):
        try:
            wr = LiquidWorld(solver(), R, SF)
            mine = np.nonzero(owner == r)[0]
            fr = Fluid(pos[mine], R, 1000.0)
            fr.velocities = vel[mine]
            fr.nonpressure_forces.append(XSPHViscosity(0.5, 0.3))
            wr.add_fluid(fr)
            sub = dist.boundary_subset(bpos, H, slabs[r], r, nranks)
            br = wr.add_boundary(Boundary(bpos[sub], wants_forces=True))
            wr.set_domain(comms[r], slabs[r][0], slabs[r][1], int(offsets[r]))
            for _ in range(nsteps):
                wr.step(DT, G)
            with lock:
                got[sub] += br.forces.astype(np.float64)
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for e in errors:  # a rank that raised leaves its peers waiting in the next exchange: report the cause first
        if e is not None:
            raise e
    assert not any(t.is_alive() for t in threads), "a rank hung"
    for c in comms:
        c.destroy()
    scale = np.abs(ref).max()
    assert scale > 0
    # the particles both ranks hold are the interesting ones
    shared = np.intersect1d(dist.boundary_subset(bpos, H, slabs[0], 0, nranks), dist.boundary_subset(bpos, H, slabs[1], 1, nranks))
    assert len(shared) > 50 and np.abs(ref[shared]).max() > 0.01 * scale
    assert np.abs(got - ref).max() < 5e-3 * scale, f"summed boundary forces differ by {np.abs(got - ref).max() / scale:.2e} of the largest force"


@pytest.mark.parametrize("host_shape", [False, True])
def test_dynamic_contact_sampling_in_a_decomposed_world(hip_lib, host_shape):
    """ColliderSampling::DynamicContactSampling (fluids_pipeline.rs:193-259) across a slab face: a spinning ball sits on the cut
    while the fluid drifts through it.  Each rank emits boundary particles for the fluid particles it owns and all ranks assemble
    the same table (World::dist_gather_emitted), ghosts are pushed out of the ball like their owners, and the reaction forces of the
    ranks add up to the undivided world's — per boundary particle, identified by the fluid particle it was projected from.
    host_shape: the arm whose two geometry calls come back to the host (salva_hip_set_boundary_dynamic_sampling_host), on the slabs only
    — the undivided world keeps the device's ball, which that arm matches bit for bit (test_host_shape_gpu.py)."""
    from salva_amd.coupling import ColliderCouplingSet, DynamicContactSampling, HostShapeSampling, RigidBody
    from test_host_shape_gpu import ball_callbacks

    pos, vel, bpos = make_scene()
    nsteps, nranks = 10, 2
    BALL_R = 0.12

    def ball():
        return RigidBody(translation=np.float32([0.02, 0.17, 0.01]), linvel=np.float32([-0.4, 0.1, 0.0]), angvel=np.float32([0.0, 0.5, 3.0]),
                         mass=5.0, principal_inertia=np.float32([0.03, 0.03, 0.03]))

    def run(w, host):
        empty = w.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
        coupling = ColliderCouplingSet()
        body = ball()
        sampling = HostShapeSampling(*ball_callbacks(body, BALL_R)) if host else DynamicContactSampling(("ball", BALL_R))
        coupling.register_coupling(empty, "ball", body, sampling)
        return empty, coupling

    w = LiquidWorld(solver(), R, SF)
    f = Fluid(pos, R, 1000.0)
    f.velocities = vel
    f.nonpressure_forces.append(XSPHViscosity(0.5, 0.3))
    w.add_fluid(f)
    w.add_boundary(Boundary(bpos))
    b, coupling = run(w, False)
    counts_ref = []
    for _ in range(nsteps):
        w.sync_to_device()
        coupling.update_boundaries(w)  # (the body is not integrated: the same pose every step, on every rank)
        w.step(DT, G)
        counts_ref.append(b.num_particles())
    ref_p, ref_v = f.positions.copy(), f.velocities.copy()
    _, ref_src = b.sources()
    ref_force = b.forces.astype(np.float64)
    ref_pts = b.positions.copy()
    assert counts_ref[-1] > 40, f"only {counts_ref[-1]} boundary particles: the ball was meant to sit in the fluid"

    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, nranks)
    owner = dist.owner_of(cx, slabs)
    comms = dist.Comm.loopback(nranks)
    offsets = np.concatenate([[0], np.cumsum([(owner == r).sum() for r in range(nranks)])])
    gid_to_global = np.concatenate([np.nonzero(owner == r)[0] for r in range(nranks)])
    out, errors = [None] * nranks, [None] * nranks

    def rank_main(r):
        try:
            wr = LiquidWorld(solver(), R, SF)
            mine = np.nonzero(owner == r)[0]
            fr = Fluid(pos[mine], R, 1000.0)
            fr.velocities = vel[mine]
            fr.nonpressure_forces.append(XSPHViscosity(0.5, 0.3))
            wr.add_fluid(fr)
            wr.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[r], r, nranks)]))
            br, cr = run(wr, host_shape)
            wr.set_domain(comms[r], slabs[r][0], slabs[r][1], int(offsets[r]))
            counts = []
            for _ in range(nsteps):
                wr.sync_to_device()
                cr.update_boundaries(wr)
                wr.step(DT, G)
                counts.append(br.num_particles())
            slot, gid = br.sources()
            out[r] = (wr.owned(), counts, slot, gid, br.positions.copy(), br.velocities.copy(), br.forces.astype(np.float64))
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for e in errors:
        if e is not None:
            raise e
    assert not any(t.is_alive() for t in threads), "a rank hung"
    for c in comms:
        c.destroy()
    # the same table on both ranks, in rank order
    (_, counts0, slot0, gid0, pts0, bv0, force0), (_, counts1, slot1, gid1, pts1, bv1, force1) = out
    assert counts0 == counts1
    assert np.array_equal(gid0, gid1) and np.array_equal(pts0, pts1) and np.array_equal(bv0, bv1) and (slot0 == 0).all() and (slot1 == 0).all()
    src = gid_to_global[gid0]
    assert len(np.unique(src)) == len(src)
    final_owner = np.full(len(pos), -1)
    got_p, got_v = np.full_like(pos, np.nan), np.full_like(vel, np.nan)
    for r in range(nranks):
        gid, p, v, _slot = out[r][0]
        final_owner[gid_to_global[gid]] = r
        got_p[gid_to_global[gid]] = p
        got_v[gid_to_global[gid]] = v
    emitted_by = final_owner[src]
    assert (emitted_by == 0).sum() > 10 and (emitted_by == 1).sum() > 10, "the ball was meant to straddle the cut"
    assert (np.diff(emitted_by) >= 0).all(), "rows are in rank order"
    # against the undivided world: the same fluid particles emitted (a particle at the very edge of the reach may differ), the same
    # points, and per point the ranks' forces add up to the single force
    assert max(abs(a - b0) for a, b0 in zip(counts0, counts_ref)) <= 2, f"{counts0} vs {counts_ref}"
    common, i_ref, i_got = np.intersect1d(ref_src, src, return_indices=True)
    assert len(common) >= len(ref_src) - 2 and len(common) >= len(src) - 2
    assert np.abs(pts0[i_got] - ref_pts[i_ref]).max() < 2e-4 * H
    scale = np.abs(ref_force).max()
    assert scale > 0
    summed = force0 + force1
    assert np.abs(force0[i_got]).max() > 0.01 * scale and np.abs(force1[i_got]).max() > 0.01 * scale, "both slabs press on the ball"
    assert np.abs(summed[i_got] - ref_force[i_ref]).max() < 5e-3 * scale, \
        f"summed forces differ by {np.abs(summed[i_got] - ref_force[i_ref]).max() / scale:.2e} of the largest"
    dp, dv = np.abs(got_p - ref_p).max(), np.abs(got_v - ref_v).max()
    assert dp < 2e-4 * H, f"positions differ by {dp / H:.2e} h"
    assert dv < 5e-3, f"velocities differ by {dv:.2e} m/s"


def test_rebalance_recuts_the_slabs_and_keeps_the_physics(hip_lib):
    """salva_hip_rebalance: three ranks start from deliberately lopsided cuts (one rank owns 60 % of the particles), re-cut
    every second step, and must end up within a few per cent of N / 3 each — while the particle states keep following the
    undivided domain (the re-cut only changes who owns what; boundaries are re-uploaded for the new slabs by the caller)."""
    pos, vel, bpos = make_scene(nx=60)
    nsteps, nranks = 10, 3
    ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, False)
    cx = dist.cell_x(pos, H)
    lo, hi = int(cx.min()), int(cx.max())
    span = hi - lo + 1
    slabs = [(lo, lo + span * 6 // 10 - 1), (lo + span * 6 // 10, lo + span * 8 // 10 - 1), (lo + span * 8 // 10, hi)]
    owner = dist.owner_of(cx, slabs)
    first_counts = [int((owner == r).sum()) for r in range(nranks)]
    assert max(first_counts) > 1.5 * len(pos) / nranks
    comms = dist.Comm.loopback(nranks)
    offsets = np.concatenate([[0], np.cumsum(first_counts)])
    order = np.concatenate([np.nonzero(owner == r)[0] for r in range(nranks)])
    results, errors, counts, final_slabs = [None] * nranks, [None] * nranks, [None] * nranks, [None] * nranks

    def rank_main(r):
        try:
            w = LiquidWorld(solver(), R, SF)
            mine = np.nonzero(owner == r)[0]
            f = Fluid(pos[mine], R, 1000.0)
            f.velocities = vel[mine]
            f.nonpressure_forces.extend(FORCES["make"]())
            w.add_fluid(f)
            b = w.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[r], r, nranks)]))
            w.set_domain(comms[r], slabs[r][0], slabs[r][1], int(offsets[r]))
            my = slabs[r]
            for k in range(nsteps):
                w.step(DT, G)
                if k % 2 == 1 and k + 1 < nsteps:
                    my = w.rebalance()
                    w.remove_boundary(b)
                    b = w.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, my, r, nranks)]))
            results[r] = w.owned()
            counts[r] = len(results[r][0])
            final_slabs[r] = my
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for e in errors:  # a rank that raised leaves its peers waiting in the next exchange: report the cause first
        if e is not None:
            raise e
    assert not any(t.is_alive() for t in threads), "a rank hung"
    for c in comms:
        c.destroy()
    assert sum(counts) == len(pos)
    # cuts sit on cell planes (~ len(pos) / span particles each): within two planes of the ideal
    per_plane = len(pos) / span
    assert max(abs(c - len(pos) / nranks) for c in counts) < 2.5 * per_plane, (first_counts, counts, final_slabs)
    assert final_slabs[0][1] + 1 == final_slabs[1][0] and final_slabs[1][1] + 1 == final_slabs[2][0]
    got_p = np.full_like(pos, np.nan)
    for gid, p, v, _slot in results:
        got_p[order[gid]] = p
    assert np.isfinite(got_p).all()
    assert np.abs(got_p - ref_p).max() < 2e-4 * H, f"positions differ by {np.abs(got_p - ref_p).max() / H:.2e} h"


def test_rebalance_moves_cuts_by_whole_slabs_between_ranks_of_different_extent(hip_lib):
    """a re-cut invalidates the shortcut by which dist_prepare derives the next cell bounding box (arrivals within
    two planes of a face, inside the sender's previous y/z box).  Four ranks; ranks 0 and 1 start with thin slabs that hold a
    LOW, WIDE part of the fluid, rank 3 with a thick slab under a tall column (larger y extent); the first re-cut moves two
    adjacent cuts by more than a slab width, so rank 1 receives particles from deep inside rank 2 whose y range it has never
    seen, and mirrors some of them to rank 0 in the same step.  Must neither raise (flags 2 / 4) nor lose particles, and
    the states must keep following the undivided domain."""
    pos, vel, bpos = make_scene(nx=56, ny=6, nz=10, seed=9)
    # a tall column on top of the last third of the block
    col = scenes.jitter(scenes.cube_fluid_positions(16, 10, 10, R), 0.1 * R, 10).astype(np.float32)
    col[:, 0] += pos[:, 0].max() - col[:, 0].max()
    col[:, 1] += pos[:, 1].max() - col[:, 1].min() + 2 * R
    col[:, 2] += pos[:, 2].min() - col[:, 2].min()
    pos = np.concatenate([pos, col])
    vel = np.concatenate([vel, np.zeros_like(col)])
    vel[:, 0] = 0.0  # (no drift: the owners change through the re-cut only)
    nsteps, nranks = 8, 4
    ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, False)
    cx = dist.cell_x(pos, H)
    lo, hi = int(cx.min()), int(cx.max())
    slabs = [(lo, lo + 3), (lo + 4, lo + 7), (lo + 8, lo + 8 + (hi - lo - 8) // 2), (lo + 9 + (hi - lo - 8) // 2, hi)]
    owner = dist.owner_of(cx, slabs)
    first_counts = [int((owner == r).sum()) for r in range(nranks)]
    comms = dist.Comm.loopback(nranks)
    offsets = np.concatenate([[0], np.cumsum(first_counts)])
    order = np.concatenate([np.nonzero(owner == r)[0] for r in range(nranks)])
    results, errors, slabs_seen = [None] * nranks, [None] * nranks, [None] * nranks

    def rank_main(r):
        try:
            w = LiquidWorld(solver(), R, SF)
            mine = np.nonzero(owner == r)[0]
            f = Fluid(pos[mine], R, 1000.0)
            f.velocities = vel[mine]
            f.nonpressure_forces.extend(FORCES["make"]())
            w.add_fluid(f)
            b = w.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[r], r, nranks)]))
            w.set_domain(comms[r], slabs[r][0], slabs[r][1], int(offsets[r]))
            seen = [slabs[r]]
            for k in range(nsteps):
                w.step(DT, G)
                if k in (0, 2, 4):
                    my = w.rebalance()
                    seen.append(my)
                    w.remove_boundary(b)
                    b = w.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, my, r, nranks)]))
            results[r] = w.owned()
            slabs_seen[r] = seen
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for e in errors:
        if e is not None:
            raise e
    assert not any(t.is_alive() for t in threads), "a rank hung"
    for c in comms:
        c.destroy()
    # the first re-cut moved the cut between ranks 1 and 2 by at least a whole (old) slab of rank 1
    assert slabs_seen[2][1][0] - slabs_seen[2][0][0] >= 2, slabs_seen
    assert sum(len(r[0]) for r in results) == len(pos)
    got_p = np.full_like(pos, np.nan)
    for gid, p, v, _slot in results:
        got_p[order[gid]] = p
    assert np.isfinite(got_p).all()
    assert np.abs(got_p - ref_p).max() < 2e-4 * H, f"positions differ by {np.abs(got_p - ref_p).max() / H:.2e} h"


def test_queries_in_a_decomposed_run(hip_lib):
    """particles_intersecting_aabb / _shape are per-rank operations in a decomposed run: each rank reports the particles it OWNS
    (ghosts are the neighbour's to report) with their global ids; the union over the ranks is the undivided world's answer."""
    pos, vel, bpos = make_scene()
    nsteps, nranks = 6, 2
    w = LiquidWorld(solver(), R, SF)
    f = Fluid(pos, R, 1000.0)
    f.velocities = vel
    f.nonpressure_forces.extend(FORCES["make"]())
    fh = w.add_fluid(f)
    w.add_boundary(Boundary(bpos))
    for _ in range(nsteps):
        w.step(DT, G)
    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, nranks)
    cut_x = slabs[1][0] * H  # the box and the ball straddle the cut
    p_now = np.array(fh.positions)
    mid = p_now.mean(axis=0)
    box = ((cut_x - 1.5 * H, mid[1] - 2 * H, mid[2] - 2 * H), (cut_x + 1.5 * H, mid[1] + 2 * H, mid[2] + 2 * H))
    ball = ((cut_x + 0.3 * H, mid[1], mid[2]), (0.0, 0.0, 0.0, 1.0), ("ball", 2.2 * H))
    ref_box = sorted(i for k, _, i in w.particles_intersecting_aabb(*box) if k == "fluid")
    ref_ball = sorted(i for k, _, i in w.particles_intersecting_shape(*ball) if k == "fluid")
    assert len(ref_box) > 50 and len(ref_ball) > 50
    del w

    owner = dist.owner_of(cx, slabs)
    comms = dist.Comm.loopback(nranks)
    offsets = np.concatenate([[0], np.cumsum([int((owner == r).sum()) for r in range(nranks)])])
    order = np.concatenate([np.nonzero(owner == r)[0] for r in range(nranks)])  # global id -> index in `pos`
    got_box, got_ball, errors = [None] * nranks, [None] * nranks, [None] * nranks

    def rank_main(r):
        try:
            wr = LiquidWorld(solver(), R, SF)
            mine = np.nonzero(owner == r)[0]
            fr = Fluid(pos[mine], R, 1000.0)
            fr.velocities = vel[mine]
            fr.nonpressure_forces.extend(FORCES["make"]())
            wr.add_fluid(fr)
            wr.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[r], r, nranks)]))
            wr.set_domain(comms[r], slabs[r][0], slabs[r][1], int(offsets[r]))
            for _ in range(nsteps):
                wr.step(DT, G)
            got_box[r] = [i for k, _, i in wr.particles_intersecting_aabb(*box) if k == "fluid"]
            got_ball[r] = [i for k, _, i in wr.particles_intersecting_shape(*ball) if k == "fluid"]
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for e in errors:
        if e is not None:
            raise e
    for c in comms:
        c.destroy()
    assert all(len(g) > 0 for g in got_box), "both ranks own part of the box"
    union_box = sorted(int(order[g]) for r in range(nranks) for g in got_box[r])
    union_ball = sorted(int(order[g]) for r in range(nranks) for g in got_ball[r])
    assert len(set(union_box)) == len(union_box), "a particle was reported by two ranks"
    # positions agree to rounding between the two runs: a particle within 1e-4 h of the query surface may flip
    assert len(set(union_box) ^ set(ref_box)) <= 2 and len(set(union_ball) ^ set(ref_ball)) <= 2, (len(union_box), len(ref_box))


def test_particles_are_created_and_deleted_in_a_running_decomposed_world(hip_lib):
    """An emitter and a sink in a decomposed run (faucet3.rs:69-104 in miniature): after 4 steps every rank adds a small block
    above its part of the fluid (collective salva_hip_add_particles) and all ranks delete the same list of ids — a band of
    particles that straddles the cut (collective salva_hip_delete_owned).  The undivided world does the same with
    `Fluid::add_particles` and `delete_particle_at_next_timestep`; afterwards both run on and must agree like every other slab
    test (ids: new particles are numbered after the largest id, rank by rank)."""
    pos, vel, bpos = make_scene()
    vel[:, 0] -= 2.0  # (no drift: the emitters stay above their ranks)
    n0, nranks, before, after = len(pos), 2, 4, 6
    cx = dist.cell_x(pos, H)
    slabs = dist.split_slabs(cx, nranks)
    owner = dist.owner_of(cx, slabs)
    order = np.concatenate([np.nonzero(owner == r)[0] for r in range(nranks)])  # global id -> index in `pos`
    inv = np.empty(n0, np.int64)
    inv[order] = np.arange(n0)                                                   # index in `pos` -> global id
    offsets = np.concatenate([[0], np.cumsum([int((owner == r).sum()) for r in range(nranks)])])
    top = pos[:, 1].max()
    # one 4x3x4 block per rank, two cells above the fluid, well inside the rank's slab
    blocks = []
    for r in range(nranks):
        xm = 0.5 * (slabs[r][0] + slabs[r][1] + 1) * H
        b = scenes.cube_fluid_positions(4, 3, 4, R) + np.float32([xm, top + 2 * H, pos[:, 2].mean()])
        blocks.append(b.astype(np.float32))
    # the sink: every particle whose x lies within 0.6 h of the cut and in the upper half of the block at the start
    cut_x = slabs[1][0] * H
    doomed = np.nonzero((np.abs(pos[:, 0] - cut_x) < 1.1 * H) & (pos[:, 1] > np.median(pos[:, 1])))[0]
    assert len(doomed) >= 100 and {0, 1} <= set(owner[doomed].tolist())

    # ---- undivided world
    w = LiquidWorld(solver(), R, SF)
    f = Fluid(pos, R, 1000.0)
    f.velocities = vel
    f.nonpressure_forces.extend(FORCES["make"]())
    fh = w.add_fluid(f)
    w.add_boundary(Boundary(bpos))
    for _ in range(before):
        w.step(DT, G)
    for b in blocks:
        fh.add_particles(b)
    for i in doomed:
        fh.delete_particle_at_next_timestep(int(i))
    ref_stats = [w.step(DT, G) for _ in range(after)]
    keep = np.ones(n0 + sum(len(b) for b in blocks), bool)
    keep[doomed] = False
    ref_pos = np.array(fh.positions)
    assert len(ref_pos) == int(keep.sum())
    # id of every surviving particle of the undivided world, in its (compacted) host order
    new_ids = n0 + np.arange(sum(len(b) for b in blocks))
    ref_ids = np.concatenate([inv, new_ids])[keep]
    del w

    # ---- two slabs
    comms = dist.Comm.loopback(nranks)
    results, errors, stats = [None] * nranks, [None] * nranks, [None] * nranks
    doomed_ids = inv[doomed].astype(np.uint32)

    def rank_main(r):
        try:
            wr = LiquidWorld(solver(), R, SF)
            mine = np.nonzero(owner == r)[0]
            fr = Fluid(pos[mine], R, 1000.0)
            fr.velocities = vel[mine]
            fr.nonpressure_forces.extend(FORCES["make"]())
            hr = wr.add_fluid(fr)
            wr.add_boundary(Boundary(bpos[dist.boundary_subset(bpos, H, slabs[r], r, nranks)]))
            wr.set_domain(comms[r], slabs[r][0], slabs[r][1], int(offsets[r]))
            for _ in range(before):
                wr.step(DT, G)
            wr.add_owned(hr, blocks[r])
            left = wr.delete_owned(doomed_ids)  # the same list on every rank: each removes what it owns
            assert left == int((owner == r).sum()) + len(blocks[r]) - int((owner[doomed] == r).sum()), left
            stats[r] = [wr.step(DT, G) for _ in range(after)]
            results[r] = wr.owned()
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    for e in errors:
        if e is not None:
            raise e
    assert not any(t.is_alive() for t in threads), "a rank hung"
    for c in comms:
        c.destroy()
    got = {}
    for r in range(nranks):
        gid, p, v, _slot = results[r]
        for g, x in zip(gid.tolist(), p):
            assert g not in got, "a particle is owned twice"
            got[g] = x
    assert sorted(got) == sorted(ref_ids.tolist()), "the sets of surviving ids differ"
    got_pos = np.stack([got[int(g)] for g in ref_ids])
    err = np.linalg.norm(got_pos - ref_pos, axis=1).max()
    assert err < 2e-4 * H * after, f"positions differ by {err / H:.2e} h"
    for k in range(after):
        assert stats[0][k].n_divergence_iters == stats[1][k].n_divergence_iters
        assert abs(stats[0][k].n_divergence_iters - ref_stats[k].n_divergence_iters) <= 2 and abs(stats[0][k].n_pressure_iters - ref_stats[k].n_pressure_iters) <= 1
        assert sum(int(stats[r][k].nparticles) for r in range(nranks)) == len(ref_pos)


def test_cpp_mirror_runs_a_decomposed_world(hip_lib):
    """examples/slabs3.cpp: salva::Comm + LiquidWorld::set_domain / owned / delete_owned of include/salva_hip.hpp — two and three
    loopback slabs driven from C++ host threads; the program itself checks that every particle ends up owned exactly once, that
    the ranks took the same solver iterations, and that a collective removal leaves the expected number of particles."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "slabs3"], check=True, capture_output=True)
    for args in (["2", "10"], ["3", "6"]):
        r = subprocess.run([os.path.join(root, "examples", "slabs3")] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "slabs3 OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("classes", [False, True])
def test_strays_far_out_in_y_and_z_fold_away_in_a_decomposed_run(hip_lib, monkeypatch, classes):
    """Round 6 (missing 3): the reference's hash grid never cares where a stray particle is
    (/root/reference/src/geometry/hgrid.rs:22-63); the dense cell table did, wherever it could not fold — decomposed runs among them
    (ghost planes are found by absolute x-cell).  They now fold y and z: fifty particles that have left the tank hundreds of cells
    below and beside it ride on the torus, the slabs compute what the undivided (and, here, unfolded) world computes, and nobody
    dies of E_CAPACITY.  SALVA_HIP_FOLD_CELLS=16 forces the fold on this small scene: x — 48 cells — must stay whole on the slabs.
    `classes`: the same with the strays' slots in launches of their own (SALVA_HIP_CLASSES=1) — in the undivided world, where a
    speculative apply's convergence test once ran per launch (fixed in round 6), and on the slabs, whose evaluate passes launch
    twice already (interior | border tiles, World::evaluate_split): four launches per pass, the same sums."""
    if classes:
        monkeypatch.setenv("SALVA_HIP_CLASSES", "1")
    else:
        monkeypatch.delenv("SALVA_HIP_CLASSES", raising=False)
    pos, vel, bpos = make_scene(nx=48, ny=10, nz=10, seed=9)
    rng = np.random.default_rng(3)
    stray = np.zeros((50, 3), np.float32)
    stray[:, 0] = rng.uniform(pos[:, 0].min(), pos[:, 0].max(), 50)
    stray[:25, 1] = rng.uniform(-60.0, -20.0, 25)   # fallen out of the tank: 200 .. 600 cells below
    stray[25:, 1] = rng.uniform(0.2, 0.6, 25)
    stray[25:, 2] = rng.uniform(15.0, 40.0, 25)      # sprayed sideways
    stray[:25, 2] = rng.uniform(pos[:, 2].min(), pos[:, 2].max(), 25)
    svel = np.zeros((50, 3), np.float32)
    svel[:, 1] = -3.0
    pos2, vel2 = np.concatenate([pos, stray]), np.concatenate([vel, svel])
    nsteps = 10
    monkeypatch.setenv("SALVA_HIP_NO_FOLD", "1")
    ref_p, ref_v, ref_stats = run_single(pos2, vel2, bpos, nsteps, False)
    monkeypatch.delenv("SALVA_HIP_NO_FOLD")
    monkeypatch.setenv("SALVA_HIP_FOLD_CELLS", "16")
    got_p, got_v, stats, seen, counts, slabs = run_slabs(pos2, vel2, bpos, nsteps, 3, False)
    assert (seen == 1).all()
    for k in range(nsteps):
        it = {(s[k].n_divergence_iters, s[k].n_pressure_iters) for s in stats}
        assert len(it) == 1
        assert sum(int(s[k].ncontacts) for s in stats) == int(ref_stats[k].ncontacts)  # the contact SETS are the unfolded grid's
    assert np.abs(got_p - ref_p).max() < 2e-4 * H and np.abs(got_v - ref_v).max() < 5e-3
    assert np.abs(got_p[-50:] - ref_p[-50:]).max() < 1e-5  # (the strays themselves: free flight)


def test_decomposed_solves_run_their_applies_beside_the_all_reduce(hip_lib, monkeypatch):
    """Round 6: in a decomposed divergence solve the all-reduced convergence test and the apply pass no longer
    wait for each other — the apply runs speculatively into the second w buffer on the second stream while error sums -> all-reduce
    -> decision run on the main one (World::run_solve; the single domain's double buffer, dfsph.hip spec_decide).  Same sums, same
    decisions: three slabs with it and without it (SALVA_HIP_NO_SPEC_DIST=1) end bit for bit in the same state, iteration by
    iteration — and both agree with the undivided world as before."""
    pos, vel, bpos = make_scene(nx=44, ny=12, nz=10, seed=21)
    vel[:, 1] -= np.float32(2.5)  # driven into the tank's floor: the divergence solve needs 10 .. 50 iterations step after step
    nsteps = 12
    ref_p, ref_v, ref_stats = run_single(pos, vel, bpos, nsteps, False)
    monkeypatch.setenv("SALVA_HIP_NO_SPEC_DIST", "1")
    p0, v0, s0, seen0, _, _ = run_slabs(pos, vel, bpos, nsteps, 3, False)
    monkeypatch.delenv("SALVA_HIP_NO_SPEC_DIST")
    p1, v1, s1, seen1, _, _ = run_slabs(pos, vel, bpos, nsteps, 3, False)
    its = [s1[0][k].n_divergence_iters for k in range(nsteps)]
    assert sum(i >= 4 for i in its[:-1]) >= 5, its  # the speculative path switches on from four iterations (of the previous step) on
    assert (seen0 == 1).all() and (seen1 == 1).all()
    for k in range(nsteps):
        a = {(s[k].n_divergence_iters, s[k].n_pressure_iters, float(s[k].divergence_error)) for s in s0}
        b = {(s[k].n_divergence_iters, s[k].n_pressure_iters, float(s[k].divergence_error)) for s in s1}
        assert len(a) == 1 and a == b, (k, a, b)
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    assert np.abs(p1 - ref_p).max() < 2e-4 * H and np.abs(v1 - ref_v).max() < 5e-3
