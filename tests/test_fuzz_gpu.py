"""Differential test of the object bookkeeping around the hot path: a seeded random sequence of LiquidWorld operations
(step, Fluid::add_particles, delete_particle_at_next_timestep, remove_fluid, add_fluid, host edits of velocities) is
applied to the HIP world and to the oracle; after every step particle counts and contacts must be equal and the states
within the parity tolerance (contact sets up to pairs sitting exactly on d = h).  Covers the interplay the single-purpose tests cannot: which arrays travel, which solver
buffers are compacted, inherited (remove_fluid's positional buffers) or restarted, and when."""
import numpy as np
import pytest

from parity import DT, GRAVITY
from oracle import oracle as O
from salva_amd import (Akinci2013SurfaceTension, ArtificialViscosity, Boundary, DFSPHSolver, Fluid, He2014SurfaceTension, IISPHSolver,
                       InteractionGroups, LiquidWorld, WCSPHSurfaceTension, XSPHViscosity, scenes)

pytestmark = pytest.mark.gpu

R = 0.025


def _block(rng, nx, ny, nz, origin):
    p = scenes.jitter(scenes.cube_fluid_positions(nx, ny, nz, R), 0.05 * R, seed=int(rng.integers(1 << 30)))
    return (p + np.float32(origin)).astype(np.float32)


@pytest.mark.parametrize("solver,seed", [("dfsph", 1), ("dfsph", 2), ("iisph", 3), ("dfsph", 4), ("iisph", 5), ("dfsph", 6)])
def test_random_operation_sequences_match_oracle(solver, seed):
    rng = np.random.default_rng(seed)
    o = O.OracleWorld(R, 2.0, O.DFSPH if solver == "dfsph" else O.IISPH)
    def new_world():
        return LiquidWorld(DFSPHSolver() if solver == "dfsph" else IISPHSolver(), R, 2.0)

    w = new_world()
    handles = []  # handles[slot] mirrors the FluidSet's dense order (swap-remove)
    force_offset = 0.0  # boundary forces accumulated by worlds that were replaced through a checkpoint

    def add_fluid(origin, density):
        pos = _block(rng, 4, 4, 4, origin)
        vel = scenes.random_velocities(len(pos), 0.1, seed=int(rng.integers(1 << 30)))
        groups = [(1, 0xFFFFFFFF), (2, 0xFFFFFFFF), (4, 0xFFFFFFFB)][int(rng.integers(3))]  # the last one ignores its own kind
        f = Fluid(pos, R, density, InteractionGroups(*groups))
        f.velocities = vel
        k = o.add_fluid(pos, density, vel, *groups)
        # a random list of built-in forces, in list order on both sides (predict_advection walks the list, dfsph_solver.rs:580-603)
        for kind in rng.choice(["xsph", "artificial", "akinci", "he2014", "wcsph", "none"], size=2):
            if kind == "xsph":
                f.nonpressure_forces.append(XSPHViscosity(0.5, 0.2)); o.add_xsph(k, 0.5, 0.2)
            elif kind == "artificial":
                f.nonpressure_forces.append(ArtificialViscosity(1.0, 0.0)); o.add_artificial_viscosity(k, 1.0, 0.0)
            elif kind == "akinci":
                f.nonpressure_forces.append(Akinci2013SurfaceTension(0.5, 2.0)); o.add_akinci2013(k, 0.5, 2.0)
            elif kind == "he2014":
                f.nonpressure_forces.append(He2014SurfaceTension(0.5, 0.2)); o.add_he2014(k, 0.5, 0.2)
            elif kind == "wcsph":
                f.nonpressure_forces.append(WCSPHSurfaceTension(0.1, 0.0)); o.add_wcsph_tension(k, 0.1, 0.0)
        handles.append(w.add_fluid(f))
        assert k == len(handles) - 1

    floor = scenes.plane_lattice(40, 12, 0.0, R, -6 * 2 * R + R, -6 * 2 * R + R, layers=1)
    bounds = [w.add_boundary(Boundary(floor, wants_forces=True))]  # bounds[slot]: the BoundarySet's dense order
    o.add_boundary(floor, wants_forces=True)
    add_fluid([0.0, 0.25, 0.0], 1000.0)
    add_fluid([0.5, 0.25, 0.0], 800.0)
    nsteps, log = 0, []
    next_x = 1.0
    for _ in range(60):
        op = rng.choice(["step", "step", "step", "add_particles", "delete", "remove_fluid", "add_fluid", "set_velocities",
                         "add_boundary", "move_boundary", "remove_boundary", "far_particle", "query", "checkpoint"])
        log.append(op)
        if op == "step":
            # dt changes exercise the TimestepManager lag (inv_dt of the previous substep), dt = 0 the no-substep path
            dt = float(rng.choice([DT, DT, DT / 2, 0.0]))
            g = GRAVITY if rng.random() < 0.8 else (1.0, -9.81, 0.5)
            so, sh = o.step(dt, g), w.step(dt, g)
            if dt == 0.0:
                continue
            nsteps += 1
            # contacts are decided on states that agree to ~1e-6 r: a pair sitting on d = h may flip (each flip = 2 directed contacts)
            assert abs(so.ncontacts - sh.ncontacts) <= 6, log
            for k, h in enumerate(handles):
                assert o.fluid_len(k) == h.num_particles(), log
                ref_p, ref_v = o.fluid_vec(k, "positions"), o.fluid_vec(k, "velocities")
                assert np.abs(h.positions - ref_p).max() < 1e-4 * R * nsteps, (log, k)
                vref = max(np.abs(ref_v).max(), 2 * R / DT * 1e-2)
                assert np.abs(h.velocities - ref_v).max() < 2e-4 * nsteps * vref, (log, k)
                for boundary in (False, True):  # per-particle contact counts (fluid-fluid incl. other fluids, fluid-boundary)
                    dc = w.contact_counts(h, boundary).astype(np.int64) - o.contact_counts(k, boundary).astype(np.int64)
                    assert np.abs(dc).max(initial=0) <= 1 and np.count_nonzero(dc) <= 6, (log, k, boundary)
            fref = o.boundary_vec(0, "forces")  # accumulated since the start (nothing clears them without a coupling manager)
            fscale = np.abs(fref).max()
            if fscale > 0:
                assert np.abs(bounds[0].forces + force_offset - fref).max() < 5e-3 * fscale, log
        elif op == "add_particles" and handles:
            k = int(rng.integers(len(handles)))
            base = handles[k].positions.mean(0) + np.float32([0.0, 0.35, 0.0])
            pos = _block(rng, 2, 2, 2, base)
            vel = np.tile(np.float32([0.0, -0.5, 0.0]), (len(pos), 1)) if rng.random() < 0.5 else None
            handles[k].add_particles(pos, vel)
            o.add_particles(k, pos, vel)
        elif op == "delete" and handles:
            k = int(rng.integers(len(handles)))
            n = handles[k].num_particles()
            for i in rng.choice(n, size=min(5, n), replace=False):
                handles[k].delete_particle_at_next_timestep(int(i))
                o.delete_particle_at_next_timestep(k, int(i))
        elif op == "remove_fluid" and len(handles) >= 2:
            k = int(rng.integers(len(handles)))
            w.remove_fluid(handles[k])
            o.remove_fluid(k)
            handles[k] = handles[-1]
            handles.pop()
        elif op == "add_fluid" and len(handles) < 4:
            add_fluid([next_x, 0.25, 0.0], float(rng.choice([600.0, 1000.0, 1200.0])))
            next_x += 0.5
        elif op == "set_velocities" and handles:
            k = int(rng.integers(len(handles)))
            v = handles[k].velocities.copy()
            v[:, 0] += np.float32(0.2)
            handles[k].velocities = v
            o.set_fluid_velocities(k, v)
        elif op == "checkpoint" and len(log) >= 2 and log[-2] == "step":
            # restart from a checkpoint in a clean window (right after a step: nothing pending): a fresh world with the same
            # objects takes over; the oracle just carries on
            st = w.checkpoint()
            force_offset = force_offset + bounds[0].forces
            w2 = new_world()
            new_handles, new_bounds = [], []
            for k, hdl in enumerate(handles):
                f2 = Fluid(st[f"fluid{k}_positions"], R, hdl.density0, hdl.interaction_groups)
                f2.nonpressure_forces = list(hdl.nonpressure_forces)
                new_handles.append(w2.add_fluid(f2))
            for k, b in enumerate(bounds):
                new_bounds.append(w2.add_boundary(Boundary(st[f"boundary{k}_positions"], b.interaction_groups, wants_forces=b.wants_forces)))
            w2.restore(st)
            w, handles, bounds = w2, new_handles, new_bounds
        elif op == "far_particle" and handles:
            # a stray particle tens of metres away: thousands of empty tiles in the bounding box (compact tile tables)
            k = int(rng.integers(len(handles)))
            pos = np.float32([[float(rng.uniform(15, 30)), float(rng.uniform(0.5, 3)), float(rng.uniform(-3, 3))]])
            handles[k].add_particles(pos)
            o.add_particles(k, pos)
        elif op == "query" and handles:
            # LiquidWorld::particles_intersecting_aabb against a brute-force evaluation on the oracle's positions
            k = int(rng.integers(len(handles)))
            c = o.fluid_vec(k, "positions").mean(0).astype(np.float32) if o.fluid_len(k) == handles[k].num_particles() else np.zeros(3, np.float32)
            mins, maxs = c - np.float32([0.08, 0.2, 0.06]), c + np.float32([0.05, 0.03, 0.09])
            got = sorted((kind, handles.index(obj) if kind == "fluid" else bounds.index(obj), i)
                         for kind, obj, i in w.particles_intersecting_aabb(mins, maxs))
            ref = []
            for kind, objs in (("fluid", handles), ("boundary", bounds)):
                for idx, obj in enumerate(objs):
                    p = np.asarray(obj.positions, np.float32)
                    d = np.maximum(np.maximum(mins - p, p - maxs), np.float32(0))
                    hit = np.nonzero((d * d).sum(1, dtype=np.float32) < np.float32(R) * np.float32(R))[0]
                    ref += [(kind, idx, int(i)) for i in hit]
            assert got == sorted(ref), log
        elif op == "add_boundary" and len(bounds) < 3:
            # a small plate under one of the fluids, between it and the floor
            k = int(rng.integers(len(handles))) if handles else 0
            cx = float(handles[k].positions[:, 0].mean()) if handles else 0.0
            plate = scenes.plane_lattice(6, 6, 0.0, R, cx - 3 * 2 * R + R, -3 * 2 * R + R, layers=1) + np.float32([0.0, 0.08, 0.0])
            bounds.append(w.add_boundary(Boundary(plate)))
            o.add_boundary(plate)
        elif op == "move_boundary" and bounds:
            k = int(rng.integers(len(bounds)))
            pos = np.asarray(bounds[k].positions, np.float32) + np.float32([0.004, 0.0, 0.0])
            vel = np.tile(np.float32([0.004 / DT, 0.0, 0.0]), (len(pos), 1))
            bounds[k].positions = pos
            bounds[k].velocities = vel
            o.set_boundary_particles(k, pos, vel)
        elif op == "remove_boundary" and len(bounds) >= 2:
            k = int(rng.integers(1, len(bounds)))  # the floor stays
            w.remove_boundary(bounds[k])
            o.remove_boundary(k)
            bounds[k] = bounds[-1]
            bounds.pop()
    assert nsteps >= 4, log
