"""User-defined NonPressureForce implementations (solver/nonpressure_force.rs:10-30): the host
callback in the middle of the substep against the oracle running the same callback, and a host re-implementation of
XSPH over the exported contacts against the device's built-in kernel."""
import numpy as np
import pytest

from parity import DT, GRAVITY
from oracle import oracle as O
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, NonPressureForce, XSPHViscosity, _lib, scenes

pytestmark = pytest.mark.gpu

R = 0.025


class CustomForceField(NonPressureForce):
    """examples3d/custom_forces3.rs:67-90: acc += dir / dist towards `origin` where dist > 0.1."""

    def __init__(self, origin):
        self.origin = np.float32(origin)
        self.calls = 0
        self.seen_dt = []

    @staticmethod
    def field(origin, positions):
        d = origin - positions
        dist = np.sqrt((d * d).sum(1, dtype=positions.dtype))
        ok = dist > 0.1  # Unit::try_new_and_get(v, 0.1)
        out = np.zeros_like(positions)
        out[ok] = d[ok] / dist[ok, None] / dist[ok, None]
        return out

    def solve(self, timestep, kernel_radius, fluid_fluid_contacts, fluid_boundaries_contacts, fluid, boundaries, densities):
        self.calls += 1
        self.seen_dt.append((timestep.dt(), timestep.inv_dt()))
        assert abs(kernel_radius - 4 * R) < 1e-7 and len(densities) == fluid.num_particles() and (densities > 0).all()
        fluid.accelerations += self.field(self.origin, fluid.positions)


class HostXSPH(NonPressureForce):
    """xsph_viscosity.rs:31-95 written against the mirror's ParticlesContacts, like a user plugin would."""

    def __init__(self, fc, bc):
        self.fc, self.bc = np.float32(fc), np.float32(bc)

    def solve(self, timestep, kernel_radius, ff, fb, fluid, boundaries, densities):
        inv_dt = np.float32(timestep.inv_dt())
        bvol = [b.volumes for b in boundaries]  # one download each (a property access goes to the device)
        bvelocities = [b.velocities for b in boundaries]
        for i in range(fluid.num_particles()):
            added_fluid_vel = np.zeros(3, np.float32)
            for c in ff.particle_contacts(i):
                if c.i_model == c.j_model:
                    added_fluid_vel += (fluid.velocities[c.j] - fluid.velocities[c.i]) * (self.fc * np.float32(c.weight) * fluid.particle_mass(c.j) / densities[c.j])
            added_boundary_vel = np.zeros(3, np.float32)
            for c in fb.particle_contacts(i):
                delta = (bvelocities[c.j_model][c.j] - fluid.velocities[c.i]) * (self.bc * np.float32(c.weight) * bvol[c.j_model][c.j] * fluid.density0 / densities[c.i])
                added_boundary_vel += delta
            fluid.accelerations[i] += added_fluid_vel * inv_dt + added_boundary_vel * inv_dt


def _scene():
    pos = scenes.jitter(scenes.cube_fluid_positions(7, 7, 7, R), 0.1 * R, seed=21)
    pos[:, 1] += np.float32(7 * R + 2 * R)
    vel = scenes.random_velocities(len(pos), 0.3, seed=22)
    floor = scenes.plane_lattice(12, 12, 0.0, R, -6 * 2 * R + R, -6 * 2 * R + R, layers=1)
    return pos, vel, floor


def test_custom_force_field_matches_oracle_with_the_same_callback():
    pos, vel, floor = _scene()
    origin = np.float32([0.3, 0.4, -0.2])
    nsteps = 8
    # oracle: XSPH, then the custom field, in list order
    o = O.OracleWorld(R, 2.0, O.DFSPH)
    fo = o.add_fluid(pos, 1000.0, vel)
    o.add_xsph(fo, 0.5, 0.0)
    ocalls = []

    def ofield(world, f, positions, velocities, densities, accelerations):
        ocalls.append(len(positions))
        accelerations += CustomForceField.field(origin, positions.astype(np.float32)).astype(np.float64)

    o.add_custom_force(fo, ofield)
    o.add_boundary(floor)
    # device
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    fl.velocities = vel
    force = CustomForceField(origin)
    fl.nonpressure_forces += [XSPHViscosity(0.5, 0.0), force]
    h = w.add_fluid(fl)
    w.add_boundary(Boundary(floor))
    for _ in range(nsteps):
        o.step(DT, GRAVITY)
        w.step(DT, GRAVITY)
    assert force.calls == nsteps == len(ocalls)
    # timestep.dt() lags by one substep inside predict_advection: 0 on the first, then dt (dfsph_solver.rs:693-702)
    assert force.seen_dt[0] == (0.0, 0.0) and abs(force.seen_dt[1][0] - DT) < 1e-9 and abs(force.seen_dt[1][1] - 1 / DT) < 1e-3
    ref_p, ref_v = o.fluid_vec(fo, "positions"), o.fluid_vec(fo, "velocities")
    assert np.abs(h.positions - ref_p).max() < 1e-4 * R * nsteps
    vref = max(np.abs(ref_v).max(), 2 * R / DT * 1e-2)
    assert np.abs(h.velocities - ref_v).max() < 1e-4 * nsteps * vref
    # and the field did pull the fluid: compare with a run without it
    w2 = LiquidWorld(DFSPHSolver(), R, 2.0)
    f2 = Fluid(pos, R, 1000.0)
    f2.velocities = vel
    f2.nonpressure_forces.append(XSPHViscosity(0.5, 0.0))
    h2 = w2.add_fluid(f2)
    w2.add_boundary(Boundary(floor))
    for _ in range(nsteps):
        w2.step(DT, GRAVITY)
    assert np.abs(h.velocities - h2.velocities).max() > 0.05


def test_host_xsph_over_exported_contacts_equals_the_device_kernel():
    pos, vel, floor = _scene()
    bvel = np.zeros_like(floor)
    bvel[:, 0] = 0.5  # a moving floor, so that the boundary term matters
    out = []
    for forces in ([XSPHViscosity(0.5, 0.3)], [HostXSPH(0.5, 0.3)]):
        w = LiquidWorld(DFSPHSolver(), R, 2.0)
        fl = Fluid(pos, R, 1000.0)
        fl.velocities = vel
        fl.nonpressure_forces += forces
        h = w.add_fluid(fl)
        b = Boundary(floor)
        b.velocities = bvel
        w.add_boundary(b)
        for _ in range(3):
            w.step(DT, GRAVITY)
        out.append((h.positions.copy(), h.velocities.copy()))
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-6 * R * 30
    assert np.abs(out[0][1] - out[1][1]).max() < 2e-5 * np.abs(out[0][1]).max()


def test_callback_errors_and_misuse_are_reported():
    pos, vel, floor = _scene()

    class Broken(NonPressureForce):
        def solve(self, *a):
            raise RuntimeError("boom")

    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    fl.nonpressure_forces.append(Broken())
    w.add_fluid(fl)
    with pytest.raises(RuntimeError, match="boom"):
        w.step(DT, GRAVITY)
    # the state accessors only exist inside the callback
    out = np.zeros((len(pos), 3), np.float32)
    import ctypes as C
    rc = w._L.salva_hip_force_get_state(w._h, 0, out.ctypes.data_as(C.POINTER(C.c_float)), None, None)
    assert rc == _lib.E_INVALID

    class Reentrant(NonPressureForce):
        def __init__(self, world):
            self.world, self.rc = world, None

        def solve(self, *a):
            g = (C.c_float * 3)(0, 0, 0)
            self.rc = self.world._L.salva_hip_step(self.world._h, DT, g, None)

    w3 = LiquidWorld(DFSPHSolver(), R, 2.0)
    f3 = Fluid(pos, R, 1000.0)
    r = Reentrant(w3)
    f3.nonpressure_forces.append(r)
    w3.add_fluid(f3)
    w3.step(DT, GRAVITY)
    assert r.rc == _lib.E_INVALID
