"""Opt-in CFL sub-stepping: `TimestepManager::max_substep` (timestep_manager.rs:36-46) with the clamp
`compute_substep` leaves commented out (:90-93), HIP path against the oracle's restatement of the same on the dam break.
Off — the default — a step is one substep, as the reference runs."""
import numpy as np
import pytest

from parity import GRAVITY, Scene, max_norm_diff
from salva_amd import _lib, scenes

pytestmark = pytest.mark.gpu
R = 0.025
DT = 1.0 / 60.0  # a frame: the reference's 1/200 never trips the CFL bound in this scene


def _dam_break(solver="dfsph", forces=(("xsph", 0.5, 0.0),)):
    s = Scene(R, 2.0, solver)
    fluid, shell = scenes.tank(12, 16, 12, R, wall_cells=10)
    s.add_fluid(scenes.jitter(fluid, 0.05 * R, seed=42), None, 1000.0, forces=list(forces))
    s.add_boundary(shell)
    return s


def _falling_block(solver="dfsph"):
    """A stirred block (random velocities, XSPH) released 2.4 m above a floor it does not reach within the test: gravity takes it
    through the CFL bound after a few frame-sized steps and the flow stays smooth, so two f32 runs stay together to rounding and the
    substeps can be compared one by one."""
    s = Scene(R, 2.0, solver)
    fluid = scenes.jitter(scenes.cube_fluid_positions(12, 16, 12, R), 0.05 * R, seed=42)
    fluid[:, 1] += np.float32(2.4 + 16 * R)
    s.add_fluid(fluid, scenes.random_velocities(len(fluid), 0.2, seed=7), 1000.0, forces=[("xsph", 0.5, 0.0)])
    s.add_boundary(scenes.plane_lattice(24, 24, 0.0, R, -24 * R, -24 * R, layers=2))
    return s


def _structure(w, sw, mode):
    assert w.counters.nsubsteps == len(sw)
    # (mode 2 cuts the last substep at the end of the step: only that one may fall below dt / max_num_substeps)
    assert all(DT / 10 * (1 - 1e-6) <= x <= DT * (1 + 1e-6) for x in (sw if mode == 1 else sw[:-1])) and 0 < sw[-1] <= DT * (1 + 1e-6)
    assert sum(sw) >= DT * (1 - 1e-6) and (mode == 1 or abs(sum(sw) - DT) < 1e-6)
    c = w.counters
    assert c.step_time > 0 and c.stages.solver_time > 0  # the timers add up over the substeps


@pytest.mark.parametrize("solver,mode", [("dfsph", 1), ("dfsph", 2), ("iisph", 1)])
def test_cfl_substeps_match_the_oracle(solver, mode):
    """Frame-sized steps (1/60 s) of a falling, stirred block: from about step 8 on every step is cut into 2 ... 10 substeps of
    0.4 * 2r / max_i |v_i + a_i t|.  The flow is smooth, so the device follows the oracle's f32 build substep by substep: the same
    NUMBER of substeps in every step, lengths within max(1e-4 (k + 1), 3 x the oracle's own f32-vs-f64 distance), contacts and
    iteration counts under the same yardstick, final positions to 1e-4 r per substep."""
    s = _falling_block(solver)
    w, (fl,), _ = s.make_hip()
    o = s.make_oracle()          # (one thread: the oracle's summation order, hence its trajectory, is the same in every run)
    o64 = s.make_oracle(f64=True)
    for x in (o, o64):
        x.set_cfl(mode)
    w.set_cfl_substepping(mode)
    w.counters.enable()
    nsub_total, multi = 0, 0
    n = 36
    for k in range(n):
        st = w.step(DT, GRAVITY)
        so = o.step(DT, GRAVITY)
        s64 = o64.step(DT, GRAVITY)
        sw, sr, sd = np.asarray(w.substeps()), np.asarray(o.substeps()), np.asarray(o64.substeps())
        _structure(w, sw, mode)
        assert len(sw) == len(sr) == len(sd), (k, sw, sr, sd)
        tol = np.maximum(1e-4 * (k + 1) * sr, 3 * np.abs(sr - sd))
        if mode == 2:  # (the cut last substep is a difference of nearly equal numbers: absolute, not relative, agreement)
            tol[-1] = max(tol[-1], np.sum(tol[:-1]))
        assert (np.abs(sw - sr) <= tol).all(), (k, sw, sr, sd)
        slack = 0 if k == 0 else max(4, int(2e-5 * so.ncontacts) * (k + 1), 3 * abs(int(so.ncontacts) - int(s64.ncontacts)))
        assert abs(int(st.ncontacts) - int(so.ncontacts)) <= slack, (k, st.ncontacts, so.ncontacts, s64.ncontacts)
        assert abs(st.n_pressure_iters - so.n_press_iters) <= max(1, abs(so.n_press_iters - s64.n_press_iters)), k
        assert abs(st.n_divergence_iters - so.n_div_iters) <= max(1, abs(so.n_div_iters - s64.n_div_iters)), k
        nsub_total += len(sw)
        multi += len(sw) > 1
    assert multi >= 20 and max(len(w.substeps()), 2) >= 2, "the scene never sub-stepped: the test would prove nothing"
    assert len(w.substeps()) >= 5  # by the end a frame is cut into many substeps
    d = max_norm_diff(fl.positions, o.fluid_vec(0, "positions")) / R
    d64 = max_norm_diff(o.fluid_vec(0, "positions"), o64.fluid_vec(0, "positions")) / R
    assert d < max(1e-4 * nsub_total, 3 * d64), (d, d64)


@pytest.mark.parametrize("solver,mode", [("dfsph", 1), ("dfsph", 2), ("iisph", 1)])
def test_cfl_substepping_through_a_dam_break(solver, mode):
    """The same through an impact (a column collapsing in a tank, 30 frame-sized steps).  Here the substep is decided by the fastest
    splash particle and two f32 trajectories part by rounding within a dozen steps — in mode 2 the cut last substep of one step
    even sets the next step's XSPH scale (1 / dt of the previous substep) — so only what does not depend on shadowing a chaotic
    trajectory is asserted: the structure of every step, and that the device sub-steps as much as the oracle does."""
    s = _dam_break(solver)
    w, (fl,), _ = s.make_hip()
    o = s.make_oracle()
    w.set_cfl_substepping(mode)
    o.set_cfl(mode)
    w.counters.enable()
    nd, no = [], []
    for k in range(30):
        w.step(DT, GRAVITY)
        o.step(DT, GRAVITY)
        sw = np.asarray(w.substeps())
        _structure(w, sw, mode)
        nd.append(len(sw)); no.append(len(o.substeps()))
        if k < 10:  # (before the impact the two still agree step by step)
            # (atol: mode 2's cut last substep is a difference of nearly equal numbers)
            assert nd[-1] == no[-1] and np.allclose(sw, o.substeps(), rtol=1e-3, atol=1e-3 * DT), (k, sw, o.substeps())
    assert sum(x > 1 for x in nd) >= 5, nd
    assert abs(sum(nd) - sum(no)) <= max(3, sum(no) // 10), (nd, no)
    assert np.isfinite(fl.positions).all()


def test_cfl_is_off_by_default_and_can_be_switched_off_again():
    s = _dam_break()
    w0, (f0,), _ = s.make_hip()
    w1, (f1,), _ = s.make_hip()
    w1.set_cfl_substepping(1)
    w1.set_cfl_substepping(0)
    for _ in range(30):
        w0.step(DT, GRAVITY)
        w1.step(DT, GRAVITY)
        assert w0.substeps() == w1.substeps() == [np.float32(DT)] and w0.counters.nsubsteps == 1
    assert np.array_equal(f0.positions, f1.positions)
    with pytest.raises(_lib.SalvaHipError):
        w1.set_cfl_substepping(3)
    with pytest.raises(_lib.SalvaHipError):
        w1.set_cfl_substepping(1, 0.4, 3, 2)


def _raft_scene():
    """A stirred block dropped onto a light dynamic raft (two layers of collider-local samples, StaticSampling) beside a kinematic
    paddle: the raft wants forces, and every substep's impulse changes the velocity its boundary particles take in the next."""
    from salva_amd.coupling import RigidBody

    pos = scenes.jitter(scenes.cube_fluid_positions(10, 12, 10, R), 0.05 * R, seed=11)
    pos[:, 1] += np.float32(float(-pos[:, 1].min()) + 4 * R)  # its lowest layer two diameters above the raft's upper one
    vel = scenes.random_velocities(len(pos), 0.1, seed=12)
    vel[:, 1] -= np.float32(2.5)  # fast enough for the CFL bound to cut a frame-sized step
    raft_pts = scenes.plane_lattice(16, 16, 0.0, R, -16 * R + R, -16 * R + R, layers=2)
    paddle_pts = scenes.plane_lattice(3, 8, 0.0, R, -3 * R + R, -8 * R + R, layers=1)[:, [0, 2, 1]]
    raft = RigidBody(translation=np.float32([10 * R, 0.0, 10 * R]), mass=1.5, principal_inertia=np.float32([0.02, 0.04, 0.02]),
                     local_com=np.float32([0.0, -R, 0.0]))
    paddle = RigidBody(translation=np.float32([10 * R, 40 * R, 10 * R]), angvel=np.float32([0.0, 4.0, 0.0]), dynamic=False)
    return pos, vel, raft_pts, paddle_pts, raft, paddle


def test_cfl_substepping_with_a_force_transmitting_collider():
    """`coupling.update_boundaries` and `coupling.transmit_forces` run INSIDE the substep loop in the reference (liquid_world.rs:94-103,
    :146; fluids_pipeline.rs:160-193, 266-287): the impulse of one substep (force * that substep's dt) reaches the body before the next
    substep samples its velocity.  salva_hip_set_coupling_callback puts the host at the same two points of every substep; the oracle
    has the same hook.  Same bodies, same frame-sized steps, CFL mode 1: the substep COUNTS agree step by step, and the raft ends up
    where the oracle's does."""
    import copy

    from oracle import oracle as O
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity
    from salva_amd.coupling import ColliderCouplingSet, StaticSampling

    nsteps = 8

    def run_oracle():
        pos, vel, raft_pts, paddle_pts, raft, paddle = _raft_scene()
        w = O.OracleWorld(R, 2.0, O.DFSPH)
        f = w.add_fluid(pos, 1000.0, vel)
        w.add_xsph(f, 0.5, 0.5)
        empty = np.zeros((0, 3), np.float32)
        bodies = [raft, paddle]
        for pts in (raft_pts, paddle_pts):
            w.set_boundary_sampling(w.add_boundary(empty), pts)
        w.set_cfl(1, 0.4, 1, 10)
        impulses = []

        def manager(phase, dt):
            for b, body in enumerate(bodies):
                if phase == 0:
                    w.update_boundary_pose(b, body.translation, body.rotation, body.linvel, body.angvel, body.center_of_mass(), True, body.is_dynamic())
                elif body.is_dynamic():
                    F, T = w.boundary_wrench(b, body.center_of_mass())
                    body.apply_impulse(np.float32(F) * np.float32(dt))
                    body.apply_torque_impulse(np.float32(T) * np.float32(dt))
                    impulses.append(float(np.linalg.norm(F)) * dt)

        w.set_coupling_callback(manager)
        subs = []
        for _ in range(nsteps):
            w.step(DT, GRAVITY)
            subs.append(w.substeps())
            for body in bodies:
                body.integrate(DT, (0.0, 0.0, 0.0))  # (the raft only feels the fluid)
        return w.fluid_vec(f, "positions"), copy.deepcopy(raft), subs, impulses

    def run_hip():
        pos, vel, raft_pts, paddle_pts, raft, paddle = _raft_scene()
        w = LiquidWorld(DFSPHSolver(), R, 2.0)
        fl = Fluid(pos, R, 1000.0)
        fl.velocities = vel
        fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.5))
        h = w.add_fluid(fl)
        c = ColliderCouplingSet()
        c.register_coupling(w.add_boundary(Boundary(np.zeros((0, 3), np.float32))), "raft", raft, StaticSampling(raft_pts))
        c.register_coupling(w.add_boundary(Boundary(np.zeros((0, 3), np.float32))), "paddle", paddle, StaticSampling(paddle_pts))
        w.set_cfl_substepping(1)
        subs = []
        for _ in range(nsteps):
            w.step_with_coupling(DT, GRAVITY, c)
            subs.append(w.substeps())
            for body in (raft, paddle):
                body.integrate(DT, (0.0, 0.0, 0.0))
        return h.positions.copy(), copy.deepcopy(raft), subs

    opos, oraft, osubs, oimp = run_oracle()
    hpos, hraft, hsubs = run_hip()
    assert max(len(s) for s in osubs) >= 3 and sum(oimp) > 0  # the steps were cut, and the raft was pushed in between
    assert [len(s) for s in hsubs] == [len(s) for s in osubs], (hsubs, osubs)
    for a, b in zip(hsubs, osubs):
        assert np.allclose(a, b, rtol=2e-3, atol=1e-6), (a, b)
    assert abs(oraft.linvel[1]) > 0.05  # it moved
    assert np.abs(hraft.linvel - oraft.linvel).max() <= 2e-3 * max(1.0, float(np.abs(oraft.linvel).max())), (hraft.linvel, oraft.linvel)
    assert np.abs(hraft.angvel - oraft.angvel).max() <= 5e-3 * max(1.0, float(np.abs(oraft.angvel).max())), (hraft.angvel, oraft.angvel)
    assert np.abs(hraft.translation - oraft.translation).max() <= 1e-4
    assert max_norm_diff(hpos, opos) / R <= 2e-2  # (eight frames through an impact: rounding-level drift of a chaotic splash)


def test_cfl_with_a_coupled_boundary_needs_the_callback():
    """Without salva_hip_set_coupling_callback one wrench per step() cannot carry what the reference transmits per substep: the raw
    combination stays refused (the Python and C++ mirrors' step_with_coupling register the callback themselves)."""
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld
    from salva_amd.coupling import ColliderCouplingSet, RigidBody, StaticSampling

    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    w.add_fluid(Fluid(scenes.cube_fluid_positions(6, 6, 6, R) + np.float32([0, 8 * R, 0]), R, 1000.0))
    b = w.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
    c = ColliderCouplingSet()
    c.register_coupling(b, "raft", RigidBody(translation=np.float32([0, 0, 0]), mass=1.0, principal_inertia=np.float32([1, 1, 1])),
                        StaticSampling(scenes.plane_lattice(8, 8, 0.0, R, -8 * R, -8 * R, layers=1)))
    w.set_cfl_substepping(0)
    w.step_with_coupling(DT, GRAVITY, c)  # (uploads the sampling, marks the boundary as wanting forces)
    w.set_cfl_substepping(1)
    with pytest.raises(_lib.SalvaHipError, match="coupling_callback"):
        w.step(DT, GRAVITY)
    w.step_with_coupling(DT, GRAVITY, c)  # with the manager in the loop it runs
    assert w.counters.nsubsteps >= 1
