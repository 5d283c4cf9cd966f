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


def test_cfl_refuses_coupled_boundaries_that_want_forces():
    """The reference transmits a coupled collider's impulse per substep (fluids_pipeline.rs:266-287): one wrench per step() cannot
    carry that, so the combination is refused instead of answered wrongly."""
    from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld
    from salva_amd.coupling import ColliderCouplingSet, RigidBody, StaticSampling

    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    w.add_fluid(Fluid(scenes.cube_fluid_positions(6, 6, 6, R) + np.float32([0, 8 * R, 0]), R, 1000.0))
    b = w.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
    c = ColliderCouplingSet()
    c.register_coupling(b, "raft", RigidBody(translation=np.float32([0, 0, 0]), mass=1.0, principal_inertia=np.float32([1, 1, 1])),
                        StaticSampling(scenes.plane_lattice(8, 8, 0.0, R, -8 * R, -8 * R, layers=1)))
    w.set_cfl_substepping(1)
    with pytest.raises(_lib.SalvaHipError, match="substep"):
        w.step_with_coupling(DT, GRAVITY, c)
    w.set_cfl_substepping(0)
    w.step_with_coupling(DT, GRAVITY, c)
