"""ColliderSampling::DynamicContactSampling on the device (integrations/rapier/fluids_pipeline.rs:193-259)
against the oracle's restatement: a tilted, parentless cuboid and a ball attached to a moving dynamic body, both overlapped
by the fluid from the first step so that the push-out branch runs at once.

Step 0 starts from identical state on both sides, so everything the arm produces is compared bit for bit: which particles
emitted a boundary particle, the projected points and their velocities, the pushed-out fluid positions / velocities, and the
contact set found from the stale cells.  The following steps (solver rounding differs) are compared like every other scene."""
import ctypes as C

import numpy as np
import pytest

from parity import DT, GRAVITY, max_norm_diff
from oracle import oracle as O
from salva_amd import Boundary, DFSPHSolver, Fluid, IISPHSolver, LiquidWorld, NonPressureForce, XSPHViscosity, _lib, scenes
from salva_amd.coupling import ColliderCouplingSet, DynamicContactSampling, RigidBody

pytestmark = pytest.mark.gpu

R = 0.025
CUBOID_HE = (0.30, 0.04, 0.22)
BALL_R = 0.11
# (oracle kind, oracle parameters, mirror shape) of the fixed collider under the block and of the body that enters it
SHAPES = {
    "cuboid+ball": ((2, CUBOID_HE, ("cuboid", CUBOID_HE)), (1, [BALL_R], ("ball", BALL_R))),
    # a thin disc (cylinder along its local y) as the floor and a capsule as the body
    "cylinder+capsule": ((4, [0.04, 0.34], ("cylinder", 0.04, 0.34)), (3, [0.07, 0.08], ("capsule", 0.07, 0.08))),
}


def _scene(n=14):
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.2 * R, seed=5)
    pos[:, 1] += np.float32(n * R + 0.02)  # the block's lowest layer starts inside the slab's top face
    vel = scenes.random_velocities(len(pos), 0.6, seed=6)
    slab = RigidBody(translation=np.float32([0.03, -0.03, -0.02]), rotation=scenes.quat_from_scaled_axis((0.06, 0.02, 0.2)))
    ball = RigidBody(translation=np.float32([0.08, 0.30, 0.05]), linvel=np.float32([0.3, 0.9, -0.2]), angvel=np.float32([1.0, -2.0, 0.5]),
                     local_com=np.float32([0.01, 0.0, -0.02]), mass=0.8, principal_inertia=np.float32([0.004, 0.004, 0.004]))
    return pos, vel, slab, ball


BALL_V0 = np.float32([1.0, 0.1, 0.0])


def _calm_scene(n=14):
    """The block starts just above a slightly tilted slab and settles on it; a heavy ball drifts into its side."""
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=7)
    pos[:, 1] += np.float32(n * R + 0.075)
    vel = scenes.random_velocities(len(pos), 0.1, seed=8)
    slab = RigidBody(translation=np.float32([0.0, -0.03, 0.0]), rotation=scenes.quat_from_scaled_axis((0.02, 0.0, 0.04)))
    ball = RigidBody(translation=np.float32([-0.52, 0.30, 0.02]), linvel=BALL_V0.copy(), angvel=np.float32([0.0, 0.0, -3.0]),
                     local_com=np.float32([0.0, 0.01, 0.0]), mass=20.0, principal_inertia=np.float32([0.1, 0.1, 0.1]))
    return pos, vel, slab, ball


class Probe(NonPressureForce):
    """A user force that adds nothing: it records the positions the solver works on, i.e. the state right after
    update_boundaries pushed particles out of the colliders (positions are integrated at the very end of the step)."""

    def __init__(self):
        self.positions = None

    def solve(self, timestep, kernel_radius, ff, fb, fluid, boundaries, densities):
        self.positions = fluid.positions.copy()


def _oracle_world(solver, pos, vel, f64=False, probe=None, shapes="cuboid+ball"):
    w = O.OracleWorld(R, 2.0, O.DFSPH if solver == "dfsph" else O.IISPH, f64=f64)
    f = w.add_fluid(pos, 1000.0, vel)
    w.add_xsph(f, 0.5, 0.5)
    if probe is not None:
        w.add_custom_force(f, lambda world, fl, positions, velocities, densities, acc: probe.append(positions.astype(np.float32)))
    empty = np.zeros((0, 3), np.float32)
    b0, b1 = w.add_boundary(empty), w.add_boundary(empty)
    w.set_boundary_dynamic_sampling(b0, SHAPES[shapes][0][0], SHAPES[shapes][0][1])
    w.set_boundary_dynamic_sampling(b1, SHAPES[shapes][1][0], SHAPES[shapes][1][1])
    return w, f


def _oracle_pose(w, slab, ball):
    w.update_boundary_pose(0, slab.translation, slab.rotation, has_body=False)
    w.update_boundary_pose(1, ball.translation, ball.rotation, ball.linvel, ball.angvel, ball.center_of_mass(), True, True)


def _hip_world(solver, pos, vel, slab, ball, probe=None, shapes="cuboid+ball"):
    w = LiquidWorld(DFSPHSolver() if solver == "dfsph" else IISPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    fl.velocities = vel
    fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.5))
    if probe is not None:
        fl.nonpressure_forces.append(probe)
    h = w.add_fluid(fl)
    bounds = [w.add_boundary(Boundary(np.zeros((0, 3), np.float32))) for _ in range(2)]
    coupling = ColliderCouplingSet()
    coupling.register_coupling(bounds[0], "slab", None, DynamicContactSampling(SHAPES[shapes][0][2]))
    coupling.register_coupling(bounds[1], "ball", ball, DynamicContactSampling(SHAPES[shapes][1][2]))
    return w, h, bounds, coupling


def _hip_pose(w, coupling, bounds, slab):
    w.sync_to_device()
    coupling.update_boundaries(w)
    pose = slab.pose()  # the parentless collider's pose goes through the raw entry point (the set only knows bodies)
    pose.has_body = 0
    _lib.check(w._L.salva_hip_update_boundary_pose(w._h, bounds[0]._slot, pose))


def _by_source(fluid_ids, particle_ids, *arrays):
    order = np.lexsort((particle_ids, fluid_ids))
    return [particle_ids[order]] + [a[order] for a in arrays]


@pytest.mark.parametrize("solver,shapes", [("dfsph", "cuboid+ball"), ("iisph", "cuboid+ball"), ("dfsph", "cylinder+capsule")])
def test_first_step_is_bit_exact(solver, shapes):
    pos, vel, slab, ball = _scene()
    oprobe, gprobe = [], Probe()
    o, f = _oracle_world(solver, pos, vel, probe=oprobe, shapes=shapes)
    w, h, bounds, coupling = _hip_world(solver, pos, vel, slab, ball, probe=gprobe, shapes=shapes)
    # a previous substep length, as a continued run would carry it: the prediction x + v dt is exercised from the first step
    o.set_timestep(DT, 1.0 / DT)
    w.sync_to_device()
    _lib.check(w._L.salva_hip_set_timestep(w._h, DT, 1.0 / DT))
    _oracle_pose(o, slab, ball)
    _hip_pose(w, coupling, bounds, slab)
    so = o.step(DT, GRAVITY)
    st = w.step(DT, GRAVITY)
    for b in range(2):
        n = o.boundary_len(b)
        assert n > 50 and bounds[b].num_particles() == n, f"boundary {b}: {bounds[b].num_particles()} points vs {n}"
        of, op = o.boundary_sources(b)
        idx_o, po, vo = _by_source(of, op, o.boundary_vec(b, "positions").astype(np.float32), o.boundary_vec(b, "velocities").astype(np.float32))
        gf, gp = bounds[b].sources()
        idx_g, pg, vg = _by_source(gf, gp, bounds[b].positions, bounds[b].velocities)
        assert np.array_equal(idx_o, idx_g), f"boundary {b}: different fluid particles were sampled"
        assert np.array_equal(po, pg), f"boundary {b}: projections differ by {np.abs(po - pg).max():.3e}"
        assert np.array_equal(vo, vg), f"boundary {b}: velocity_at_point differs by {np.abs(vo - vg).max():.3e}"
        if b == 0:
            assert not vo.any()
        else:
            assert np.abs(vo).max() > 0.5
    # the pushed-out positions, as the solver saw them
    assert len(oprobe) == 1 and gprobe.positions is not None
    moved = np.abs(oprobe[0] - pos).max(axis=1) > 0
    assert moved.sum() > 20, "the scene did not exercise the push-out branch"
    assert np.array_equal(oprobe[0], gprobe.positions), f"pushed positions differ by {np.abs(oprobe[0] - gprobe.positions).max():.3e}"
    # the contact search ran from the stale cells: identical contact sets
    assert int(st.ncontacts) == int(so.ncontacts)
    assert np.array_equal(w.contact_counts(h), o.contact_counts(f))
    assert np.array_equal(w.contact_counts(h, True), o.contact_counts(f, True))
    d = max_norm_diff(h.positions, o.fluid_vec(f, "positions")) / R
    vref = max(float(np.abs(o.fluid_vec(f, "velocities")).max()), 2 * R / DT * 1e-2)
    dv = max_norm_diff(h.velocities, o.fluid_vec(f, "velocities")) / vref
    assert d < 1e-4 and dv < 1e-4, f"after the first step positions differ by {d:.2e} r, velocities by {dv:.2e} v_ref"


@pytest.mark.parametrize("solver", ["dfsph", "iisph"])
def test_thirty_steps_with_a_moving_ball(solver):
    """The block settles on the slab while the ball (a dynamic body, slowed down by the fluid it displaces) enters its side."""
    nsteps = 30
    pos, vel, slab, ball_o = _calm_scene()
    _, _, _, ball_g = _calm_scene()
    o, f = _oracle_world(solver, pos, vel)
    o64, f64 = _oracle_world(solver, pos, vel, f64=True)
    _, _, _, ball_64 = _calm_scene()
    w, h, bounds, coupling = _hip_world(solver, pos, vel, slab, ball_g)
    counts = []
    for k in range(nsteps):
        _oracle_pose(o, slab, ball_o)
        _oracle_pose(o64, slab, ball_64)
        _hip_pose(w, coupling, bounds, slab)
        so = o.step(DT, GRAVITY)
        o64.step(DT, GRAVITY)
        st = w.step(DT, GRAVITY)
        for ow, body in ((o, ball_o), (o64, ball_64)):
            F, T = ow.boundary_wrench(1, body.center_of_mass())
            body.apply_impulse(np.float32(F) * np.float32(DT))
            body.apply_torque_impulse(np.float32(T) * np.float32(DT))
            body.integrate(DT, (0.0, 0.0, 0.0))
        coupling.transmit_forces(w, DT)
        ball_g.integrate(DT, (0.0, 0.0, 0.0))
        n_o = [o.boundary_len(b) for b in range(2)]
        n_g = [bounds[b].num_particles() for b in range(2)]
        counts.append((n_g, n_o))
        for b in range(2):
            assert abs(n_g[b] - n_o[b]) <= max(3, n_o[b] // 50), f"step {k}: boundary {b} emitted {n_g[b]} points, oracle {n_o[b]}"
        slack = 0 if k == 0 else max(8, int(2e-4 * so.ncontacts))
        assert abs(int(st.ncontacts) - int(so.ncontacts)) <= slack, f"step {k}: contacts {st.ncontacts} vs {so.ncontacts}"
    po, vo = o.fluid_vec(f, "positions"), o.fluid_vec(f, "velocities")
    vref = max(float(np.abs(vo).max()), 2 * R / DT * 1e-2)
    tol_p = max(1e-4 * nsteps, 2 * max_norm_diff(po, o64.fluid_vec(f64, "positions")) / R)
    tol_v = max(1e-4 * nsteps, 2 * max_norm_diff(vo, o64.fluid_vec(f64, "velocities")) / vref)
    d = max_norm_diff(h.positions, po) / R
    dv = max_norm_diff(h.velocities, vo) / vref
    assert d < tol_p, f"positions differ by {d:.2e} r (tolerance {tol_p:.2e})"
    assert dv < tol_v, f"velocities differ by {dv:.2e} v_ref (tolerance {tol_v:.2e})"
    # the ball felt the fluid: its velocity changed, the same way on both sides
    assert np.abs(ball_o.linvel - BALL_V0).max() > 1e-3
    scale = max(np.abs(ball_o.linvel - BALL_V0).max(), 1e-3)
    tol_b = max(1e-3, 2 * np.abs(ball_o.linvel - ball_64.linvel).max() / scale)
    assert np.abs(ball_g.linvel - ball_o.linvel).max() / scale < tol_b
    print("emitted (gpu, oracle) per step:", counts)


def test_rejected_uses():
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    w.add_fluid(Fluid(scenes.cube_fluid_positions(4, 4, 4, R), R, 1000.0))
    b = w.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
    w.sync_to_device()
    bad = _lib.Shape()
    bad.kind = 7
    with pytest.raises(_lib.SalvaHipError) as e:
        _lib.check(w._L.salva_hip_set_boundary_dynamic_sampling(w._h, b._slot, C.byref(bad), 1, 0xFFFFFFFF))
    assert e.value.code == _lib.E_INVALID
    bad.kind, bad.params[0] = _lib.SHAPE_BALL, -1.0
    with pytest.raises(_lib.SalvaHipError):
        _lib.check(w._L.salva_hip_set_boundary_dynamic_sampling(w._h, b._slot, C.byref(bad), 1, 0xFFFFFFFF))
    # a boundary with uploaded particles is not dynamically sampled: its sources cannot be asked for
    u32p = C.POINTER(C.c_uint32)
    with pytest.raises(_lib.SalvaHipError):
        _lib.check(w._L.salva_hip_get_boundary_sources(w._h, b._slot, C.cast(None, u32p), C.cast(None, u32p)))


def test_cpp_mirror_dynamic_coupling_example():
    """examples/dynamic_coupling3.cpp: a half-density ball dropped on a pool through include/salva_hip.hpp's
    Boundary::dynamic_ball + ColliderCouplingSet.  It must fall, meet the water (boundary particles appear), be slowed down and
    stay above the pool floor."""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "dynamic_coupling3")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples")])
    out = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = out.stdout.strip().splitlines()
    rows = [re.match(r"step (\d+): ball y (-?[\d.]+) vy (-?[\d.]+) \|angvel\| ([\d.]+), lowest sample y (-?[\d.]+), (\d+) samples", ln) for ln in lines]
    assert all(rows), lines
    y = [float(m.group(2)) for m in rows]
    vy = [float(m.group(3)) for m in rows]
    samples = [int(m.group(6)) for m in rows]
    assert y[0] > y[-1], lines                       # it fell
    assert max(samples) > 20, lines                  # the fluid was projected onto it
    assert abs(vy[-1]) < 1.0 and y[-1] > 0.1, lines  # free fall for 2 s would be 19.6 m/s and far below the floor


def test_counters_report_the_boundary_update():
    pos, vel, slab, ball = _calm_scene(10)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    h = w.add_fluid(fl)
    b = w.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
    coupling = ColliderCouplingSet()
    coupling.register_coupling(b, "slab", slab, DynamicContactSampling(("cuboid", CUBOID_HE)))
    slab.dynamic = False
    w.counters.enable()
    for _ in range(3):
        w.step_with_coupling(DT, GRAVITY, coupling)
    c = w.counters
    assert c.cd.boundary_update_time > 0.0 and c.cd.grid_insertion_time >= 0.0
    assert c.stages.collision_detection_time >= c.cd.boundary_update_time
    assert b.num_particles() > 0 and h.num_particles() == len(pos)


def test_fluids_pipeline_mirror_steps_like_the_manual_loop():
    """`FluidsPipeline` (fluids_pipeline.rs:18-61) = LiquidWorld(DFSPH) + coupling set; its step is step_with_coupling."""
    from salva_amd.coupling import FluidsPipeline

    pos, vel, slab, ball = _calm_scene(10)
    _, _, _, ball2 = _calm_scene(10)
    slab.dynamic = False

    def build(world, coupling, the_ball):
        fl = Fluid(pos, R, 1000.0)
        fl.velocities = vel
        h = world.add_fluid(fl)
        b0 = world.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
        b1 = world.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
        coupling.register_coupling(b0, "slab", slab, DynamicContactSampling(("cuboid", CUBOID_HE)))
        coupling.register_coupling(b1, "ball", the_ball, DynamicContactSampling(("ball", BALL_R)))
        return h

    pipe = FluidsPipeline(R, 2.0)
    hp = build(pipe.liquid_world, pipe.coupling, ball)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    c = ColliderCouplingSet()
    hw = build(w, c, ball2)
    for _ in range(5):
        pipe.step(GRAVITY, DT)
        w.step_with_coupling(DT, GRAVITY, c)
        for body in (ball, ball2):
            body.integrate(DT, (0.0, 0.0, 0.0))
    assert np.array_equal(hp.positions, hw.positions) and np.array_equal(ball.linvel, ball2.linvel)


def test_surface_tension3_literal_scene():
    """examples3d/surface_tension3.rs — the reference's own use of DynamicContactSampling: a droplet (Akinci2013 + artificial
    viscosity, decimetre units) falls on a fixed cuboid whose boundary particles exist only where the droplet is.  Through the
    FluidsPipeline mirror, 120 steps (the droplet is within reach of the ground from the start, lands within 40 steps, spreads) against the oracle: emitted points and contacts
    per step, then the droplet's state against max(stated tolerance, the oracle's own f32-vs-f64 distance)."""
    from salva_amd import Akinci2013SurfaceTension, ArtificialViscosity
    from salva_amd.coupling import FluidsPipeline

    sc = scenes.surface_tension3()
    r, pos, g = sc["radius"], sc["fluid"], sc["gravity"]
    pipe = FluidsPipeline(r, 2.0)
    fl = Fluid(pos, r, 1000.0)
    fl.nonpressure_forces.append(Akinci2013SurfaceTension(1.0, 0.0))
    fl.nonpressure_forces.append(ArtificialViscosity(0.01, 0.01))
    h = pipe.liquid_world.add_fluid(fl)
    bo = pipe.liquid_world.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
    ground = RigidBody(dynamic=False)  # RigidBodyBuilder::fixed()
    pipe.coupling.register_coupling(bo, "ground", ground, DynamicContactSampling(("cuboid", sc["ground_half_extents"])))

    def oracle(f64):
        o = O.OracleWorld(r, 2.0, O.DFSPH, f64=f64)
        f = o.add_fluid(pos, 1000.0)
        o.add_akinci2013(f, 1.0, 0.0)
        o.add_artificial_viscosity(f, 0.01, 0.01)
        b = o.add_boundary(np.zeros((0, 3), np.float32))
        o.set_boundary_dynamic_sampling(b, 2, sc["ground_half_extents"])
        o.update_boundary_pose(b, (0, 0, 0), (0, 0, 0, 1), (0, 0, 0), (0, 0, 0), (0, 0, 0), True, False)
        return o

    o, o64 = oracle(False), oracle(True)
    nsteps, emitted = 120, []
    for k in range(nsteps):
        st = pipe.step(g, DT)
        so = o.step(DT, g)
        o64.step(DT, g)
        ng, no = bo.num_particles(), o.boundary_len(0)
        emitted.append((ng, no))
        d = max_norm_diff(h.positions, o.fluid_vec(0, "positions")) / r
        if d < 1e-2:  # while the two runs are the same flow: the same particles are within reach of the ground
            assert abs(ng - no) <= max(2, no // 50), f"step {k}: {ng} boundary particles, oracle {no}"
            assert abs(int(st.ncontacts) - int(so.ncontacts)) <= (0 if k == 0 else max(8, int(2e-4 * so.ncontacts))), (k, st.ncontacts, so.ncontacts)
    assert max(e[1] for e in emitted) > 100 and emitted[9][1] > 0, emitted  # a wetted patch that grows as the droplet lands
    # the flow is regular for about 20 steps (the oracle's own f32 and f64 runs are 3e-3 r apart there, 0.2 r after 30 steps and
    # decorrelated after 40: Akinci's normalised cohesion terms); the per-step checks above covered that stretch
    assert sum(1 for k in range(20) if emitted[k][0] == emitted[k][1]) >= 18, emitted[:20]
    po = o.fluid_vec(0, "positions")
    noise = max_norm_diff(po, o64.fluid_vec(0, "positions")) / r
    d = max_norm_diff(h.positions, po) / r
    assert d < max(1e-4 * nsteps, 10.0 * noise), f"after {nsteps} steps: {d:.3e} r vs oracle, oracle f32-f64 {noise:.3e} r"
    # bulk state once the runs have decorrelated: the droplet rests on the ground (top face at y = 0.02) in both, nothing fell
    # through, centre of mass and spread agree to a particle radius
    pg = h.positions
    assert pg[:, 1].min() > 0.02 and po[:, 1].min() > 0.02
    assert np.abs(pg.mean(axis=0) - po.mean(axis=0)).max() < 2 * r
    assert abs(pg[:, 1].max() - po[:, 1].max()) < 4 * r
    assert not bo.wants_forces  # fixed body: boundary.forces = None (fluids_pipeline.rs:163-165)


def test_dynamic_sampling_on_a_folded_grid(monkeypatch):
    """Round 6: worlds with dynamically sampled colliders fold their grid too (before, a stray particle blew their dense cell table
    up without bound: VERDICT r05, missing 3).  The pass reads each particle's cell back from its sort key, which on a torus names
    the cell modulo the period: dcs.hip picks the image the particle's position says.  A torus of 8 cells per axis — the block is
    7 cells wide, the slab under it wider than the period — against the unfolded grid: the same particles emit, the same points
    come out, bit for bit, step after step; and a particle 400 cells away costs nothing."""
    def run(env):
        for k in ("SALVA_HIP_NO_FOLD", "SALVA_HIP_FOLD_CELLS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pos, vel, slab, ball = _calm_scene(12)
        pos = np.concatenate([pos, np.float32([[0.1, -40.0, 0.05], [30.0, 0.4, -0.2]])])  # two strays: far below, far aside
        vel = np.concatenate([vel, np.zeros((2, 3), np.float32)])
        w, h, bounds, coupling = _hip_world("dfsph", pos, vel, slab, ball)
        out = []
        for _ in range(12):
            _hip_pose(w, coupling, bounds, slab)
            st = w.step(DT, GRAVITY)
            coupling.transmit_forces(w, DT)
            ball.integrate(DT, (0.0, 0.0, 0.0))
            out.append((int(st.ncontacts), st.n_divergence_iters, st.n_pressure_iters,
                        [np.array(b.positions, dtype=np.float32).copy() for b in bounds], [b.sources() for b in bounds]))
        return out, np.array(h.positions, dtype=np.float32).copy(), ball.linvel.copy()

    ref, pref, vref = run({"SALVA_HIP_NO_FOLD": "1"})
    got, pgot, vgot = run({"SALVA_HIP_FOLD_CELLS": "8"})
    assert sum(len(p) for p in ref[-1][3]) > 50  # both colliders are sampled
    for k, (a, b) in enumerate(zip(ref, got)):
        assert a[:3] == b[:3], (k, a[:3], b[:3])
        for pa, pb in zip(a[3], b[3]):
            assert np.array_equal(pa, pb), k
        for sa, sb in zip(a[4], b[4]):
            assert [np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(sa, sb)] == [True] * len(sa), k
    assert np.abs(pgot - pref).max() <= 1e-6 and np.abs(vgot - vref).max() <= 1e-5
