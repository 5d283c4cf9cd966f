"""Rigid-body coupling on the device against the oracle's restatement of the StaticSampling arm of
src/integrations/rapier/fluids_pipeline.rs: a dynamic raft under a falling block, a kinematic spinning paddle, and a
parentless static collider, all driven by the same host-side rigid bodies on both sides."""
import copy
import ctypes as C

import numpy as np
import pytest

from parity import DT, GRAVITY
from oracle import oracle as O
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, _lib, scenes
from salva_amd.coupling import ColliderCouplingSet, RigidBody, StaticSampling

pytestmark = pytest.mark.gpu

R = 0.025


def _scene():
    pos = scenes.jitter(scenes.cube_fluid_positions(8, 8, 8, R), 0.05 * R, seed=11)
    pos[:, 1] += np.float32(8 * R + 3 * R)
    vel = scenes.random_velocities(len(pos), 0.05, seed=12)
    raft_pts = scenes.plane_lattice(12, 12, 0.0, R, -6 * 2 * R + R, -6 * 2 * R + R, layers=2)
    paddle_pts = scenes.plane_lattice(3, 8, 0.0, R, -3 * R + R, -8 * R + R, layers=1)[:, [0, 2, 1]]  # a vertical blade
    wall_pts = scenes.plane_lattice(2, 12, 0.0, R, 0.0, -6 * 2 * R + R, layers=1)[:, [1, 0, 2]]
    raft = RigidBody(translation=np.float32([0.0, 0.0, 0.0]), mass=2.0, principal_inertia=np.float32([0.02, 0.04, 0.02]),
                     local_com=np.float32([0.0, -R, 0.0]))
    paddle = RigidBody(translation=np.float32([0.0, 8 * R, 0.0]), angvel=np.float32([0.0, 6.0, 0.0]), dynamic=False)
    return pos, vel, raft_pts, paddle_pts, wall_pts, raft, paddle


def _run_oracle(nsteps):
    pos, vel, raft_pts, paddle_pts, wall_pts, raft, paddle = _scene()
    w = O.OracleWorld(R, 2.0, O.DFSPH)
    f = w.add_fluid(pos, 1000.0, vel)
    w.add_xsph(f, 0.5, 0.5)
    empty = np.zeros((0, 3), np.float32)
    for pts in (raft_pts, paddle_pts, wall_pts):
        b = w.add_boundary(empty)
        w.set_boundary_sampling(b, pts)
    bodies = [raft, paddle, None]
    snap = {}
    for step in range(nsteps):
        for b, body in enumerate(bodies):
            if body is None:
                w.update_boundary_pose(b, translation=(0.35, 0.0, 0.0), has_body=False)
            else:
                w.update_boundary_pose(b, body.translation, body.rotation, body.linvel, body.angvel, body.center_of_mass(),
                                       True, body.is_dynamic())
        if step == nsteps - 1:
            snap["bpos"] = [w.boundary_vec(b, "positions") for b in range(3)]
            snap["bvel"] = [w.boundary_vec(b, "velocities") for b in range(3)]
        w.step(DT, GRAVITY)
        for b, body in enumerate(bodies):
            if body is not None and body.is_dynamic():
                F, T = w.boundary_wrench(b, body.center_of_mass())
                snap["wrench"] = (F, T)
                body.apply_impulse(np.float32(F) * np.float32(DT))
                body.apply_torque_impulse(np.float32(T) * np.float32(DT))
        for body in (raft, paddle):
            body.integrate(DT, (0.0, 0.0, 0.0))  # the raft only feels the fluid
    snap["pos"], snap["vel"] = w.fluid_vec(f, "positions"), w.fluid_vec(f, "velocities")
    snap["raft"] = copy.deepcopy(raft)
    snap["bforce"] = w.boundary_vec(0, "forces")
    return snap


def _run_hip(nsteps):
    pos, vel, raft_pts, paddle_pts, wall_pts, raft, paddle = _scene()
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    fl.velocities = vel
    fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.5))
    h = w.add_fluid(fl)
    coupling = ColliderCouplingSet()
    bounds = [w.add_boundary(Boundary(np.zeros((0, 3), np.float32))) for _ in range(3)]
    static = RigidBody(translation=np.float32([0.35, 0.0, 0.0]))
    coupling.register_coupling(bounds[0], "raft", raft, StaticSampling(raft_pts))
    coupling.register_coupling(bounds[1], "paddle", paddle, StaticSampling(paddle_pts))
    coupling.register_coupling(bounds[2], "wall", None, StaticSampling(wall_pts))
    snap = {}
    for step in range(nsteps):
        w.sync_to_device()
        coupling.update_boundaries(w)
        # a parentless collider still has a pose (the set above only knows bodies): hand it over through the raw entry point
        pose = static.pose()
        pose.has_body = 0
        _lib.check(w._L.salva_hip_update_boundary_pose(w._h, bounds[2]._slot, pose))
        if step == nsteps - 1:
            snap["bpos"] = [b.positions for b in bounds]
            snap["bvel"] = [b.velocities for b in bounds]
        w.step(DT, GRAVITY)
        if step == nsteps - 1:
            com = raft.center_of_mass()
            F, T = np.zeros(3, np.float32), np.zeros(3, np.float32)
            fp = C.POINTER(C.c_float)
            _lib.check(w._L.salva_hip_get_boundary_wrench(w._h, bounds[0]._slot, com.ctypes.data_as(fp), F.ctypes.data_as(fp),
                                                          T.ctypes.data_as(fp)))
            snap["wrench"] = (F, T)
            snap["bforce"] = bounds[0].forces
        coupling.transmit_forces(w, DT)
        for body in (raft, paddle):
            body.integrate(DT, (0.0, 0.0, 0.0))
    snap["pos"], snap["vel"] = h.positions.copy(), h.velocities.copy()
    snap["raft"] = copy.deepcopy(raft)
    snap["flags"] = [b.wants_forces for b in bounds]
    return snap


def test_coupled_bodies_match_oracle():
    nsteps = 12
    ref = _run_oracle(nsteps)
    got = _run_hip(nsteps)
    assert got["flags"] == [True, False, False]
    for b in range(3):
        assert np.abs(got["bpos"][b] - ref["bpos"][b]).max() < 1e-6      # pose * point
        assert np.abs(got["bvel"][b] - ref["bvel"][b]).max() < 1e-5      # velocity_at_point(local point)
    assert np.abs(ref["bvel"][1]).max() > 0.1 and not ref["bvel"][2].any()
    assert np.abs(got["pos"] - ref["pos"]).max() < 1e-4 * R * nsteps
    vref = max(np.abs(ref["vel"]).max(), 2 * R / DT * 1e-2)
    assert np.abs(got["vel"] - ref["vel"]).max() < 1e-4 * nsteps * vref
    # the raft was pushed down (and tilted by the paddle-stirred, off-centre load)
    assert ref["raft"].linvel[1] < -1e-3
    fscale = np.abs(ref["bforce"]).max()
    assert fscale > 0 and np.abs(got["bforce"] - ref["bforce"]).max() < 2e-3 * fscale
    F, T = got["wrench"]
    Fr, Tr = ref["wrench"]
    assert np.abs(F - Fr).max() < 1e-3 * np.abs(Fr).max() and np.abs(T - Tr).max() < 1e-3 * max(np.abs(Tr).max(), 1e-3 * np.abs(Fr).max())
    assert np.abs(got["raft"].linvel - ref["raft"].linvel).max() < 1e-3 * np.abs(ref["raft"].linvel).max()
    assert np.abs(got["raft"].angvel - ref["raft"].angvel).max() < 1e-3 * max(np.abs(ref["raft"].angvel).max(), 1e-3)


def test_pose_without_sampling_is_rejected():
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    b = w.add_boundary(Boundary(scenes.plane_lattice(4, 4, 0.0, R, 0.0, 0.0, layers=1)))
    w.sync_to_device()
    pose = RigidBody().pose()
    with pytest.raises(_lib.SalvaHipError) as e:
        _lib.check(w._L.salva_hip_update_boundary_pose(w._h, b._slot, pose))
    assert e.value.code == _lib.E_INVALID


def test_cpp_mirror_coupling_example():
    """examples/coupling3.cpp: a half-density box dropped on a pool through include/salva_hip.hpp's ColliderCouplingSet
    (pose in, wrench out).  It must fall, hit the water, and end up floating: decelerated to a small velocity with its
    lowest face below the surface and the box centre above the pool floor."""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "coupling3")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples")])
    out = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = out.stdout.strip().splitlines()
    rows = [re.match(r"step (\d+): box y (-?[\d.]+) vy (-?[\d.]+) \|angvel\| ([\d.]+), lowest sample y (-?[\d.]+), (\d+) samples", ln) for ln in lines]
    assert all(rows), lines
    y = [float(m[2]) for m in rows]
    vy = [float(m[3]) for m in rows]
    assert int(rows[-1][6]) == 56
    surface = 9 * 0.05                      # 8 layers over the floor layer
    assert y[0] < (8 + 5) * 0.05            # it fell ...
    assert min(vy) < -0.2                   # ... picked up speed ...
    assert abs(vy[-1]) < 0.25               # ... and was stopped by the water, not by the floor
    assert 0.15 < y[-1] < surface + 0.1 and float(rows[-1][5]) < surface


def test_unregistered_moving_collider_keeps_its_velocities():
    """`unregister_coupling` (fluids_pipeline.rs:116-125) leaves the boundary in the world with the particles — and the
    VELOCITIES — the last pose gave it.  The library used to keep the upload's "at rest" mark for such a boundary and then skipped its
    velocities in the predicted-density pass (StepCtx::bvel_zero).  A world whose paddle was detached must step exactly like a world
    that was handed the same particles and velocities as a plain boundary."""
    pos, vel, _, paddle_pts, _, _, paddle = _scene()
    pos = pos.copy(); pos[:, 1] -= np.float32(3 * R)  # the block sits on the blade: every step has fluid-boundary contacts

    def world():
        w = LiquidWorld(DFSPHSolver(), R, 2.0)
        fl = Fluid(pos, R, 1000.0)
        fl.velocities = vel
        fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.5))
        return w, w.add_fluid(fl)

    w1, f1 = world()
    b1 = w1.add_boundary(Boundary(np.zeros((0, 3), np.float32)))
    coupling = ColliderCouplingSet()
    coupling.register_coupling(b1, "paddle", copy.deepcopy(paddle), StaticSampling(paddle_pts))
    w1.sync_to_device()
    coupling.update_boundaries(w1)
    assert coupling.unregister_coupling("paddle") is b1
    bpos, bvel = b1.positions.copy(), b1.velocities.copy()
    assert np.abs(bvel).max() > 0.1  # the blade spins

    w2, f2 = world()
    plain = Boundary(bpos)
    plain.velocities = bvel
    w2.add_boundary(plain)
    for _ in range(6):
        s1 = w1.step(DT, GRAVITY)
        s2 = w2.step(DT, GRAVITY)
        assert (s1.ncontacts, s1.n_divergence_iters, s1.n_pressure_iters) == (s2.ncontacts, s2.n_divergence_iters, s2.n_pressure_iters)
    assert np.array_equal(f1.positions, f2.positions) and np.array_equal(f1.velocities, f2.velocities)
    # ... and the velocities matter in this scene: the same blade at rest gives another trajectory
    w3, f3 = world()
    w3.add_boundary(Boundary(bpos))
    for _ in range(6):
        w3.step(DT, GRAVITY)
    assert np.abs(f3.velocities - f2.velocities).max() > 1e-4
