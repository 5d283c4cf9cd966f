"""ColliderSampling::DynamicContactSampling for colliders whose geometry stays with the host
(salva_hip_set_boundary_dynamic_sampling_host; integrations/rapier/fluids_pipeline.rs:193-259 is shape-generic: it only calls
`compute_aabb` and `project_point_and_get_feature`).

1. The host arm against the device arm: the same tilted cuboid and moving ball, once as built-in shapes and once as host
   shapes whose callbacks restate the two parry calls in numpy f32, operation by operation as dcs.hip does.  Both runs
   must agree bit for bit — same sampled particles, same projections, same fluid state — for several steps.
2. A shape the library has no code for (a torus): the fluid comes to rest on it, no particle ends up inside it, every
   emitted boundary particle lies on its surface.
"""
import numpy as np
import pytest

from parity import DT, GRAVITY
from salva_amd import Boundary, DFSPHSolver, Fluid, LiquidWorld, XSPHViscosity, _lib, scenes
from salva_amd.coupling import ColliderCouplingSet, DynamicContactSampling, HostShapeSampling, RigidBody

pytestmark = pytest.mark.gpu

F = np.float32
R = 0.025
CUBOID_HE = (0.30, 0.04, 0.22)
BALL_R = 0.11


def quat_rot(q, v):
    """dcs.hip quat_rot on rows of v: t = 2 q.vec x v; v' = (t w + q.vec x t) + v, every operation rounded to f32."""
    qx, qy, qz, qw = (F(x) for x in q)
    vx, vy, vz = v[:, 0], v[:, 1], v[:, 2]
    tx, ty, tz = (qy * vz - qz * vy) * F(2), (qz * vx - qx * vz) * F(2), (qx * vy - qy * vx) * F(2)
    cx, cy, cz = qy * tz - qz * ty, qz * tx - qx * tz, qx * ty - qy * tx
    return np.stack([(tx * qw + cx) + vx, (ty * qw + cy) + vy, (tz * qw + cz) + vz], axis=1).astype(F)


def to_local(body, pts):
    t = body.translation.astype(F)
    q = body.rotation.astype(F)
    return quat_rot((-q[0], -q[1], -q[2], q[3]), (pts - t).astype(F))


def to_world(body, loc):
    return (quat_rot(body.rotation.astype(F), loc) + body.translation.astype(F)).astype(F)


def ball_callbacks(body, r):
    r = F(r)

    def aabb():
        t = body.translation.astype(F)
        return t - r, t + r

    def project(pts):
        l = to_local(body, pts)
        d2 = (l[:, 0] * l[:, 0] + l[:, 1] * l[:, 1]) + l[:, 2] * l[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            f = r / np.sqrt(d2)
        return to_world(body, (l * f[:, None]).astype(F)), d2 <= r * r

    return aabb, project


def cuboid_callbacks(body, he):
    he = np.asarray(he, F)

    def aabb():
        i, j, k, w = (F(x) for x in body.rotation)
        ww, ii, jj, kk = w * w, i * i, j * j, k * k
        ij, wk, wj, ik, jk, wi = i * j * F(2), w * k * F(2), w * j * F(2), i * k * F(2), j * k * F(2), w * i * F(2)
        m = np.array([[ww + ii - jj - kk, ij - wk, wj + ik], [wk + ij, ww - ii + jj - kk, jk - wi], [ik - wj, wi + jk, ww - ii - jj + kk]], F)
        a = np.abs(m)
        ext = ((a[:, 0] * he[0] + a[:, 1] * he[1]) + a[:, 2] * he[2]).astype(F)
        t = body.translation.astype(F)
        return t - ext, t + ext

    def project(pts):
        l = to_local(body, pts)
        mins_pt, pt_maxs = (-he - l).astype(F), (l - he).astype(F)
        shift = (np.maximum(mins_pt, F(0)) - np.maximum(pt_maxs, F(0))).astype(F)
        inside = np.all(shift == 0, axis=1)
        # inside: the nearest face — the largest of mins - p, p - maxs over the axes, first axis wins ties (parry's loop order)
        for row in np.nonzero(inside)[0]:
            best, best_id, is_mins = -np.inf, 0, False
            for a in range(3):
                if mins_pt[row, a] < pt_maxs[row, a]:
                    if pt_maxs[row, a] > best:
                        best_id, is_mins, best = a, False, pt_maxs[row, a]
                elif mins_pt[row, a] > best:
                    best_id, is_mins, best = a, True, mins_pt[row, a]
            shift[row] = 0
            shift[row, best_id] = best if is_mins else -best
        return to_world(body, (l + shift).astype(F)), inside

    return aabb, project


def _scene(n=14):
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.2 * R, seed=5)
    pos[:, 1] += np.float32(n * R + 0.02)  # the block's lowest layer starts inside the slab's top face
    vel = scenes.random_velocities(len(pos), 0.6, seed=6)
    slab = RigidBody(translation=F([0.03, -0.03, -0.02]), rotation=scenes.quat_from_scaled_axis((0.06, 0.02, 0.2)), dynamic=False)
    ball = RigidBody(translation=F([0.08, 0.30, 0.05]), linvel=F([0.3, 0.9, -0.2]), angvel=F([1.0, -2.0, 0.5]),
                     local_com=F([0.01, 0.0, -0.02]), mass=0.8, principal_inertia=F([0.004, 0.004, 0.004]), dynamic=False)
    # (both bodies are kinematic: the wrench on a dynamic body is an atomically accumulated sum whose last bit varies from
    # run to run, after which two runs of the SAME arm stop being bit-identical as well)
    return pos, vel, slab, ball


def _world(pos, vel, slab, ball, host):
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    fl.velocities = vel
    fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.5))
    h = w.add_fluid(fl)
    bounds = [w.add_boundary(Boundary(np.zeros((0, 3), F))) for _ in range(2)]
    c = ColliderCouplingSet()
    if host:
        c.register_coupling(bounds[0], "slab", slab, HostShapeSampling(*cuboid_callbacks(slab, CUBOID_HE)))
        c.register_coupling(bounds[1], "ball", ball, HostShapeSampling(*ball_callbacks(ball, BALL_R)))
    else:
        c.register_coupling(bounds[0], "slab", slab, DynamicContactSampling(("cuboid", CUBOID_HE)))
        c.register_coupling(bounds[1], "ball", ball, DynamicContactSampling(("ball", BALL_R)))
    return w, h, bounds, c


def test_host_shapes_match_the_device_arm_bit_for_bit():
    pos, vel, slab_a, ball_a = _scene()
    _, _, slab_b, ball_b = _scene()
    wa, ha, ba, ca = _world(pos, vel, slab_a, ball_a, host=False)
    wb, hb, bb, cb = _world(pos, vel, slab_b, ball_b, host=True)
    for w in (wa, wb):  # a previous substep length, so that the prediction x + v dt is exercised from the first step
        w.sync_to_device()
        _lib.check(w._L.salva_hip_set_timestep(w._h, DT, 1.0 / DT))
    for step in range(6):
        wa.step_with_coupling(DT, GRAVITY, ca)
        wb.step_with_coupling(DT, GRAVITY, cb)
        for body in (ball_a, ball_b):
            body.integrate(DT, (0.0, 0.0, 0.0))
        for k in range(2):
            na, nb = ba[k].num_particles(), bb[k].num_particles()
            assert na == nb and (step > 0 or na > 50), f"step {step} boundary {k}: {nb} points from the host arm vs {na}"
            fa, pa = ba[k].sources()
            fb_, pb = bb[k].sources()
            assert np.array_equal(pa, pb), f"step {step} boundary {k}: different fluid particles were sampled"
            assert np.array_equal(ba[k].positions, bb[k].positions), f"step {step} boundary {k}: projections differ"
            assert np.array_equal(ba[k].velocities, bb[k].velocities), f"step {step} boundary {k}: velocities differ"
        assert np.array_equal(ha.positions, hb.positions), f"step {step}: fluid positions differ"
        assert np.array_equal(ha.velocities, hb.velocities), f"step {step}: fluid velocities differ"


TORUS_R, TORUS_r = 0.22, 0.07  # about the world y axis, centred at the origin


def torus_callbacks():
    def aabb():
        e = F([TORUS_R + TORUS_r, TORUS_r, TORUS_R + TORUS_r])
        return -e, e

    def project(pts):
        p = pts.astype(np.float64)
        planar = np.hypot(p[:, 0], p[:, 2])
        d2 = np.where(planar[:, None] > 1e-12, p[:, [0, 2]] / np.maximum(planar, 1e-300)[:, None], np.array([[1.0, 0.0]]))
        ring = np.stack([d2[:, 0] * TORUS_R, np.zeros(len(p)), d2[:, 1] * TORUS_R], axis=1)
        d = p - ring
        dist = np.linalg.norm(d, axis=1)
        n = np.where(dist[:, None] > 1e-12, d / np.maximum(dist, 1e-300)[:, None], np.array([[0.0, 1.0, 0.0]]))
        return (ring + n * TORUS_r).astype(F), dist <= TORUS_r

    return aabb, project


def torus_distance(p):
    p = p.astype(np.float64)
    return np.hypot(np.hypot(p[:, 0], p[:, 2]) - TORUS_R, p[:, 1]) - TORUS_r


def test_a_shape_without_device_code_holds_the_fluid():
    n = 12
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=3)
    pos[:, 1] += F(TORUS_r + n * R + 0.01)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.5))
    h = w.add_fluid(fl)
    b = w.add_boundary(Boundary(np.zeros((0, 3), F)))
    c = ColliderCouplingSet()
    c.register_coupling(b, "torus", None, HostShapeSampling(*torus_callbacks()))
    most = 0
    for step in range(120):
        w.step_with_coupling(DT, GRAVITY, c)
        most = max(most, b.num_particles())
        if b.num_particles():
            on = np.abs(torus_distance(b.positions))
            assert on.max() < 1e-5, f"step {step}: a boundary particle is {on.max():.2e} off the torus"
        # a particle whose prediction lies inside is pushed to 0.1 r outside before the solve; none may get deeper than a step's travel
        assert torus_distance(h.positions).min() > -1.0 * R, f"step {step}: a particle is {-torus_distance(h.positions).min() / R:.2f} r inside the torus"
    assert most > 100
    # the part of the block above the hole (and outside the ring) has fallen past it
    assert (h.positions[:, 1] < -0.3).sum() > 100


def test_host_shape_errors():
    import ctypes as C

    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    w.add_fluid(Fluid(scenes.cube_fluid_positions(4, 4, 4, R), R, 1000.0))
    b = w.add_boundary(Boundary(np.zeros((0, 3), F)))
    w.sync_to_device()
    bad = _lib.HostShape()  # null callbacks
    with pytest.raises(_lib.SalvaHipError):
        _lib.check(w._L.salva_hip_set_boundary_dynamic_sampling_host(w._h, b._slot, C.byref(bad), 1, 0xFFFFFFFF))
    # an aabb callback that returns NaN aborts the step with an error instead of walking an undefined cell range
    c = ColliderCouplingSet()
    c.register_coupling(b, "nan", None, HostShapeSampling(lambda: (F([np.nan] * 3), F([np.nan] * 3)), lambda p: (p, np.zeros(len(p), bool))))
    with pytest.raises(_lib.SalvaHipError):
        w.step_with_coupling(DT, GRAVITY, c)


def test_unregistering_a_host_shape_detaches_it_before_its_callbacks_die():
    """the library keeps the host shape's callbacks and calls them in every step; `unregister_coupling` /
    replacing the entry dropped the ctypes thunks while they were still registered.  Now the set detaches the boundary first
    (salva_hip_clear_boundary_sampling): it stays in the world as a plain boundary with the particles it last held, as the
    reference's does (fluids_pipeline.rs:116-125), and later steps never enter the retired callbacks."""
    import gc

    n = 10
    pos = scenes.jitter(scenes.cube_fluid_positions(n, n, n, R), 0.1 * R, seed=5)
    pos[:, 1] += F(TORUS_r + n * R + 0.01)
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    fl = Fluid(pos, R, 1000.0)
    fl.nonpressure_forces.append(XSPHViscosity(0.5, 0.5))
    h = w.add_fluid(fl)
    b = w.add_boundary(Boundary(np.zeros((0, 3), F)))
    calls = {"n": 0}
    aabb, project = torus_callbacks()

    def counted_project(p):
        calls["n"] += 1
        return project(p)

    c = ColliderCouplingSet()
    c.register_coupling(b, "torus", None, HostShapeSampling(aabb, counted_project))
    for _ in range(40):
        w.step_with_coupling(DT, GRAVITY, c)
        if b.num_particles() > 20:
            break
    held = b.num_particles()
    assert held > 20 and calls["n"] > 0
    before_p = np.array(b.positions)
    assert c.unregister_coupling("torus") is b
    gc.collect()  # the thunks are gone now: a library that still held them would call freed memory below
    seen = calls["n"]
    for _ in range(5):
        w.step_with_coupling(DT, GRAVITY, c)
    assert calls["n"] == seen, "a retired host shape was called"
    assert b.num_particles() == held and np.array_equal(b.positions, before_p), "the detached boundary keeps its last particles"
    assert np.isfinite(h.positions).all()
    # replacing an uploaded entry detaches the old sampling the same way; the new one takes over at the next step
    c.register_coupling(b, "torus", None, HostShapeSampling(aabb, counted_project))
    w.step_with_coupling(DT, GRAVITY, c)
    old = c.register_coupling(b, "torus", None, HostShapeSampling(*torus_callbacks()))
    assert old is b
    gc.collect()
    seen = calls["n"]
    for _ in range(3):
        w.step_with_coupling(DT, GRAVITY, c)
    assert calls["n"] == seen


def test_an_exception_in_a_host_shape_callback_surfaces_from_the_step():
    """ctypes prints and swallows an exception raised inside a callback; the thunks park it and the step re-raises it."""
    w = LiquidWorld(DFSPHSolver(), R, 2.0)
    pos = scenes.cube_fluid_positions(6, 6, 6, R)
    w.add_fluid(Fluid(pos, R, 1000.0))
    b = w.add_boundary(Boundary(np.zeros((0, 3), F)))
    c = ColliderCouplingSet()

    class Boom(RuntimeError):
        pass

    def bad_project(p):
        raise Boom("projection failed")

    e = F([1.0, 1.0, 1.0])
    c.register_coupling(b, "x", None, HostShapeSampling(lambda: (-e, e), bad_project))
    with pytest.raises(Boom):
        w.step_with_coupling(DT, GRAVITY, c)

    def bad_aabb():
        raise Boom("no box")

    c.register_coupling(b, "x", None, HostShapeSampling(bad_aabb, lambda p: (p, np.zeros(len(p), bool))))
    with pytest.raises(Boom):
        w.step_with_coupling(DT, GRAVITY, c)


def test_cpp_mirror_host_shape_example():
    """examples/host_shape3.cpp: the torus through include/salva_hip.hpp's `Boundary::dynamic_host_shape`, with
    `DFSPHSolverT<Poly6Kernel, SpikyKernel>` as the solver (the C++ mirror of the two round-3 additions in one program)."""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "host_shape3"], check=True, capture_output=True)
    out = subprocess.run([os.path.join(root, "examples", "host_shape3"), "120"], check=True, capture_output=True, text=True, timeout=300).stdout
    rows = [re.search(r"step (\d+): (\d+) boundary samples \(max (\d+)\), worst sample off the surface ([0-9.e+-]+), deepest fluid particle ([0-9.e+-]+) r, (\d+) particles below", l)
            for l in out.strip().splitlines()]
    assert len(rows) == 4 and all(rows), out
    assert int(rows[-1].group(3)) > 100, out                       # the fluid was projected onto the torus
    assert all(float(m.group(4)) < 1e-5 for m in rows), out        # every sample lies on its surface
    assert all(float(m.group(5)) > -1.0 for m in rows), out        # no particle deeper than one radius inside it
    assert int(rows[-1].group(6)) > 100, out                       # the part above the hole fell through
