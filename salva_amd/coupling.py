"""Host mirror of salva's rigid-body coupling (src/integrations/rapier/fluids_pipeline.rs): the StaticSampling arm and the
DynamicContactSampling arm — on the device for ball, cuboid, capsule and cylinder colliders, with the two parry calls left to the
host for every other shape (HostShapeSampling).

rapier is not part of this project (and not available here): `RigidBody` below carries exactly the state the coupling
reads and writes — pose, velocities, centre of mass, mass properties — with rapier's formulas for the three methods the
coupling calls (`velocity_at_point`, `apply_impulse`, `apply_torque_impulse`).  A real integration passes rapier's values
through `SalvaHipRigidPose` instead.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _lib as L

F32 = np.float32


def quat_rotate(q, v):
    """nalgebra UnitQuaternion * Vector3, q = (i, j, k, w): t = 2 q.vec x v; v + w t + q.vec x t."""
    qv = np.asarray(q[:3], F32)
    v = np.asarray(v, F32)
    t = np.cross(qv, v).astype(F32) * F32(2)
    return (t * F32(q[3]) + np.cross(qv, t).astype(F32) + v).astype(F32)


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], F32)


@dataclass
class RigidBody:
    """The slice of rapier3d's RigidBody the coupling touches."""
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3, F32))
    rotation: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1], F32))  # (i, j, k, w)
    linvel: np.ndarray = field(default_factory=lambda: np.zeros(3, F32))
    angvel: np.ndarray = field(default_factory=lambda: np.zeros(3, F32))
    local_com: np.ndarray = field(default_factory=lambda: np.zeros(3, F32))
    mass: float = 1.0
    principal_inertia: np.ndarray = field(default_factory=lambda: np.ones(3, F32))  # body frame, about the centre of mass
    dynamic: bool = True

    def is_dynamic(self) -> bool:
        return self.dynamic

    def center_of_mass(self) -> np.ndarray:
        return (quat_rotate(self.rotation, self.local_com) + self.translation).astype(F32)

    def velocity_at_point(self, point) -> np.ndarray:
        """rapier: linvel + angvel x (point - world_com)."""
        return (self.linvel + np.cross(self.angvel, np.asarray(point, F32) - self.center_of_mass())).astype(F32)

    def apply_impulse(self, impulse):
        if self.dynamic:
            self.linvel = (self.linvel + np.asarray(impulse, F32) / F32(self.mass)).astype(F32)

    def apply_torque_impulse(self, torque_impulse):
        if self.dynamic:
            conj = np.array([-self.rotation[0], -self.rotation[1], -self.rotation[2], self.rotation[3]], F32)
            local = quat_rotate(conj, torque_impulse) / self.principal_inertia
            self.angvel = (self.angvel + quat_rotate(self.rotation, local)).astype(F32)

    def integrate(self, dt: float, gravity=(0.0, -9.81, 0.0)):
        """Symplectic Euler, enough for the examples and tests (collisions are the rigid-body engine's business)."""
        if self.dynamic:
            self.linvel = (self.linvel + np.asarray(gravity, F32) * F32(dt)).astype(F32)
        self.translation = (self.translation + self.linvel * F32(dt)).astype(F32)
        w = self.angvel * F32(dt)
        ang = float(np.linalg.norm(w))
        if ang > 0:
            axis = w / ang
            dq = np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]]).astype(F32)
            q = quat_mul(dq, self.rotation)
            self.rotation = (q / np.linalg.norm(q)).astype(F32)

    def pose(self) -> L.RigidPose:
        p = L.RigidPose()
        p.translation[:] = [float(x) for x in self.translation]
        p.rotation[:] = [float(x) for x in self.rotation]
        p.linvel[:] = [float(x) for x in self.linvel]
        p.angvel[:] = [float(x) for x in self.angvel]
        p.world_com[:] = [float(x) for x in self.center_of_mass()]
        p.has_body, p.is_dynamic = 1, int(self.dynamic)
        return p


class StaticSampling:
    """ColliderSampling::StaticSampling(points): collider-local sample points (fluids_pipeline.rs:36-41)."""

    def __init__(self, points):
        self.points = np.ascontiguousarray(points, F32).reshape(-1, 3)


def make_shape(shape) -> "L.Shape":
    """("ball", r) | ("cuboid", (hx, hy, hz)) | ("capsule", half_height, radius) | ("cylinder", half_height, radius) -> SalvaHipShape"""
    s = L.Shape()
    if shape[0] == "ball":
        s.kind, s.params[0] = L.SHAPE_BALL, float(shape[1])
    elif shape[0] == "cuboid":
        s.kind = L.SHAPE_CUBOID
        s.params[:] = [float(x) for x in shape[1]]
    elif shape[0] in ("capsule", "cylinder"):
        s.kind = L.SHAPE_CAPSULE if shape[0] == "capsule" else L.SHAPE_CYLINDER
        s.params[0], s.params[1] = float(shape[1]), float(shape[2])
    else:
        raise ValueError("built-in collider shapes: ('ball', radius), ('cuboid', half_extents), ('capsule', half_height, radius), "
                         "('cylinder', half_height, radius)")
    return s


class DynamicContactSampling:
    """ColliderSampling::DynamicContactSampling (fluids_pipeline.rs:42-43) for a collider of shape ("ball", radius),
    ("cuboid", (hx, hy, hz)), ("capsule", half_height, radius) (parry Capsule::new_y) or ("cylinder", half_height, radius) (axis = local
    y): the boundary's particles are the projections of the nearby fluid particles onto the collider, recomputed inside every step
    on the device (salva_hip_set_boundary_dynamic_sampling)."""

    def __init__(self, shape):
        self.shape = make_shape(shape)


class HostShapeSampling:
    """ColliderSampling::DynamicContactSampling for a collider whose shape the library has no code for (triangle mesh, height
    field, convex polyhedron, compound, ...): the loop of fluids_pipeline.rs:193-259 runs on the device, its two calls into the shape
    come back to the host once per step (salva_hip_set_boundary_dynamic_sampling_host):

      aabb()              -> (mins, maxs)               `collider.shape().compute_aabb(collider.position())`
      project(points[n,3]) -> (projections[n,3], is_inside[n])
                                                         `collider.shape().project_point_and_get_feature(collider.position(), pt)`

    both in world space, f32.  The callables see the collider through their closure (e.g. the RigidBody whose pose they apply)."""

    def __init__(self, aabb, project):
        self._aabb, self._project = aabb, project

        # An exception raised inside a ctypes callback is printed and swallowed by ctypes: the step would go on with whatever the
        # output arrays held.  The thunks therefore catch it, park it on this object and hand the library a NaN box / all-outside
        # projections (a NaN box makes the step fail with E_INVALID); LiquidWorld.step_with_coupling re-raises it after the step.
        self._error = None

        def aabb_cb(_user, mins, maxs):
            try:
                lo, hi = self._aabb()
                for a in range(3):
                    mins[a], maxs[a] = float(lo[a]), float(hi[a])
            except BaseException as e:  # noqa: BLE001
                self._error = self._error or e
                for a in range(3):
                    mins[a] = maxs[a] = float("nan")

        def project_cb(_user, n, pts, proj, inside):
            try:
                p = np.ctypeslib.as_array(pts, shape=(n, 3))
                out, ins = self._project(p.copy())
                np.ctypeslib.as_array(proj, shape=(n, 3))[:] = np.asarray(out, F32).reshape(n, 3)
                np.ctypeslib.as_array(inside, shape=(n,))[:] = np.asarray(ins).astype(np.uint8).reshape(n)
            except BaseException as e:  # noqa: BLE001
                self._error = self._error or e
                np.ctypeslib.as_array(proj, shape=(n, 3))[:] = np.ctypeslib.as_array(pts, shape=(n, 3))
                np.ctypeslib.as_array(inside, shape=(n,))[:] = 0

        # (the ctypes thunks must outlive the registration: they hang on this object, which the coupling entry keeps)
        self._thunks = (L.HOST_AABB_FN(aabb_cb), L.HOST_PROJECT_FN(project_cb))
        self.shape = L.HostShape(self._thunks[0], self._thunks[1], None)


@dataclass
class _Entry:
    boundary: object
    body: Optional[RigidBody]
    sampling: object  # StaticSampling | DynamicContactSampling | HostShapeSampling
    uploaded: bool = False


class ColliderCouplingSet:
    """fluids_pipeline.rs:64-136 — one entry per coupled collider; here the collider is identified by any hashable key
    and its pose is the body's (a collider attached at the body origin)."""

    def __init__(self):
        self.entries: Dict[object, _Entry] = {}

    def register_coupling(self, boundary, collider, body: Optional[RigidBody], sampling_method):
        old = self.entries.get(collider)
        if old is not None:
            self._detach(old)
        self.entries[collider] = _Entry(boundary, body, sampling_method)
        return old.boundary if old else None

    def unregister_coupling(self, collider):
        e = self.entries.pop(collider, None)
        if e is not None:
            self._detach(e)
        return e.boundary if e else None

    @staticmethod
    def _detach(e: _Entry):
        """fluids_pipeline.rs:116-125: the entry goes, the boundary stays in the world with the particles it holds.  The library
        must forget the sampling method BEFORE the entry is dropped: a host shape's ctypes thunks die with it, and every step calls
        them while they are registered (salva_hip_clear_boundary_sampling)."""
        b = e.boundary
        w = b._world
        if not e.uploaded or w is None or b._slot < 0:
            return
        pos, vel = w._boundary_particles(b)
        L.check(w._L.salva_hip_clear_boundary_sampling(w._h, b._slot))
        b._positions, b._velocities = np.ascontiguousarray(pos, F32).copy(), np.ascontiguousarray(vel, F32).copy()
        b._sampled = b._dynamic = False
        b._n_sampled = 0
        b._dirty = False  # (the device already holds exactly these particles)
        e.uploaded = False

    def raise_pending(self):
        """Re-raise an exception a host-shape callback parked during the last step (HostShapeSampling)."""
        for e in self.entries.values():
            err = getattr(e.sampling, "_error", None)
            if err is not None:
                e.sampling._error = None
                raise err

    # CouplingManager::update_boundaries (:146-264), StaticSampling arm
    def update_boundaries(self, world):
        for e in self.entries.values():
            b = e.boundary
            if b._world is not world:
                continue
            if not e.uploaded and isinstance(e.sampling, HostShapeSampling):
                b._sampled = b._dynamic = True
                L.check(world._L.salva_hip_set_boundary_dynamic_sampling_host(
                    world._h, b._slot, C.byref(e.sampling.shape), b.interaction_groups.memberships, b.interaction_groups.filter))
                b._dirty = False
                e.uploaded = True
            if not e.uploaded and isinstance(e.sampling, DynamicContactSampling):
                b._sampled = b._dynamic = True
                L.check(world._L.salva_hip_set_boundary_dynamic_sampling(
                    world._h, b._slot, C.byref(e.sampling.shape), b.interaction_groups.memberships, b.interaction_groups.filter))
                b._dirty = False
                e.uploaded = True
            if not e.uploaded:
                b._sampled = True
                L.check(world._L.salva_hip_set_boundary_sampling(
                    world._h, b._slot, len(e.sampling.points), e.sampling.points.ctypes.data_as(C.POINTER(C.c_float)),
                    b.interaction_groups.memberships, b.interaction_groups.filter))
                b._n_sampled = len(e.sampling.points)
                b._dirty = False
                e.uploaded = True
            if e.body is not None:
                pose = e.body.pose()
                b.wants_forces = e.body.is_dynamic()
            else:
                pose = L.RigidPose()
                pose.rotation[3] = 1.0
            L.check(world._L.salva_hip_update_boundary_pose(world._h, b._slot, C.byref(pose)))

    # CouplingManager::transmit_forces (:266-287)
    def transmit_forces(self, world, dt: float):
        for e in self.entries.values():
            b = e.boundary
            if b._world is not world or e.body is None or not b.wants_forces or b.num_particles() == 0:
                continue
            com = e.body.center_of_mass()
            f = np.zeros(3, F32)
            t = np.zeros(3, F32)
            fp = C.POINTER(C.c_float)
            L.check(world._L.salva_hip_get_boundary_wrench(world._h, b._slot, com.ctypes.data_as(fp), f.ctypes.data_as(fp),
                                                           t.ctypes.data_as(fp)))
            e.body.apply_impulse(f * F32(dt))
            e.body.apply_torque_impulse(t * F32(dt))


class FluidsPipeline:
    """integrations/rapier/fluids_pipeline.rs:18-61: the liquid world (always DFSPH there, :35) and the coupling set in one object.
    `step(gravity, dt, colliders, bodies)` is `liquid_world.step_with_coupling(dt, gravity, &mut coupling.as_manager_mut(colliders,
    bodies))` (:48-60); in this mirror the bodies hang on the coupling entries (`register_coupling(boundary, collider, body,
    sampling)`), so the two set arguments are accepted for signature compatibility and not looked at."""

    def __init__(self, particle_radius: float, smoothing_factor: float):
        from .world import DFSPHSolver, LiquidWorld  # (world.py imports nothing from here: no cycle at call time)

        self.liquid_world = LiquidWorld(DFSPHSolver(), particle_radius, smoothing_factor)
        self.coupling = ColliderCouplingSet()

    def step(self, gravity, dt: float, colliders=None, bodies=None):
        return self.liquid_world.step_with_coupling(dt, gravity, self.coupling)
