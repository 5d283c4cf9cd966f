"""Host-side mirror of the salva3d API for the `LiquidWorld::step` path, on top of libsalva_hip.

Same names, argument meaning and error behaviour as the reference (paths relative to /root/reference):
  LiquidWorld            src/liquid_world.rs:17-209
  Fluid / Boundary       src/object/fluid.rs:12-185, src/object/boundary.rs:11-84
  InteractionGroups      src/object/interaction_groups.rs:6-79
  DFSPHSolver/IISPHSolver pub tuning fields: src/solver/pressure/dfsph_solver.rs:21-38,54-70; iisph_solver.rs:21-30,48-64
  XSPHViscosity / ArtificialViscosity / Akinci2013SurfaceTension   src/solver/{viscosity,surface_tension}/*.rs

State lives in HBM between steps.  `fluid.positions` / `fluid.velocities` are numpy views of host copies that are
refreshed lazily from the device the first time they are read after a step, and uploaded again when assigned
(`fluid.velocities = arr`) or after `fluid.mark_dirty()` for in-place edits.  This module never computes physics
itself: everything goes through the C ABI, and importing it without the built library raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L

F32 = np.float32


def _fp(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _as_vec3(a, n=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=F32).reshape(-1, 3)
    if n is not None and len(a) != n:
        raise ValueError(f"expected {n} 3-vectors, got {len(a)}")
    return a


@dataclass(frozen=True)
class InteractionGroups:
    """interaction_groups.rs: `test(a,b) = (a.m & b.f) != 0 && (b.m & a.f) != 0`; default = GROUP_1 / ALL."""
    memberships: int = 1
    filter: int = 0xFFFFFFFF

    @staticmethod
    def default() -> "InteractionGroups":
        return InteractionGroups()

    @staticmethod
    def all() -> "InteractionGroups":
        return InteractionGroups(0xFFFFFFFF, 0xFFFFFFFF)

    @staticmethod
    def none() -> "InteractionGroups":
        return InteractionGroups(0, 0)

    def test(self, rhs: "InteractionGroups") -> bool:
        return (self.memberships & rhs.filter) != 0 and (rhs.memberships & self.filter) != 0


# ------------------------------------------------------------------------------------------------ forces
class NonPressureForce:
    """solver/nonpressure_force.rs:10-30.  The built-ins below run on the device.  A subclass that implements

        solve(self, timestep, kernel_radius, fluid_fluid_contacts, fluid_boundaries_contacts, fluid, boundaries, densities)

    like the trait (examples3d/custom_forces3.rs:67-90) runs on the host in the middle of the substep, at its place in
    the fluid's force list (include/salva_hip.h, SALVA_HIP_FORCE_CUSTOM): `timestep.dt()` / `inv_dt()`, the two
    `ParticlesContacts`, a view of the fluid with this substep's `positions` / `velocities` / `volumes` / `density0` and an
    `accelerations` array to add to, the boundaries and the densities — the slow path, the state crosses PCIe."""

    def _desc(self) -> L.ForceDesc:
        if not callable(getattr(self, "solve", None)):
            raise NotImplementedError("a custom NonPressureForce must implement solve(...)")
        d = L.ForceDesc()
        d.kind = L.FORCE_CUSTOM
        return d


class TimestepView:
    """The two TimestepManager getters forces use (timestep_manager.rs:60-72)."""

    def __init__(self, dt, inv_dt):
        self._dt, self._inv_dt = dt, inv_dt

    def dt(self):
        return self._dt

    def inv_dt(self):
        return self._inv_dt


@dataclass
class Contact:
    """geometry/contacts.rs:40-55"""
    i: int
    i_model: int
    j: int
    j_model: int
    weight: float
    gradient: np.ndarray


class ParticlesContacts:
    """geometry/contacts.rs:57-131 over the device's exported CSR lists: `particle_contacts(i)` yields the reference's
    `Contact`s (weight and gradient evaluated with the cubic spline, kernel/cubic_spline_kernel.rs); the raw arrays
    (`offsets`, `j_model`, `j`) are there for vectorised forces."""

    def __init__(self, i_model, offsets, j_model, j, positions_i, positions_of_model, h):
        self.i_model, self.offsets, self.j_model, self.j = i_model, offsets, j_model, j
        self._pi, self._pj, self._h = positions_i, positions_of_model, h

    def len(self):
        return len(self.offsets) - 1

    def particle_contacts(self, i):
        out = []
        for k in range(int(self.offsets[i]), int(self.offsets[i + 1])):
            jm, j = int(self.j_model[k]), int(self.j[k])
            d = (self._pi[i] - self._pj(jm)[j]).astype(F32)
            r = F32(np.sqrt(F32(d[0] * d[0] + d[1] * d[1]) + F32(d[2] * d[2])))
            w, dw = _cubic_spline(r, F32(self._h))
            grad = (d / r * dw).astype(F32) if r > np.finfo(F32).eps else np.zeros(3, F32)
            out.append(Contact(i, self.i_model, j, jm, float(w), grad))
        return out


def _cubic_spline(r, h):
    """CubicSplineKernel::{scalar_apply, scalar_apply_diff} (cubic_spline_kernel.rs:12-33, 55-79) in f32."""
    q = F32(r / h)
    norm = F32(8.0 / np.pi) / (h * h * h)
    if q <= 0.5:
        return norm * (F32(6) * (q * q * q - q * q) + F32(1)), norm * F32(6) * (F32(3) * q * q - F32(2) * q) / h
    if q <= 1.0:
        return norm * F32(2) * (F32(1) - q) ** 3, -norm * F32(6) * (F32(1) - q) ** 2 / h
    return F32(0), F32(0)


class FluidView:
    """What `solve` sees of the `Fluid` (object/fluid.rs:12-34)."""

    def __init__(self, positions, velocities, volumes, density0):
        self.positions, self.velocities, self.volumes, self.density0 = positions, velocities, volumes, density0
        self.accelerations = np.zeros_like(positions)

    def num_particles(self):
        return len(self.positions)

    def particle_mass(self, i):
        return self.volumes[i] * self.density0


class XSPHViscosity(NonPressureForce):
    def __init__(self, fluid_viscosity_coefficient: float, boundary_viscosity_coefficient: float):
        self.fluid_viscosity_coefficient = fluid_viscosity_coefficient
        self.boundary_viscosity_coefficient = boundary_viscosity_coefficient

    def _desc(self):
        d = L.ForceDesc()
        d.kind = L.FORCE_XSPH
        d.p[0], d.p[1] = self.fluid_viscosity_coefficient, self.boundary_viscosity_coefficient
        return d


class ArtificialViscosity(NonPressureForce):
    def __init__(self, fluid_viscosity_coefficient: float, boundary_viscosity_coefficient: float):
        self.alpha = 1.0
        self.beta = 0.0
        self.speed_of_sound = 10.0
        self.fluid_viscosity_coefficient = fluid_viscosity_coefficient
        self.boundary_viscosity_coefficient = boundary_viscosity_coefficient

    def _desc(self):
        d = L.ForceDesc()
        d.kind = L.FORCE_ARTIFICIAL
        d.p[0], d.p[1] = self.fluid_viscosity_coefficient, self.boundary_viscosity_coefficient
        d.p[2], d.p[3], d.p[4] = self.alpha, self.beta, self.speed_of_sound
        return d


class Akinci2013SurfaceTension(NonPressureForce):
    def __init__(self, fluid_tension_coefficient: float, boundary_adhesion_coefficient: float):
        self.fluid_tension_coefficient = fluid_tension_coefficient
        self.boundary_adhesion_coefficient = boundary_adhesion_coefficient

    def _desc(self):
        d = L.ForceDesc()
        d.kind = L.FORCE_AKINCI2013
        d.p[0], d.p[1] = self.fluid_tension_coefficient, self.boundary_adhesion_coefficient
        return d


class He2014SurfaceTension(NonPressureForce):
    """solver::He2014SurfaceTension::new(fluid_tension_coefficient, boundary_tension_coefficient), he2014_surface_tension.rs:12-29."""

    def __init__(self, fluid_tension_coefficient: float, boundary_tension_coefficient: float):
        self.fluid_tension_coefficient = fluid_tension_coefficient
        self.boundary_tension_coefficient = boundary_tension_coefficient

    def _desc(self):
        d = L.ForceDesc()
        d.kind = L.FORCE_HE2014
        d.p[0], d.p[1] = self.fluid_tension_coefficient, self.boundary_tension_coefficient
        return d


class WCSPHSurfaceTension(NonPressureForce):
    """solver::WCSPHSurfaceTension::new(fluid_tension_coefficient, boundary_tension_coefficient), wcsph_surface_tension.rs:15-28.
    A non-zero boundary coefficient is rejected at step time: the reference's boundary loop (:66-83) panics."""

    def __init__(self, fluid_tension_coefficient: float, boundary_tension_coefficient: float):
        self.fluid_tension_coefficient = fluid_tension_coefficient
        self.boundary_tension_coefficient = boundary_tension_coefficient

    def _desc(self):
        d = L.ForceDesc()
        d.kind = L.FORCE_WCSPH_TENSION
        d.p[0], d.p[1] = self.fluid_tension_coefficient, self.boundary_tension_coefficient
        return d


# ------------------------------------------------------------------------------------------------ solvers
class DFSPHViscosity(NonPressureForce):
    """solver::DFSPHViscosity (viscous DFSPH), dfsph_viscosity.rs:85-125: `new(viscosity_coefficient)` + pub tuning fields.
    After a step, `num_iterations` / `last_error` hold what its solve loop did (the reference only has a commented-out print)."""

    def __init__(self, viscosity_coefficient: float):
        # assert!(viscosity_coefficient >= 0 && <= 1, "The viscosity coefficient must be between 0.0 and 1.0.") :104-108
        if not (0.0 <= viscosity_coefficient <= 1.0):
            raise ValueError("The viscosity coefficient must be between 0.0 and 1.0.")
        self.min_viscosity_iter = 1
        self.max_viscosity_iter = 50
        self.max_viscosity_error = 0.01
        self.viscosity_coefficient = float(viscosity_coefficient)
        self.num_iterations = 0
        self.last_error = 0.0

    def _desc(self):
        d = L.ForceDesc()
        d.kind = L.FORCE_DFSPH_VISCOSITY
        d.p[0] = self.viscosity_coefficient
        d.p[1], d.p[2], d.p[3] = self.min_viscosity_iter, self.max_viscosity_iter, self.max_viscosity_error
        return d


class CubicSplineKernel:
    """kernel/cubic_spline_kernel.rs — the default KernelDensity and KernelGradient."""
    kind = L.KERNEL_CUBIC_SPLINE


class Poly6Kernel:
    """kernel/poly6_kernel.rs"""
    kind = L.KERNEL_POLY6


class SpikyKernel:
    """kernel/spiky_kernel.rs"""
    kind = L.KERNEL_SPIKY


class ViscosityKernel:
    """kernel/viscosity_kernel.rs"""
    kind = L.KERNEL_VISCOSITY


class DFSPHSolver:
    """dfsph_solver.rs:54-70 defaults.  `DFSPHSolver(KernelDensity, KernelGradient)` mirrors the type parameters of
    `DFSPHSolver<KernelDensity, KernelGradient>` (:17-20); both default to CubicSplineKernel."""
    kind = L.SOLVER_DFSPH

    def __init__(self, kernel_density=CubicSplineKernel, kernel_gradient=CubicSplineKernel):
        self.kernel_density, self.kernel_gradient = kernel_density, kernel_gradient
        self.min_pressure_iter = 1
        self.max_pressure_iter = 50
        self.max_density_error = 0.05
        self.min_divergence_iter = 1
        self.max_divergence_iter = 50
        self.max_divergence_error = 0.1


class IISPHSolver:
    """iisph_solver.rs:48-64 defaults; `IISPHSolver(KernelDensity, KernelGradient)` as for DFSPHSolver (:17-20)."""
    kind = L.SOLVER_IISPH

    def __init__(self, kernel_density=CubicSplineKernel, kernel_gradient=CubicSplineKernel):
        self.kernel_density, self.kernel_gradient = kernel_density, kernel_gradient
        self.min_pressure_iter = 1
        self.max_pressure_iter = 50
        self.max_density_error = 0.05
        self.min_divergence_iter = 1
        self.max_divergence_iter = 50
        self.max_divergence_error = 0.1


# ------------------------------------------------------------------------------------------------ objects
class Fluid:
    """object/fluid.rs.  `Fluid::new(positions, particle_radius, density0, interaction_groups)`."""

    def __init__(self, particle_positions, particle_radius: float, density0: float,
                 interaction_groups: InteractionGroups = InteractionGroups()):
        pos = _as_vec3(particle_positions) if len(particle_positions) else np.zeros((0, 3), F32)
        n = len(pos)
        self.nonpressure_forces: List[NonPressureForce] = []
        self._positions = pos.copy()
        self._velocities = np.zeros((n, 3), F32)
        self._accelerations = np.zeros((n, 3), F32)
        self._volumes = np.full(n, self.particle_volume(particle_radius), F32)
        self.density0 = float(density0)
        self._deleted = np.zeros(n, bool)
        self._particle_radius = float(particle_radius)
        self.interaction_groups = interaction_groups
        # device synchronisation state
        self._world: Optional["LiquidWorld"] = None
        self._slot = -1
        self._dirty = L.DIRTY_ALL
        self._resized = True
        self._device_newer = False
        self._acc_set = False
        self._acc_touched = False   # host accelerations may be non-zero (cleared after the step only then: 12 MB at 10^6 particles)
        self._maybe_deleted = False # delete_particle_at_next_timestep was called since the last removal

    @staticmethod
    def particle_volume(particle_radius: float) -> np.float32:
        r = F32(particle_radius)  # fluid.rs:110-120
        return F32(r * r * r * F32(8.0 * 0.8))

    # ---- lazily synchronised pub fields
    def _pull(self):
        if self._device_newer and self._world is not None:
            self._world._download_fluid(self)

    @property
    def positions(self) -> np.ndarray:
        self._pull()
        return self._positions

    @positions.setter
    def positions(self, v):
        self._pull()
        self._positions = _as_vec3(v, self.num_particles()).copy()
        self._dirty |= L.DIRTY_POSITIONS

    @property
    def velocities(self) -> np.ndarray:
        self._pull()
        return self._velocities

    @velocities.setter
    def velocities(self, v):
        self._pull()
        self._velocities = _as_vec3(v, self.num_particles()).copy()
        self._dirty |= L.DIRTY_VELOCITIES

    @property
    def accelerations(self) -> np.ndarray:
        self._acc_touched = True  # the caller may write into the array
        return self._accelerations

    @accelerations.setter
    def accelerations(self, v):
        self._accelerations = _as_vec3(v, self.num_particles()).copy()
        self._dirty |= L.DIRTY_ACCELERATIONS
        self._acc_set = True
        self._acc_touched = True

    @property
    def volumes(self) -> np.ndarray:
        return self._volumes

    @volumes.setter
    def volumes(self, v):
        v = np.ascontiguousarray(v, dtype=F32).reshape(-1)
        if len(v) != self.num_particles():
            raise ValueError("volumes length mismatch")
        self._volumes = v.copy()
        self._dirty |= L.DIRTY_VOLUMES

    def mark_dirty(self, mask: int = L.DIRTY_POSITIONS | L.DIRTY_VELOCITIES | L.DIRTY_VOLUMES):
        """Call after editing the arrays in place (numpy cannot observe element writes)."""
        self._dirty |= mask

    # ---- fluid.rs API
    def particle_radius(self) -> float:
        return self._particle_radius

    def default_particle_volume(self) -> float:
        return float(self.particle_volume(self._particle_radius))

    def num_particles(self) -> int:
        return len(self._positions)

    def particle_mass(self, i: int) -> np.float32:
        return F32(self._volumes[i] * F32(self.density0))

    def delete_particle_at_next_timestep(self, particle: int):
        self._deleted[particle] = True
        self._maybe_deleted = True

    def num_deleted_particles(self) -> int:
        return int(self._deleted.sum())

    def deleted_particles_mask(self) -> np.ndarray:
        self._maybe_deleted = True  # the caller may write into the mask
        return self._deleted

    def add_particles(self, positions, velocities=None):
        """fluid.rs:126-150.  A fluid that already lives on the device grows there (salva_hip_add_particles): nothing it
        holds is re-uploaded or downloaded."""
        pos = _as_vec3(positions)
        k = len(pos)
        vel = _as_vec3(velocities, k) if velocities is not None else np.zeros((k, 3), F32)
        w = self._world
        if w is not None and not self._resized and self._dirty and k:
            # host edits of the existing particles go up first (no removal, nothing resized), so that the append below can run
            # on the device: only there do the new particles see what the reference's `velocity_changes[slot].resize(n)`
            # would hand them (World::sticky)
            w._sync_fluid(self, apply_removal=False)
        if w is not None and not self._resized and not self._dirty and k:  # (pending deletions stay pending: indices are unchanged)
            L.check(w._L.salva_hip_add_particles(w._h, self._slot, k, _fp(pos), _fp(vel)))
            self._positions = np.concatenate([self._positions, pos])  # stale rows are refreshed by the next _pull()
            self._velocities = np.concatenate([self._velocities, vel])
            self._accelerations = np.concatenate([self._accelerations, np.zeros((k, 3), F32)])
            self._volumes = np.concatenate([self._volumes, np.full(k, self.default_particle_volume(), F32)])
            self._deleted = np.concatenate([self._deleted, np.zeros(k, bool)])
            return
        self._pull()
        dv = self._world._fetch_velocity_changes(self) if (self._world is not None and not self._resized) else None
        self._positions = np.concatenate([self._positions, pos])
        self._velocities = np.concatenate([self._velocities, vel])
        self._accelerations = np.concatenate([self._accelerations, np.zeros((k, 3), F32)])
        self._volumes = np.concatenate([self._volumes, np.full(k, self.default_particle_volume(), F32)])
        self._deleted = np.concatenate([self._deleted, np.zeros(k, bool)])
        if dv is None and self._pending_dv is not None:
            dv = self._pending_dv  # an earlier host-side edit already holds the solver state of this fluid
        if dv is not None:  # init_with_fluids resizes velocity_changes / pressures with zeros (dfsph_solver.rs:548, iisph_solver.rs:499)
            self._pending_dv = np.concatenate([dv, np.zeros((k, 4), F32)])
        self._resized = True
        self._dirty = L.DIRTY_ALL

    def transform_by(self, rotation: Optional[np.ndarray] = None, translation: Sequence[float] = (0.0, 0.0, 0.0)):
        """`Fluid::transform_by(&Isometry)`: p <- R p + t (fluid.rs:166-168)."""
        self._pull()
        p = self._positions
        if rotation is not None:
            p = (p @ np.asarray(rotation, F32).T).astype(F32)
        self._positions = (p + np.asarray(translation, F32)[None, :]).astype(F32)
        self._dirty |= L.DIRTY_POSITIONS

    _pending_dv: Optional[np.ndarray] = None


class Boundary:
    """object/boundary.rs.  `Boundary::new(positions, interaction_groups)`; `forces = Some(..)` <=> wants_forces."""

    def __init__(self, particle_positions, interaction_groups: InteractionGroups = InteractionGroups(),
                 wants_forces: bool = False):
        pos = _as_vec3(particle_positions) if len(particle_positions) else np.zeros((0, 3), F32)
        self._positions = pos.copy()
        self._velocities = np.zeros((len(pos), 3), F32)
        self.interaction_groups = interaction_groups
        self.wants_forces = wants_forces
        self._world: Optional["LiquidWorld"] = None
        self._slot = -1
        self._dirty = True
        self._sampled = False  # positions / velocities are produced on the device from a pose (salva_amd.coupling)
        self._n_sampled = 0
        self._dynamic = False  # ColliderSampling::DynamicContactSampling: re-emitted by every step, the device knows the count

    @property
    def positions(self):
        if self._sampled and self._world is not None:
            return self._world._boundary_particles(self)[0]
        return self._positions

    @positions.setter
    def positions(self, v):
        self._positions = _as_vec3(v).copy()
        if len(self._velocities) != len(self._positions):
            self._velocities = np.zeros((len(self._positions), 3), F32)
        self._dirty = True

    @property
    def velocities(self):
        if self._sampled and self._world is not None:
            return self._world._boundary_particles(self)[1]
        return self._velocities

    @velocities.setter
    def velocities(self, v):
        self._velocities = _as_vec3(v, len(self._positions)).copy()
        self._dirty = True

    def mark_dirty(self):
        self._dirty = True

    def num_particles(self) -> int:
        if self._dynamic and self._world is not None:
            return int(self._world._L.salva_hip_boundary_len(self._world._h, self._slot))
        return self._n_sampled if self._sampled else len(self._positions)

    def sources(self):
        """Dynamically sampled boundaries: (fluid slot, particle index) behind each boundary particle; in a decomposed world the
        index is the particle's global id (LiquidWorld.owned() / local_view()), which another rank may hold."""
        n = self.num_particles()
        f, i = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        if n:
            u32p = C.POINTER(C.c_uint32)
            L.check(self._world._L.salva_hip_get_boundary_sources(self._world._h, self._slot, f.ctypes.data_as(u32p), i.ctypes.data_as(u32p)))
        return f, i

    @property
    def volumes(self) -> np.ndarray:
        """boundary.volumes: V_b = 1 / sum W (recomputed by the solver, dfsph_solver.rs:72-96)."""
        if self._world is None:
            return np.zeros(self.num_particles(), F32)
        return self._world._boundary_field(self, volumes=True)

    @property
    def forces(self) -> Optional[np.ndarray]:
        if not self.wants_forces:
            return None
        if self._world is None:
            return np.zeros((self.num_particles(), 3), F32)
        return self._world._boundary_field(self, volumes=False)

    def clear_forces(self, resize_buffer: bool = False):
        if self._world is not None and self.wants_forces:
            self._world._sync_boundaries()
            L.check(self._world._L.salva_hip_clear_boundary_forces(self._world._h, self._slot))


@dataclass
class StagesCounters:
    """counters/stages_counters.rs:6-11 (ms)."""
    collision_detection_time: float = 0.0
    solver_time: float = 0.0


@dataclass
class CollisionDetectionCounters:
    """counters/collision_detection_counters.rs:6-17 (ms)."""
    ncontacts: int = 0
    boundary_update_time: float = 0.0
    grid_insertion_time: float = 0.0
    neighborhood_search_time: float = 0.0
    contact_sorting_time: float = 0.0


@dataclass
class SolverCounters:
    """counters/solver_counters.rs:6-11 (ms)."""
    non_pressure_resolution_time: float = 0.0
    pressure_resolution_time: float = 0.0


@dataclass
class Counters:
    """counters/mod.rs:17-72: `world.counters.{nsubsteps, step_time, custom, stages, cd, solver}` as the reference's plugins read
    them (times in ms), plus the flat fields of the device step report this mirror has always carried."""
    nsubsteps: int = 0
    step_time: float = 0.0
    custom: float = 0.0
    stages: StagesCounters = field(default_factory=StagesCounters)
    cd: CollisionDetectionCounters = field(default_factory=CollisionDetectionCounters)
    solver: SolverCounters = field(default_factory=SolverCounters)
    speculative_passes: int = 0
    discarded_passes: int = 0
    chained_passes: int = 0    # passes enqueued without a host wait between their solves (include/salva_hip.h SalvaHipCounters)
    chain_breaks: int = 0
    pregrid_adopted: int = 0   # steps whose grid part the previous step had enqueued already
    pregrid_dropped: int = 0
    light_class_passes: int = 0   # passes that ran small-halo / sparse slots in launches of their own (SalvaHipCounters)
    sparse_class_passes: int = 0
    ncontacts: int = 0
    n_divergence_iters: int = 0
    n_pressure_iters: int = 0
    divergence_error: float = 0.0
    density_error: float = 0.0
    grid_ms: float = 0.0
    solver_ms: float = 0.0
    step_ms: float = 0.0
    enabled: bool = False
    _world: object = None

    def enable(self):
        """Counters::enable (counters/mod.rs:56-63): the timers run from the next step on (salva_hip_enable_counters)."""
        self.enabled = True
        if self._world is not None:
            L.check(self._world._L.salva_hip_enable_counters(self._world._h, 1))

    def disable(self):
        """Counters::disable (counters/mod.rs:65-72) — the default (`Timer::new`, counters/timer.rs:11-18)."""
        self.enabled = False
        if self._world is not None:
            L.check(self._world._L.salva_hip_enable_counters(self._world._h, 0))


class _ObjectSet:
    """FluidSet / BoundarySet: dense slots with swap-remove (object/contiguous_arena.rs:12-135)."""

    def __init__(self):
        self._items: list = []

    def as_slice(self):
        return list(self._items)

    def get(self, handle):
        return handle if handle in self._items else None

    def __iter__(self):
        return iter(self._items)

    def __len__(self):
        return len(self._items)


class _PinnedBlock:
    """Owner of one salva_hip_host_alloc block."""

    def __init__(self, lib, p):
        self._lib, self._p = lib, p

    def __del__(self):
        if self._p:
            self._lib.salva_hip_host_free(self._p)
            self._p = None


class LiquidWorld:
    """liquid_world.rs:17-209.  `LiquidWorld::new(solver, particle_radius, smoothing_factor)`."""

    def __init__(self, solver, particle_radius: float, smoothing_factor: float, device: int = 0):
        self._L = L.lib()
        self.solver = solver
        p = L.Params()
        self._L.salva_hip_default_params(C.byref(p))
        p.particle_radius = particle_radius
        p.smoothing_factor = smoothing_factor
        p.solver = solver.kind
        p.kernel_density = getattr(solver, "kernel_density", CubicSplineKernel).kind
        p.kernel_gradient = getattr(solver, "kernel_gradient", CubicSplineKernel).kind
        p.min_pressure_iter, p.max_pressure_iter = solver.min_pressure_iter, solver.max_pressure_iter
        p.max_density_error = solver.max_density_error
        p.min_divergence_iter, p.max_divergence_iter = solver.min_divergence_iter, solver.max_divergence_iter
        p.max_divergence_error = solver.max_divergence_error
        p.device = device
        p.enable_timers = 0  # Counters start disabled, as in the reference; `world.counters.enable()` switches the timers on
        self._params = p
        h = C.c_void_p()
        L.check(self._L.salva_hip_create(C.byref(p), C.byref(h)))
        self._h = h
        self._particle_radius = float(particle_radius)
        self._fluids = _ObjectSet()
        self._boundaries = _ObjectSet()
        self._counters = Counters(_world=self)
        self._counters_stale = False
        self.last_stats = L.StepStats()
        self._nsteps = 0
        self._pending_download = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                if getattr(self, "_pending_download", None) is not None:
                    self._L.salva_hip_wait_download(h)
            finally:
                self._L.salva_hip_destroy(h)
                self._h = None

    # ---- liquid_world.rs:161-208
    def add_fluid(self, fluid: Fluid) -> Fluid:
        fluid._world, fluid._slot = self, len(self._fluids._items)
        fluid._resized, fluid._dirty = True, L.DIRTY_ALL
        fluid._forces_uploaded = None
        self._fluids._items.append(fluid)
        return fluid  # the handle is the object itself

    def add_boundary(self, boundary: Boundary) -> Boundary:
        boundary._world, boundary._slot = self, len(self._boundaries._items)
        boundary._dirty = True
        self._boundaries._items.append(boundary)
        return boundary

    def remove_fluid(self, handle: Fluid) -> Optional[Fluid]:
        if handle not in self._fluids._items:
            return None
        handle._pull()
        self._upload_new_objects()
        slot = handle._slot
        if slot < self._L.salva_hip_num_fluids(self._h):
            L.check(self._L.salva_hip_remove_fluid(self._h, slot))
        items = self._fluids._items
        last = items.pop()
        if last is not handle:
            items[slot] = last
            last._slot = slot
        handle._world, handle._slot = None, -1
        return handle

    def remove_boundary(self, handle: Boundary) -> Optional[Boundary]:
        if handle not in self._boundaries._items:
            return None
        self._upload_new_objects()
        slot = handle._slot
        if slot < self._L.salva_hip_num_boundaries(self._h):
            L.check(self._L.salva_hip_remove_boundary(self._h, slot))
        items = self._boundaries._items
        last = items.pop()
        if last is not handle:
            items[slot] = last
            last._slot = slot
        handle._world, handle._slot = None, -1
        return handle

    def fluids(self) -> _ObjectSet:
        return self._fluids

    def boundaries(self) -> _ObjectSet:
        return self._boundaries

    def h(self) -> float:
        return float(self._L.salva_hip_h(self._h))

    def particle_radius(self) -> float:
        return self._particle_radius

    # ---- host <-> device synchronisation
    def _download_fluid(self, f: Fluid):
        pend = getattr(self, "_pending_download", None)
        if pend is not None and pend[0] is f:
            self.wait_download()
            if not f._device_newer:
                return
        n = f.num_particles()
        if n:
            L.check(self._L.salva_hip_get_fluid(self._h, f._slot, _fp(f._positions), _fp(f._velocities)))
        f._device_newer = False

    # ---- asynchronous read-back (salva_hip_get_fluid_async): the renderer's per-frame read of fluid.positions / velocities
    # (testbed_plugin.rs:361-367) off the step's critical path
    def _pinned(self, n: int):
        """An (n, 3) float32 array in pinned host memory; the block is released when the last view of it is gone."""
        nfloat = max(3 * n, 1)
        p = self._L.salva_hip_host_alloc(self._h, 4 * nfloat)
        if not p:
            raise L.SalvaHipError(-1, self._L.salva_hip_last_error().decode())
        buf = (C.c_float * nfloat).from_address(p)
        buf._block = _PinnedBlock(self._L, p)  # (rides on the ctypes object numpy keeps as the array's base)
        return np.frombuffer(buf, dtype=F32, count=3 * n).reshape(n, 3)

    def download_async(self, f: Fluid):
        """Start reading `f`'s positions and velocities back as they are after the last step; returns at once.  The copy runs on
        the world's copy stream, into pinned arrays, while the next `step` computes; `wait_download()` completes it and returns
        the two arrays.  Two pairs of arrays alternate, so the pair returned by one wait stays intact while the next read-back is
        in flight (render frame k while frame k + 1 is copied)."""
        self.wait_download()
        n = f.num_particles()
        if n == 0:
            return
        pins = getattr(f, "_pins", None)
        if pins is None or len(pins[0][0]) != n:
            f._pins = pins = [(self._pinned(n), self._pinned(n)), (self._pinned(n), self._pinned(n))]
            f._pin_next = 0
        pin = pins[f._pin_next]
        f._pin_next ^= 1
        self.sync_to_device()
        L.check(self._L.salva_hip_get_fluid_async(self._h, f._slot, _fp(pin[0]), _fp(pin[1])))
        self._pending_download = (f, pin, self._nsteps)

    def wait_download(self):
        """Complete the read-back started by `download_async`: (positions, velocities) of the state it was started from (None
        when nothing is pending).  If no step has run since, they also become f.positions / f.velocities (no copy)."""
        pend = getattr(self, "_pending_download", None)
        if pend is None:
            return None
        self._pending_download = None
        f, pin, at = pend
        L.check(self._L.salva_hip_wait_download(self._h))
        if at == self._nsteps and f._device_newer:
            f._positions, f._velocities = pin
            f._device_newer = False
        return pin

    def _fetch_velocity_changes(self, f: Fluid) -> np.ndarray:
        """The solver state that must survive a re-upload of the fluid, n x 4: velocity_changes (dfsph_solver.rs:41) and, in
        the last column, the pressures IISPH warm-starts from (iisph_solver.rs:35)."""
        n = f.num_particles()
        out = np.zeros((n, 4), F32)
        if n and f._slot < self._L.salva_hip_num_fluids(self._h) and self._L.salva_hip_fluid_len(self._h, f._slot) == n:
            dv, p = np.zeros((n, 3), F32), np.zeros(n, F32)
            L.check(self._L.salva_hip_get_fluid_field(self._h, f._slot, L.FIELD_VELOCITY_CHANGE, _fp(dv)))
            L.check(self._L.salva_hip_get_fluid_field(self._h, f._slot, L.FIELD_PRESSURE, _fp(p)))
            out[:, :3], out[:, 3] = dv, p
        return out

    def _apply_particles_removal(self, f: Fluid):
        """fluid.rs:88-98 + the compaction of the solver's buffers (dfsph_solver.rs:550-560)."""
        if not f._maybe_deleted:
            return
        f._maybe_deleted = False
        if not f._deleted.any():
            return
        if not f._resized and not f._dirty and f._pending_dv is None:
            # the fluid is current on the device: compact it there (salva_hip_delete_particles) and the host copies
            # with the same mask — no download, no re-upload
            mask = np.ascontiguousarray(f._deleted, np.uint8)
            kept = int(self._L.salva_hip_delete_particles(self._h, f._slot, mask.ctypes.data_as(C.POINTER(C.c_uint8))))
            if kept < 0:
                L.check(kept)
            keep = ~f._deleted
            f._positions = np.ascontiguousarray(f._positions[keep])
            f._velocities = np.ascontiguousarray(f._velocities[keep])
            f._accelerations = np.ascontiguousarray(f._accelerations[keep])
            f._volumes = np.ascontiguousarray(f._volumes[keep])
            f._deleted = np.zeros(kept, bool)
            assert kept == len(f._positions)
            return
        f._pull()
        dv = f._pending_dv if f._pending_dv is not None else (
            self._fetch_velocity_changes(f) if not f._resized else np.zeros((f.num_particles(), 4), F32))
        keep = ~f._deleted
        f._positions = np.ascontiguousarray(f._positions[keep])
        f._velocities = np.ascontiguousarray(f._velocities[keep])
        f._accelerations = np.ascontiguousarray(f._accelerations[keep])
        f._volumes = np.ascontiguousarray(f._volumes[keep])
        f._pending_dv = np.ascontiguousarray(dv[keep])
        f._deleted = np.zeros(len(f._positions), bool)
        f._resized, f._dirty = True, L.DIRTY_ALL

    def _upload_new_objects(self):
        """Objects added since the last sync exist only on the host; the swap-removes below must see the same dense sets
        on both sides (found by tests/test_fuzz_gpu.py).  Pending particle deletions stay pending (fluid.rs:88-98: they are
        applied at the top of the next step, and the indices the caller holds refer to the uncompacted arrays until then)."""
        nf = self._L.salva_hip_num_fluids(self._h)
        for f in self._fluids:
            if f._slot >= nf:
                self._sync_fluid(f, apply_removal=False)
        self._sync_boundaries()

    def _sync_fluid(self, f: Fluid, apply_removal: bool = True):
        if apply_removal and f._maybe_deleted and f._slot >= self._L.salva_hip_num_fluids(self._h):
            # a fluid that was never uploaded: upload it uncompacted first — the reference resizes the slot's solver buffer
            # to the full particle count and filters afterwards (dfsph_solver.rs:543-560), and the buffer may be inherited
            self._sync_fluid(f, apply_removal=False)
        if apply_removal:
            self._apply_particles_removal(f)
        descs = (L.ForceDesc * max(len(f.nonpressure_forces), 1))()
        for k, force in enumerate(f.nonpressure_forces):
            descs[k] = force._desc()
            if descs[k].kind == L.FORCE_CUSTOM:
                self._install_force_callback()
        if f._resized or f._dirty:
            n = f.num_particles()
            dirty = L.DIRTY_ALL if f._resized else f._dirty
            acc = f._accelerations if (f._acc_set and (dirty & L.DIRTY_ACCELERATIONS)) else None
            state = f._pending_dv if f._resized else None
            dv = np.ascontiguousarray(state[:, :3]) if state is not None else None
            L.check(self._L.salva_hip_set_fluid(
                self._h, f._slot, n, _fp(f._positions), _fp(f._velocities), _fp(f._volumes), _fp(acc), _fp(dv),
                f.density0, f.interaction_groups.memberships, f.interaction_groups.filter, dirty))
            if state is not None and n and state[:, 3].any():
                L.check(self._L.salva_hip_set_fluid_field(self._h, f._slot, L.FIELD_PRESSURE, _fp(np.ascontiguousarray(state[:, 3]))))
            f._resized, f._dirty, f._pending_dv, f._acc_set = False, 0, None, False
        # (the force list goes down again only when it, or a coefficient in it, changed: a ctypes call per fluid and step otherwise)
        sig = (f._slot, len(f.nonpressure_forces), bytes(descs))
        if getattr(f, "_forces_uploaded", None) != sig:
            L.check(self._L.salva_hip_set_fluid_forces(self._h, f._slot, descs, len(f.nonpressure_forces)))
            f._forces_uploaded = sig

    def _sync_boundaries(self):
        for b in self._boundaries:
            if b._dirty and not b._sampled:
                L.check(self._L.salva_hip_set_boundary(
                    self._h, b._slot, b.num_particles(), _fp(b._positions), _fp(b._velocities),
                    b.interaction_groups.memberships, b.interaction_groups.filter, int(b.wants_forces)))
                b._dirty = False

    def _boundary_particles(self, b: Boundary):
        n = b.num_particles()
        pos, vel = np.zeros((n, 3), F32), np.zeros((n, 3), F32)
        if n:
            L.check(self._L.salva_hip_get_boundary_particles(self._h, b._slot, _fp(pos), _fp(vel)))
        return pos, vel

    def _boundary_field(self, b: Boundary, volumes: bool) -> np.ndarray:
        self._sync_boundaries()
        n = b.num_particles()
        if volumes:
            out = np.zeros(n, F32)
            if n:
                L.check(self._L.salva_hip_get_boundary(self._h, b._slot, _fp(out), None))
        else:
            out = np.zeros((n, 3), F32)
            if n:
                L.check(self._L.salva_hip_get_boundary(self._h, b._slot, None, _fp(out)))
        return out

    def sync_to_device(self, apply_removal: bool = True):
        """Upload what the host changed.  `apply_removal=False` leaves particles marked with
        delete_particle_at_next_timestep in place (they exist until the next step, fluid.rs:88-98)."""
        for f in self._fluids:
            self._sync_fluid(f, apply_removal)
        self._sync_boundaries()

    @property
    def counters(self) -> Counters:
        """liquid_world.counters (counters/mod.rs:17-72) of the last step; read from the device library on first access after
        a step (a Python loop that only steps does not pay for it)."""
        if self._counters_stale:
            self._counters_stale = False
            c, st = self._counters, self.last_stats
            c.ncontacts = int(st.ncontacts)
            c.n_divergence_iters, c.n_pressure_iters = st.n_divergence_iters, st.n_pressure_iters
            c.divergence_error, c.density_error = st.divergence_error, st.density_error
            c.grid_ms, c.solver_ms, c.step_ms = st.grid_ms, st.solver_ms, st.step_ms
            t = L.CountersStruct()
            L.check(self._L.salva_hip_get_counters(self._h, C.byref(t)))
            c.nsubsteps, c.step_time, c.custom = int(t.nsubsteps), t.step_time, t.custom
            c.stages = StagesCounters(t.stages.collision_detection_time, t.stages.solver_time)
            c.cd = CollisionDetectionCounters(int(t.cd.ncontacts), t.cd.boundary_update_time, t.cd.grid_insertion_time,
                                              t.cd.neighborhood_search_time, t.cd.contact_sorting_time)
            c.solver = SolverCounters(t.solver.non_pressure_resolution_time, t.solver.pressure_resolution_time)
            c.speculative_passes, c.discarded_passes = int(t.speculative_passes), int(t.discarded_passes)
            c.chained_passes, c.chain_breaks = int(t.chained_passes), int(t.chain_breaks)
            c.pregrid_adopted, c.pregrid_dropped = int(t.pregrid_adopted), int(t.pregrid_dropped)
            c.light_class_passes, c.sparse_class_passes = int(t.light_class_passes), int(t.sparse_class_passes)
        return self._counters

    # ---- liquid_world.rs:62-158
    def step(self, dt: float, gravity=(0.0, -9.81, 0.0)) -> L.StepStats:
        self.sync_to_device()
        gk = tuple(gravity)
        if getattr(self, "_gravity_key", None) != gk:
            self._gravity_key, self._gravity_c = gk, (C.c_float * 3)(*[float(x) for x in gravity])
        g = self._gravity_c
        st = L.StepStats()
        rc = self._L.salva_hip_step(self._h, dt, g, C.byref(st))
        self._nsteps += 1
        for f in self._fluids:
            f._device_newer = True
            if f._acc_touched:
                f._accelerations[:] = 0  # integrate_and_clear_accelerations
                f._acc_touched = False
        err, self._force_cb_error = getattr(self, "_force_cb_error", None), None
        if err is not None:
            raise err
        L.check(rc)
        for f in self._fluids:
            for k, force in enumerate(f.nonpressure_forces):
                if isinstance(force, DFSPHViscosity):
                    it, err = C.c_int32(0), C.c_float(0)
                    L.check(self._L.salva_hip_get_force_stats(self._h, f._slot, k, C.byref(it), C.byref(err)))
                    force.num_iterations, force.last_error = it.value, err.value
        self.last_stats = st
        self._counters_stale = True  # the Counters tree is fetched when somebody looks at it (`world.counters`), not on every step
        return st

    # ---- host NonPressureForce::solve in the middle of the substep (SALVA_HIP_FORCE_CUSTOM)
    def _install_force_callback(self):
        if getattr(self, "_force_cb", None) is not None:
            return

        def callback(_user, _world, slot, index, dt, inv_dt):
            try:
                f = next(x for x in self._fluids if x._slot == slot)
                force = f.nonpressure_forces[index]
                if getattr(self, "_comm", None) is not None:
                    self._solve_custom_force_locally(f, force, dt, inv_dt)
                    return 0
                n = f.num_particles()
                pos, vel, dens = np.zeros((n, 3), F32), np.zeros((n, 3), F32), np.zeros(n, F32)
                L.check(self._L.salva_hip_force_get_state(self._h, slot, _fp(pos), _fp(vel), _fp(dens)))
                fluid_pos = {}

                def positions_of_fluid(m):
                    if m == slot:
                        return pos
                    if m not in fluid_pos:
                        g = next(x for x in self._fluids if x._slot == m)
                        p = np.zeros((g.num_particles(), 3), F32)
                        L.check(self._L.salva_hip_force_get_state(self._h, m, _fp(p), None, None))
                        fluid_pos[m] = p
                    return fluid_pos[m]

                bpos = {}

                def positions_of_boundary(m):
                    if m not in bpos:
                        b = next(x for x in self._boundaries if x._slot == m)
                        bpos[m] = self._boundary_particles(b)[0]
                    return bpos[m]

                ff = ParticlesContacts(slot, *self.fluid_contacts(f, False), pos, positions_of_fluid, self.h())
                fb = ParticlesContacts(slot, *self.fluid_contacts(f, True), pos, positions_of_boundary, self.h())
                view = FluidView(pos, vel, np.asarray(f.volumes, F32), f.density0)
                force.solve(TimestepView(dt, inv_dt), self.h(), ff, fb, view, list(self._boundaries), dens)
                acc = np.ascontiguousarray(view.accelerations, F32)
                L.check(self._L.salva_hip_force_add_accelerations(self._h, slot, _fp(acc)))
                return 0
            except BaseException as e:  # noqa: BLE001 - reported through the step's error
                self._force_cb_error = e
                return 1

        self._force_cb = L.FORCE_CALLBACK(callback)
        self._force_cb_error = None
        L.check(self._L.salva_hip_set_force_callback(self._h, self._force_cb, None))

    # ---- the working set as it is (include/salva_hip.h "local view"): what a rank of a decomposed run can look at
    def local_view(self):
        """dict of the particles this world holds after the last step (or inside a force callback), in the order of its cell sort:
        ids (global ids in a decomposed run), fluid_slots, is_ghost, positions, velocities, densities, volumes."""
        n = int(self._L.salva_hip_local_len(self._h))
        v = dict(ids=np.zeros(n, np.uint32), fluid_slots=np.zeros(n, np.uint32), is_ghost=np.zeros(n, np.uint8),
                 positions=np.zeros((n, 3), F32), velocities=np.zeros((n, 3), F32), densities=np.zeros(n, F32), volumes=np.zeros(n, F32))
        if n:
            u32p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
            L.check(self._L.salva_hip_get_local(self._h, v["ids"].ctypes.data_as(u32p), v["fluid_slots"].ctypes.data_as(u32p),
                                                v["is_ghost"].ctypes.data_as(u8p), _fp(v["positions"]), _fp(v["velocities"]),
                                                _fp(v["densities"]), _fp(v["volumes"])))
        v["is_ghost"] = v["is_ghost"].astype(bool)
        return v

    def local_contacts(self, boundary: bool = False):
        """(offsets[n + 1], j_model, j) over the local view: `j` is a local index (fluid-fluid) or an index into boundary j_model's
        arrays as this rank uploaded them."""
        n = int(self._L.salva_hip_local_len(self._h))
        offsets = np.zeros(n + 1, np.uint64)
        u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        total = int(self._L.salva_hip_get_local_contacts(self._h, int(boundary), offsets.ctypes.data_as(u64p), None, None, 0))
        if total < 0:
            L.check(total)
        jm, j = np.zeros(max(total, 1), np.uint32), np.zeros(max(total, 1), np.uint32)
        if total:
            rc = int(self._L.salva_hip_get_local_contacts(self._h, int(boundary), offsets.ctypes.data_as(u64p), jm.ctypes.data_as(u32p),
                                                          j.ctypes.data_as(u32p), total))
            if rc < 0:
                L.check(rc)
        return offsets, jm[:total], j[:total]

    def _solve_custom_force_locally(self, f, force, dt, inv_dt):
        """A user `NonPressureForce::solve` on ONE rank of a decomposed run: the fluid it sees is the rank's part of it (owned
        particles and ghosts, local order), the contacts are the rank's lists re-indexed to that part; the accelerations it adds
        to the ghosts are dropped (their owners compute them)."""
        lv = self.local_view()
        slot = f._slot
        slots = lv["fluid_slots"]
        within = np.zeros(len(slots), np.int64)  # index of a local particle inside its own fluid's local part
        parts = {}
        for m in np.unique(slots):
            sel = np.nonzero(slots == m)[0]
            within[sel] = np.arange(len(sel))
            parts[int(m)] = sel
        sel = parts.get(slot, np.zeros(0, np.int64))

        def rows_of(offsets, jm, j, fluid_neighbours):
            cnt = (offsets[1:] - offsets[:-1]).astype(np.int64)[sel]
            new_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
            idx = np.concatenate([np.arange(int(offsets[i]), int(offsets[i + 1])) for i in sel]) if len(sel) else np.zeros(0, np.int64)
            jm2, j2 = jm[idx], j[idx]
            if fluid_neighbours:
                j2 = within[j2].astype(np.uint32)
            return new_off, jm2, j2

        pos = lv["positions"]
        bpos = {}

        def positions_of_boundary(m):
            if m not in bpos:
                b = next(x for x in self._boundaries if x._slot == m)
                bpos[m] = self._boundary_particles(b)[0]
            return bpos[m]

        ff = ParticlesContacts(slot, *rows_of(*self.local_contacts(False), True), pos[sel], lambda m: pos[parts[int(m)]], self.h())
        fb = ParticlesContacts(slot, *rows_of(*self.local_contacts(True), False), pos[sel], positions_of_boundary, self.h())
        view = FluidView(pos[sel], lv["velocities"][sel], lv["volumes"][sel], f.density0)
        view.ids, view.is_ghost = lv["ids"][sel], lv["is_ghost"][sel]
        force.solve(TimestepView(dt, inv_dt), self.h(), ff, fb, view, list(self._boundaries), lv["densities"][sel])
        acc = np.zeros((len(slots), 3), F32)
        acc[sel] = np.asarray(view.accelerations, F32)
        L.check(self._L.salva_hip_force_add_local_accelerations(self._h, _fp(acc)))

    # ---- checkpoint / restart (include/salva_hip.h "Checkpoint / restart")
    def checkpoint(self) -> dict:
        """Everything `step` carries over to the next call, as numpy arrays (np.savez-able): fluid positions / velocities /
        volumes, the solver's velocity_changes, IISPH pressures, boundary particles, and the TimestepManager's dt / inv_dt."""
        self.sync_to_device()
        dt, inv_dt = C.c_float(0), C.c_float(0)
        L.check(self._L.salva_hip_get_timestep(self._h, C.byref(dt), C.byref(inv_dt)))
        st = {"timestep": np.array([dt.value, inv_dt.value], F32), "nfluids": np.array(len(self._fluids)),
              "nboundaries": np.array(len(self._boundaries))}
        for k, f in enumerate(self._fluids):
            st[f"fluid{k}_positions"] = f.positions.copy()
            st[f"fluid{k}_velocities"] = f.velocities.copy()
            st[f"fluid{k}_volumes"] = np.asarray(f.volumes, F32).copy()
            st[f"fluid{k}_velocity_changes"] = self.velocity_changes(f)
            st[f"fluid{k}_pressures"] = self.pressures(f)
        for k, b in enumerate(self._boundaries):
            st[f"boundary{k}_positions"] = np.asarray(b.positions, F32).copy()
            st[f"boundary{k}_velocities"] = np.asarray(b.velocities, F32).copy()
        return st

    def restore(self, st: dict):
        """Load a `checkpoint()` into a world built with the same fluids / boundaries (forces, densities, groups come from
        the objects; particle counts may differ from the checkpoint's).  The run continues with the same contacts and
        iteration counts, states equal up to f32 summation order (cells are re-sorted from host order)."""
        if int(st["nfluids"]) != len(self._fluids) or int(st["nboundaries"]) != len(self._boundaries):
            raise ValueError("the checkpoint was taken from a world with different fluids / boundaries")
        for k, f in enumerate(self._fluids):
            pos = np.ascontiguousarray(st[f"fluid{k}_positions"], F32)
            n = len(pos)
            f._positions = pos.copy()
            f._velocities = np.ascontiguousarray(st[f"fluid{k}_velocities"], F32).copy()
            f._volumes = np.ascontiguousarray(st[f"fluid{k}_volumes"], F32).copy()
            f._accelerations = np.zeros((n, 3), F32)
            f._deleted = np.zeros(n, bool)
            f._pending_dv = np.concatenate([np.ascontiguousarray(st[f"fluid{k}_velocity_changes"], F32).reshape(n, 3),
                                            np.ascontiguousarray(st[f"fluid{k}_pressures"], F32).reshape(n, 1)], axis=1)
            f._resized, f._dirty, f._device_newer, f._maybe_deleted, f._acc_touched = True, L.DIRTY_ALL, False, False, False
        for k, b in enumerate(self._boundaries):
            if not b._sampled:
                b.positions = st[f"boundary{k}_positions"]
                b.velocities = st[f"boundary{k}_velocities"]
        self.sync_to_device()
        t = np.asarray(st["timestep"], F32)
        L.check(self._L.salva_hip_set_timestep(self._h, float(t[0]), float(t[1])))

    # ---- opt-in CFL sub-stepping (timestep_manager.rs:36-46 + the clamp left commented out at :90-93)
    def set_cfl_substepping(self, mode: int = 1, cfl_coeff: float = 0.4, min_num_substeps: int = 1, max_num_substeps: int = 10):
        """mode 0: off — one substep per step, the reference as it runs (`compute_substep` returns the whole step).  mode 1: the
        reference's commented clamp, literally (the last substep may overshoot the step).  mode 2: the same, cut at the remaining
        time.  The defaults are `TimestepManager::new`'s (timestep_manager.rs:23-34)."""
        L.check(self._L.salva_hip_set_cfl(self._h, int(mode), float(cfl_coeff), int(min_num_substeps), int(max_num_substeps)))
        self._cfl_mode = int(mode)

    def substeps(self):
        """Substep lengths of the last step (`counters.nsubsteps` of them)."""
        n = int(self._L.salva_hip_get_substeps(self._h, None, 0))  # (the count first: max_num_substeps is the caller's to choose)
        if n < 0:
            L.check(n)
        buf = (C.c_float * max(n, 1))()
        n = int(self._L.salva_hip_get_substeps(self._h, buf, max(n, 1)))
        if n < 0:
            L.check(n)
        return [float(buf[i]) for i in range(n)]

    def step_with_coupling(self, dt: float, gravity, coupling) -> L.StepStats:
        """LiquidWorld::step_with_coupling (liquid_world.rs:67-158) for a `salva_amd.coupling.ColliderCouplingSet`:
        update_boundaries -> the substep -> transmit_forces."""
        self.sync_to_device()
        if getattr(self, "_cfl_mode", 0):
            # CFL sub-stepping: the manager's two calls belong INSIDE the substep loop (liquid_world.rs:94-103, :146) — each substep's
            # impulse reaches the bodies before the next substep samples their velocities.  The library calls back at both points.
            return self._step_with_coupling_callback(dt, gravity, coupling)
        coupling.update_boundaries(self)
        try:
            st = self.step(dt, gravity)
        finally:
            # an exception inside a host-shape callback was parked by its thunk (ctypes cannot propagate it): it is the cause,
            # whatever the library made of the NaN box it was handed instead
            if hasattr(coupling, "raise_pending"):
                coupling.raise_pending()
        coupling.transmit_forces(self, dt)
        return st

    def _step_with_coupling_callback(self, dt: float, gravity, coupling) -> L.StepStats:
        errors = []

        def callback(_user, _world, phase, sub_dt):
            try:
                if phase == 0:
                    coupling.update_boundaries(self)
                else:
                    coupling.transmit_forces(self, float(sub_dt))
                return 0
            except BaseException as e:  # noqa: BLE001 - ctypes cannot propagate it: park it, fail the step
                errors.append(e)
                return 1

        cb = L.COUPLING_CALLBACK(callback)
        L.check(self._L.salva_hip_set_coupling_callback(self._h, cb, None))
        try:
            st = self.step(dt, gravity)
        except Exception:
            if errors:
                raise errors[0]
            raise
        finally:
            self._L.salva_hip_set_coupling_callback(self._h, L.COUPLING_CALLBACK(), None)
            if hasattr(coupling, "raise_pending"):
                coupling.raise_pending()
        return st

    # ---- multi-GPU (no counterpart in the reference; include/salva_hip.h "multi-GPU")
    def set_domain(self, comm, cell_lo: int, cell_hi: int, gid_offset: int = 0):
        """Make this world one x-slab [cell_lo, cell_hi] of a decomposed domain.  Call after adding this rank's fluids
        and boundaries and before the first step; afterwards particles are read back with `owned()`."""
        self.sync_to_device()
        L.check(self._L.salva_hip_set_domain(self._h, comm._h, int(cell_lo), int(cell_hi), int(gid_offset)))
        self._comm = comm

    def dist_timing(self):
        """dict of what the exchanges of the last step cost on this rank (salva_hip_get_dist_timing; `world.counters.enable()` first)."""
        out = (C.c_double * 4)()
        L.check(self._L.salva_hip_get_dist_timing(self._h, out))
        return {"refresh_ms": out[0], "refreshes": int(out[1]), "test_ms": out[2], "tests": int(out[3])}

    def delete_owned(self, gids) -> int:
        """Collective (every rank, between the same two steps): remove the particles of `gids` this rank owns from the next
        step on (salva_hip_delete_owned); returns how many particles the rank still owns."""
        g = np.ascontiguousarray(gids, np.uint32).ravel()
        m = int(self._L.salva_hip_delete_owned(self._h, len(g), g.ctypes.data_as(C.POINTER(C.c_uint32))))
        if m < 0:
            L.check(m)
        return m

    def add_owned(self, fluid, positions, velocities=None):
        """Collective (every rank, between the same two steps; an empty array where there is nothing to add): append particles
        to this rank of a running decomposed world (salva_hip_add_particles); they get the next free global ids."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        vel = None if velocities is None else np.ascontiguousarray(velocities, np.float32).reshape(-1, 3)
        L.check(self._L.salva_hip_add_particles(self._h, fluid._slot, len(pos), _fp(pos) if len(pos) else None, _fp(vel) if vel is not None and len(pos) else None))

    def rebalance(self):
        """Collective re-cut of the slabs for equal particle counts (salva_hip_rebalance): returns this rank's new (lo, hi)."""
        lo, hi = C.c_int32(0), C.c_int32(0)
        L.check(self._L.salva_hip_rebalance(self._h, C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def owned(self):
        """(gids, positions, velocities, fluid slots) of the particles this rank owns after the last step, sorted by gid."""
        u32p = C.POINTER(C.c_uint32)
        cap = max(int(self.last_stats.nparticles) + 1024, 1024)
        while True:
            gid = np.zeros(cap, np.uint32)
            slot = np.zeros(cap, np.uint32)
            pos = np.zeros((cap, 3), F32)
            vel = np.zeros((cap, 3), F32)
            m = int(self._L.salva_hip_get_owned(self._h, cap, gid.ctypes.data_as(u32p), _fp(pos), _fp(vel),
                                                slot.ctypes.data_as(u32p)))
            if m < 0:
                L.check(m)
            if m <= cap:
                break
            cap = m
        o = np.argsort(gid[:m], kind="stable")
        return gid[:m][o], pos[:m][o], vel[:m][o], slot[:m][o]

    # ---- solver scratch (private in the reference; exposed for the parity tests)
    def fluid_field(self, f: Fluid, field: int) -> np.ndarray:
        n = f.num_particles()
        vec = field in (L.FIELD_VELOCITY_CHANGE, L.FIELD_ACCELERATION)
        out = np.zeros((n, 3) if vec else n, F32)
        if n:
            L.check(self._L.salva_hip_get_fluid_field(self._h, f._slot, field, _fp(out)))
        return out

    def densities(self, f: Fluid) -> np.ndarray:
        return self.fluid_field(f, L.FIELD_DENSITY)

    def alphas(self, f: Fluid) -> np.ndarray:
        return self.fluid_field(f, L.FIELD_ALPHA)

    def velocity_changes(self, f: Fluid) -> np.ndarray:
        return self.fluid_field(f, L.FIELD_VELOCITY_CHANGE)

    def pressures(self, f: Fluid) -> np.ndarray:
        return self.fluid_field(f, L.FIELD_PRESSURE)

    def contact_counts(self, f: Fluid, boundary_contacts: bool = False) -> np.ndarray:
        fld = L.FIELD_NUM_BOUNDARY_CONTACTS if boundary_contacts else L.FIELD_NUM_FLUID_CONTACTS
        return self.fluid_field(f, fld).astype(np.uint32)

    def fluid_contacts(self, f: Fluid, boundary_contacts: bool = False):
        """(offsets, j_model, j): the fluid-fluid (or fluid-boundary) contacts of the last step in host order, CSR —
        `contact_manager.fluid_fluid_contacts[handle].particle_contacts(i)` of the reference (geometry/contacts.rs:57-131)."""
        n = f.num_particles()
        offsets = np.zeros(n + 1, np.uint64)
        u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        total = int(self._L.salva_hip_get_fluid_contacts(self._h, f._slot, int(boundary_contacts), offsets.ctypes.data_as(u64p), None, None, 0))
        if total < 0:
            L.check(total)
        jm, j = np.zeros(total, np.uint32), np.zeros(total, np.uint32)
        if total:
            got = int(self._L.salva_hip_get_fluid_contacts(self._h, f._slot, int(boundary_contacts), offsets.ctypes.data_as(u64p),
                                                           jm.ctypes.data_as(u32p), j.ctypes.data_as(u32p), total))
            if got < 0:
                L.check(got)
        return offsets, jm, j

    def particles_intersecting_aabb(self, mins, maxs):
        """liquid_world.rs:210-243: list of ("fluid" | "boundary", handle, particle index) whose distance to the box is
        below the particle radius (current positions)."""
        self.sync_to_device(apply_removal=False)  # a query is not a step: pending deletions stay pending
        lo = (C.c_float * 3)(*[float(x) for x in mins])
        hi = (C.c_float * 3)(*[float(x) for x in maxs])
        u32p = C.POINTER(C.c_uint32)
        cap = 1024
        while True:
            k, s, i = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            total = int(self._L.salva_hip_particles_intersecting_aabb(self._h, lo, hi, cap, k.ctypes.data_as(u32p),
                                                                       s.ctypes.data_as(u32p), i.ctypes.data_as(u32p)))
            if total < 0:
                L.check(total)
            if total <= cap:
                break
            cap = total
        out = []
        for q in range(total):
            owner = self._boundaries._items[int(s[q])] if k[q] else self._fluids._items[int(s[q])]
            out.append(("boundary" if k[q] else "fluid", owner, int(i[q])))
        return out

    def particles_intersecting_shape(self, translation, rotation, shape):
        """liquid_world.rs:245-280 for `shape` = ("ball", radius), ("cuboid", (hx, hy, hz)), ("capsule", half_height, radius) or
        ("cylinder", half_height, radius) (the last two along their local y axis) posed by the isometry
        (translation, unit quaternion (i, j, k, w)): particles within the particle radius of the solid shape."""
        self.sync_to_device(apply_removal=False)
        from .coupling import make_shape

        sh = make_shape(shape)
        t = (C.c_float * 3)(*[float(x) for x in translation])
        q = (C.c_float * 4)(*[float(x) for x in rotation])
        u32p = C.POINTER(C.c_uint32)
        cap = 1024
        while True:
            k, s, i = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            total = int(self._L.salva_hip_particles_intersecting_shape(self._h, t, q, C.byref(sh), cap, k.ctypes.data_as(u32p),
                                                                        s.ctypes.data_as(u32p), i.ctypes.data_as(u32p)))
            if total < 0:
                L.check(total)
            if total <= cap:
                break
            cap = total
        out = []
        for j in range(total):
            owner = self._boundaries._items[int(s[j])] if k[j] else self._fluids._items[int(s[j])]
            out.append(("boundary" if k[j] else "fluid", owner, int(i[j])))
        return out

    def particles_intersecting_host_shape(self, compute_aabb, distance_to_point):
        """liquid_world.rs:245-280 for any shape the host can describe (the reference's query is generic over parry's `Shape`):
        `compute_aabb() -> (mins, maxs)` is `shape.compute_aabb(pos)`, `distance_to_point(points (n, 3)) -> (n,)` is
        `shape.distance_to_point(pos, &pt, true)` per point.  Same output as `particles_intersecting_shape`."""
        self.sync_to_device(apply_removal=False)
        err = []

        def aabb_cb(_user, mins, maxs):
            try:
                lo, hi = compute_aabb()
                for a in range(3):
                    mins[a], maxs[a] = float(lo[a]), float(hi[a])
            except BaseException as e:  # noqa: BLE001 - ctypes cannot propagate it: parked, re-raised below
                err.append(e)
                for a in range(3):
                    mins[a] = maxs[a] = float("nan")

        def dist_cb(_user, n, pts, out):
            try:
                d = np.asarray(distance_to_point(np.ctypeslib.as_array(pts, shape=(n, 3)).copy()), F32).reshape(n)
                np.ctypeslib.as_array(out, shape=(n,))[:] = d
            except BaseException as e:  # noqa: BLE001
                err.append(e)
                np.ctypeslib.as_array(out, shape=(n,))[:] = np.inf

        thunks = (L.HOST_AABB_FN(aabb_cb), L.HOST_DISTANCE_FN(dist_cb))
        sh = L.HostQueryShape(thunks[0], thunks[1], None)
        u32p = C.POINTER(C.c_uint32)
        cap = 1024
        while True:
            k, s, i = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            total = int(self._L.salva_hip_particles_intersecting_host_shape(self._h, C.byref(sh), cap, k.ctypes.data_as(u32p),
                                                                             s.ctypes.data_as(u32p), i.ctypes.data_as(u32p)))
            if err:
                raise err[0]
            if total < 0:
                L.check(total)
            if total <= cap:
                break
            cap = total
        out = []
        for j in range(total):
            owner = self._boundaries._items[int(s[j])] if k[j] else self._fluids._items[int(s[j])]
            out.append(("boundary" if k[j] else "fluid", owner, int(i[j])))
        return out

    def device_bytes(self) -> int:
        return int(self._L.salva_hip_device_bytes(self._h))

    def time_variant(self, variant: int, param: int = 0, reps: int = 20):
        """Kernel-development builds only (SALVA_HIP_LIB_VARIANT=diag): (microseconds per launch, checksum of the outputs) of
        execution variant `variant` of k_pred_density."""
        if not hasattr(self._L, "salva_hip_time_variant"):
            raise RuntimeError("salva_hip_time_variant exists only in libsalva_hip_diag.so (make -C salva_amd/csrc VARIANT=diag)")
        cs = C.c_uint64(0)
        us = float(self._L.salva_hip_time_variant(self._h, variant, param, reps, C.byref(cs)))
        if us < 0:
            L.check(int(us))
        return us, int(cs.value)

    def time_kernel(self, kernel: int, reps: int = 20) -> float:
        """Average launch duration (us) of 0 k_pred_density, 1 k_divergence, 2 k_iisph_next_pressure, 3 k_iisph_dij_pj."""
        us = float(self._L.salva_hip_time_kernel(self._h, kernel, reps))
        if us < 0:
            L.check(int(us))
        return us

    def time_pred_density(self, reps: int = 20) -> float:
        """Average k_pred_density launch duration in microseconds (HIP events on the world's stream)."""
        us = float(self._L.salva_hip_time_pred_density(self._h, reps))
        if us < 0:
            L.check(int(us))
        return us
