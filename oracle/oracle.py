"""ctypes front end of the CPU oracle (`oracle/salva_oracle.cpp`).

TEST INFRASTRUCTURE ONLY: importable from `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py`, never from `salva_amd/`.  PARITY UNPINNED (see the header of salva_oracle.cpp).

The class mirrors the call sequence of the reference (`LiquidWorld::new / add_fluid / add_boundary / step`,
/root/reference/src/liquid_world.rs:39-171) so the parity tests can drive oracle and HIP path alike.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsalva_oracle.so")

DFSPH, IISPH = 0, 1
FORCE_XSPH, FORCE_ARTIFICIAL, FORCE_AKINCI2013, FORCE_DFSPH_VISCOSITY, FORCE_HE2014, FORCE_WCSPH_TENSION, FORCE_CUSTOM = 1, 2, 3, 4, 5, 6, 7


class Stats(C.Structure):
    _fields_ = [
        ("n_div_iters", C.c_int32),
        ("n_press_iters", C.c_int32),
        ("div_error", C.c_double),
        ("density_error", C.c_double),
        ("ncontacts", C.c_uint64),
        ("t_grid_ms", C.c_double),
        ("t_contacts_ms", C.c_double),
        ("t_kernels_ms", C.c_double),
        ("t_solver_ms", C.c_double),
        ("t_total_ms", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed recipe (oracle/Makefile)."""
    src = os.path.join(_HERE, "salva_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsalva_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_libs = {}


def lib(native: bool = False):
    """`native`: the -O3 -march=native timing build (oracle/Makefile), compiled on this machine on first use."""
    global _libs
    if native not in _libs:
        path = _LIB_PATH
        if native:
            path = os.path.join(_HERE, "libsalva_oracle_native.so")
            # always rebuilt: -march=native belongs to the host it was compiled on, and the file travels with the tree
            subprocess.check_call(["make", "-C", _HERE, "-B", "libsalva_oracle_native.so"], stdout=subprocess.DEVNULL)
        else:
            build()
        L = C.CDLL(path)
        vp, u64, u32, i32, f32, f64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_double
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.so_create.restype = vp
        L.so_create.argtypes = [i32, f32, f32, i32, i32]
        L.so_destroy.argtypes = [vp]
        L.so_set_shuffle_seed.argtypes = [vp, u64]
        L.so_set_threads.argtypes = [vp, i32]
        L.so_set_kernels.argtypes = [vp, i32, i32]
        L.so_kernel_scalar.argtypes = [i32, i32, C.c_float, C.c_float]
        L.so_kernel_scalar.restype = C.c_float
        L.so_kernel_scalar_f64.argtypes = [i32, i32, C.c_double, C.c_double]
        L.so_kernel_scalar_f64.restype = C.c_double
        L.so_set_solver_params.argtypes = [vp, i32, i32, f32, i32, i32, f32]
        L.so_h.restype = f64
        L.so_h.argtypes = [vp]
        L.so_set_timestep.argtypes = [vp, f32, f32]
        L.so_set_cfl.argtypes = [vp, i32, f32, i32, i32]
        L.so_get_substeps.restype = i32
        L.so_get_substeps.argtypes = [vp, C.POINTER(C.c_double), i32]
        L.so_add_fluid.argtypes = [vp, u64, fp, fp, f32, u32, u32]
        L.so_add_boundary.argtypes = [vp, u64, fp, fp, u32, u32, i32]
        L.so_add_force.argtypes = [vp, i32, i32, fp, i32]
        L.so_set_fluid_velocities.argtypes = [vp, i32, fp]
        L.so_set_fluid_volumes.argtypes = [vp, i32, fp]
        L.so_step.argtypes = [vp, f32, f32, f32, f32, C.POINTER(Stats)]
        L.so_fluid_len.restype = u64
        L.so_fluid_len.argtypes = [vp, i32]
        L.so_boundary_len.restype = u64
        L.so_boundary_len.argtypes = [vp, i32]
        L.so_get_fluid_vec.argtypes = [vp, i32, i32, dp]
        L.so_get_fluid_scalar.argtypes = [vp, i32, i32, dp]
        L.so_get_contact_counts.argtypes = [vp, i32, i32, C.POINTER(C.c_uint32)]
        L.so_get_contacts_of.restype = u64
        L.so_get_contacts_of.argtypes = [vp, i32, i32, u64, C.POINTER(C.c_uint64), u64]
        L.so_get_viscosity_stats.argtypes = [vp, i32, i32, C.POINTER(C.c_int), dp]
        L.so_get_viscosity_betas.argtypes = [vp, i32, i32, dp]
        L.so_set_boundary_sampling.argtypes = [vp, i32, u64, C.POINTER(C.c_float)]
        L.so_update_boundary_pose.argtypes = [vp, i32, dp, i32, i32]
        L.so_set_boundary_dynamic_sampling.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
        L.so_get_boundary_sources.argtypes = [vp, i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.so_get_boundary_wrench.argtypes = [vp, i32, dp, dp, dp]
        L.so_add_particles.argtypes = [vp, i32, u64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.so_delete_particle.argtypes = [vp, i32, u64]
        L.so_remove_fluid.argtypes = [vp, i32]
        L.so_remove_boundary.argtypes = [vp, i32]
        L.so_set_boundary_particles.argtypes = [vp, i32, u64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.so_set_force_callback.argtypes = [vp, FORCE_CB, vp]
        L.so_set_substep_callback.argtypes = [vp, SUBSTEP_CB, vp]
        L.so_num_forces.argtypes = [vp, i32]
        L.so_num_forces.restype = i32
        L.so_reference_would_panic.restype = i32
        L.so_reference_would_panic.argtypes = [vp]
        L.so_test_lu6.argtypes = [dp, dp, dp]
        L.so_get_boundary_vec.argtypes = [vp, i32, i32, dp]
        L.so_get_boundary_volumes.argtypes = [vp, i32, dp]
        L.so_clear_boundary_forces.argtypes = [vp, i32]
        for name in ("so_kernel_w", "so_kernel_dw", "so_cohesion_kernel", "so_adhesion_kernel"):
            getattr(L, name).restype = f32
            getattr(L, name).argtypes = [f32, f32]
        for name in ("so_kernel_w_f64", "so_kernel_dw_f64"):
            getattr(L, name).restype = f64
            getattr(L, name).argtypes = [f64, f64]
        L.so_max_threads.restype = i32
        _libs[native] = L
    return _libs[native]


SUBSTEP_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double)
FORCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                       C.POINTER(C.c_double), C.POINTER(C.c_double))


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        assert a.ndim == 2 and a.shape[1] == cols, a.shape
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleWorld:
    """CPU restatement of `salva3d::LiquidWorld` (f32 like the reference, or f64 for the noise floor)."""

    VEC_FIELDS = {"positions": 0, "velocities": 1, "velocity_changes": 2, "accelerations": 3, "dii": 4,
                  "dij_pjl": 5, "normals": 6}
    SCALAR_FIELDS = {"densities": 0, "alphas": 1, "divergences": 2, "predicted_densities": 3, "volumes": 4,
                     "aii": 5, "pressures": 6, "he2014_colors": 7, "he2014_gradcs": 8}

    def __init__(self, particle_radius: float, smoothing_factor: float = 2.0, solver: int = DFSPH,
                 f64: bool = False, threads: int = 1, native: bool = False):
        self._L = lib(native)
        self._h = self._L.so_create(int(f64), particle_radius, smoothing_factor, solver, threads)
        self.f64 = f64
        self.last_stats = Stats()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.so_destroy(self._h)
            self._h = None

    @property
    def h(self) -> float:
        return self._L.so_h(self._h)

    def set_timestep(self, dt: float, inv_dt: float):
        self._L.so_set_timestep(self._h, dt, inv_dt)

    def set_cfl(self, mode: int, cfl_coeff: float = 0.4, min_substeps: int = 1, max_substeps: int = 10):
        """Opt-in CFL sub-stepping: timestep_manager.rs:36-46 (`max_substep`) with the clamp the reference left commented out at
        :90-93.  mode 0 = off (the reference as it runs), 1 = the commented code literally (the last substep may overshoot the
        step), 2 = the same, cut at the end of the step."""
        self._L.so_set_cfl(self._h, mode, cfl_coeff, min_substeps, max_substeps)

    def substeps(self):
        """Substep lengths of the last step."""
        buf = (C.c_double * 64)()
        n = self._L.so_get_substeps(self._h, buf, 64)
        return [buf[i] for i in range(min(n, 64))]

    def set_kernels(self, density: int, gradient: int):
        """The solvers' KernelDensity / KernelGradient type parameters: 0 CubicSpline (default), 1 Poly6, 2 Spiky, 3 Viscosity."""
        self._L.so_set_kernels(self._h, density, gradient)

    def set_threads(self, n: int):
        self._L.so_set_threads(self._h, n)

    def set_shuffle_seed(self, seed: int):
        self._L.so_set_shuffle_seed(self._h, seed)

    def set_solver_params(self, min_pressure_iter=1, max_pressure_iter=50, max_density_error=0.05,
                          min_divergence_iter=1, max_divergence_iter=50, max_divergence_error=0.1):
        self._L.so_set_solver_params(self._h, min_pressure_iter, max_pressure_iter, max_density_error,
                                     min_divergence_iter, max_divergence_iter, max_divergence_error)

    def add_fluid(self, positions, density0=1000.0, velocities=None, memberships=1, filter=0xFFFFFFFF) -> int:
        pos = _f32(positions, 3)
        vel = _f32(velocities, 3) if velocities is not None else None
        return self._L.so_add_fluid(self._h, len(pos), _fp(pos), _fp(vel) if vel is not None else None,
                                    density0, memberships, filter)

    def add_boundary(self, positions, velocities=None, memberships=1, filter=0xFFFFFFFF, wants_forces=False) -> int:
        pos = _f32(positions, 3)
        vel = _f32(velocities, 3) if velocities is not None else None
        return self._L.so_add_boundary(self._h, len(pos), _fp(pos), _fp(vel) if vel is not None else None,
                                       memberships, filter, int(wants_forces))

    # ---- integrations/rapier/fluids_pipeline.rs, StaticSampling arm
    def set_boundary_sampling(self, boundary, local_points):
        pts = _f32(local_points, 3)
        self._L.so_set_boundary_sampling(self._h, boundary, len(pts), _fp(pts))

    # ---- DynamicContactSampling arm (:193-259): shape kind 1 = ball (radius), 2 = cuboid (half extents)
    def set_boundary_dynamic_sampling(self, boundary, kind, params):
        prm = np.zeros(3, np.float32)
        prm[:len(np.atleast_1d(params))] = np.atleast_1d(params)
        self._L.so_set_boundary_dynamic_sampling(self._h, boundary, int(kind), _fp(prm))

    def boundary_sources(self, boundary):
        """(fluid, particle) each point of a dynamically sampled boundary was projected from, in emission order."""
        n = self.boundary_len(boundary)
        f, p = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        u32p = C.POINTER(C.c_uint32)
        self._L.so_get_boundary_sources(self._h, boundary, f.ctypes.data_as(u32p), p.ctypes.data_as(u32p))
        return f, p

    def update_boundary_pose(self, boundary, translation=(0, 0, 0), rotation=(0, 0, 0, 1), linvel=(0, 0, 0), angvel=(0, 0, 0),
                             world_com=(0, 0, 0), has_body=True, is_dynamic=True):
        """update_boundaries (:160-193, 262); values are rounded to f32 first, like the device ABI takes them."""
        pose = np.concatenate([np.asarray(x, np.float32).astype(np.float64) for x in (translation, rotation, linvel, angvel, world_com)])
        self._L.so_update_boundary_pose(self._h, boundary, pose.ctypes.data_as(C.POINTER(C.c_double)), int(has_body), int(is_dynamic))

    def boundary_wrench(self, boundary, point):
        """transmit_forces (:266-287) as (sum f, sum (x - point) x f)."""
        c = np.asarray(point, np.float32).astype(np.float64)
        f, t = np.zeros(3), np.zeros(3)
        dp = C.POINTER(C.c_double)
        self._L.so_get_boundary_wrench(self._h, boundary, c.ctypes.data_as(dp), f.ctypes.data_as(dp), t.ctypes.data_as(dp))
        return f, t

    def add_xsph(self, fluid, fluid_coeff, boundary_coeff):
        p = _f32([fluid_coeff, boundary_coeff])
        self._L.so_add_force(self._h, fluid, FORCE_XSPH, _fp(p), 2)

    def add_artificial_viscosity(self, fluid, fluid_coeff, boundary_coeff, alpha=1.0, beta=0.0, speed_of_sound=10.0):
        p = _f32([fluid_coeff, boundary_coeff, alpha, beta, speed_of_sound])
        self._L.so_add_force(self._h, fluid, FORCE_ARTIFICIAL, _fp(p), 5)

    def add_akinci2013(self, fluid, tension_coeff, adhesion_coeff):
        p = _f32([tension_coeff, adhesion_coeff])
        self._L.so_add_force(self._h, fluid, FORCE_AKINCI2013, _fp(p), 2)

    def add_he2014(self, fluid, fluid_tension_coeff, boundary_tension_coeff):
        """solver::He2014SurfaceTension::new (he2014_surface_tension.rs:21-29)."""
        p = _f32([fluid_tension_coeff, boundary_tension_coeff])
        self._L.so_add_force(self._h, fluid, FORCE_HE2014, _fp(p), 2)

    def add_wcsph_tension(self, fluid, fluid_tension_coeff, boundary_tension_coeff):
        """solver::WCSPHSurfaceTension::new (wcsph_surface_tension.rs:22-28)."""
        p = _f32([fluid_tension_coeff, boundary_tension_coeff])
        self._L.so_add_force(self._h, fluid, FORCE_WCSPH_TENSION, _fp(p), 2)

    def add_custom_force(self, fluid, fn):
        """A user NonPressureForce (nonpressure_force.rs:10-30): fn(world, fluid, positions, velocities, densities,
        accelerations) edits `accelerations` (n x 3 float64 view) in place; contacts via world.contacts_of()."""
        if not hasattr(self, "_custom"):
            self._custom = {}

            def trampoline(_user, f, k, n, pos, vel, dens, acc):
                shape = (n, 3)
                self._custom[(f, k)](self, f, np.ctypeslib.as_array(pos, shape), np.ctypeslib.as_array(vel, shape),
                                     np.ctypeslib.as_array(dens, (n,)), np.ctypeslib.as_array(acc, shape))

            self._cb = FORCE_CB(trampoline)
            self._L.so_set_force_callback(self._h, self._cb, None)
        p = _f32([0.0])
        k = self._L.so_num_forces(self._h, fluid)
        self._L.so_add_force(self._h, fluid, FORCE_CUSTOM, _fp(p), 1)
        self._custom[(fluid, k)] = fn

    def set_coupling_callback(self, fn):
        """The `CouplingManager` of `step_with_coupling` (coupling_manager.rs:9-28): fn(phase, dt) is called inside every substep —
        phase 0 where the reference calls `coupling.update_boundaries` (liquid_world.rs:94-103; dt = the last substep's), phase 1
        where it calls `coupling.transmit_forces` (:146; dt = this substep's).  None unregisters."""
        if fn is None:
            self._substep_cb = SUBSTEP_CB()
        else:
            self._substep_cb = SUBSTEP_CB(lambda _user, phase, dt: fn(int(phase), float(dt)))
        self._L.so_set_substep_callback(self._h, self._substep_cb, None)

    def reference_would_panic(self) -> bool:
        return bool(self._L.so_reference_would_panic(self._h))

    def add_dfsph_viscosity(self, fluid, viscosity_coefficient, min_iter=1, max_iter=50, max_error=0.01):
        """solver::DFSPHViscosity::new(coefficient) with its pub tuning fields (dfsph_viscosity.rs:89-125)."""
        p = _f32([viscosity_coefficient, min_iter, max_iter, max_error])
        self._L.so_add_force(self._h, fluid, FORCE_DFSPH_VISCOSITY, _fp(p), 4)

    def viscosity_stats(self, fluid, force_index=0):
        it, err = C.c_int(0), C.c_double(0)
        self._L.so_get_viscosity_stats(self._h, fluid, force_index, C.byref(it), C.byref(err))
        return it.value, err.value

    def viscosity_betas(self, fluid, force_index=0) -> np.ndarray:
        out = np.zeros((self.fluid_len(fluid), 6, 6), dtype=np.float64)
        self._L.so_get_viscosity_betas(self._h, fluid, force_index, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def add_particles(self, fluid, positions, velocities=None):
        """Fluid::add_particles (fluid.rs:126-150)."""
        pos = _f32(positions, 3)
        vel = _f32(velocities, 3) if velocities is not None else None
        self._L.so_add_particles(self._h, fluid, len(pos), _fp(pos), _fp(vel) if vel is not None else None)

    def delete_particle_at_next_timestep(self, fluid, i):
        """fluid.rs:71-76; the removal happens at the top of the next step (liquid_world.rs:78-82)."""
        self._L.so_delete_particle(self._h, fluid, int(i))

    def remove_fluid(self, fluid):
        """LiquidWorld::remove_fluid (liquid_world.rs:171-173): swap-remove; the solver's buffers stay positional."""
        self._L.so_remove_fluid(self._h, fluid)
        if hasattr(self, "_custom"):
            self._custom = {}

    def remove_boundary(self, boundary):
        """LiquidWorld::remove_boundary (liquid_world.rs:176-178): swap-remove."""
        self._L.so_remove_boundary(self._h, boundary)

    def set_boundary_particles(self, boundary, positions, velocities=None):
        pos = _f32(positions, 3)
        vel = _f32(velocities, 3) if velocities is not None else None
        self._L.so_set_boundary_particles(self._h, boundary, len(pos), _fp(pos), _fp(vel) if vel is not None else None)

    def set_fluid_velocities(self, fluid, velocities):
        v = _f32(velocities, 3)
        assert len(v) == self.fluid_len(fluid)
        self._L.so_set_fluid_velocities(self._h, fluid, _fp(v))

    def set_fluid_volumes(self, fluid, volumes):
        v = _f32(volumes)
        assert len(v) == self.fluid_len(fluid)
        self._L.so_set_fluid_volumes(self._h, fluid, _fp(v))

    def step(self, dt, gravity=(0.0, -9.81, 0.0)) -> Stats:
        self._L.so_step(self._h, dt, gravity[0], gravity[1], gravity[2], C.byref(self.last_stats))
        return self.last_stats

    def fluid_len(self, fluid) -> int:
        return self._L.so_fluid_len(self._h, fluid)

    def boundary_len(self, b) -> int:
        return self._L.so_boundary_len(self._h, b)

    def fluid_vec(self, fluid, field) -> np.ndarray:
        out = np.zeros((self.fluid_len(fluid), 3), dtype=np.float64)
        self._L.so_get_fluid_vec(self._h, fluid, self.VEC_FIELDS[field], out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def fluid_scalar(self, fluid, field) -> np.ndarray:
        out = np.zeros(self.fluid_len(fluid), dtype=np.float64)
        self._L.so_get_fluid_scalar(self._h, fluid, self.SCALAR_FIELDS[field], out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def contact_counts(self, fluid, boundary_contacts=False) -> np.ndarray:
        out = np.zeros(self.fluid_len(fluid), dtype=np.uint32)
        self._L.so_get_contact_counts(self._h, fluid, int(boundary_contacts), out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def contacts_of(self, fluid, i, boundary_contacts=False):
        """Sorted list of (j_model, j) of particle i's contacts."""
        buf = np.zeros(4096, dtype=np.uint64)
        n = self._L.so_get_contacts_of(self._h, fluid, int(boundary_contacts), i,
                                       buf.ctypes.data_as(C.POINTER(C.c_uint64)), len(buf))
        return [(int(k) >> 32, int(k) & 0xFFFFFFFF) for k in buf[:n]]

    def boundary_vec(self, b, field) -> np.ndarray:
        out = np.zeros((self.boundary_len(b), 3), dtype=np.float64)
        self._L.so_get_boundary_vec(self._h, b, {"positions": 0, "velocities": 1, "forces": 2}[field],
                                    out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def boundary_volumes(self, b) -> np.ndarray:
        out = np.zeros(self.boundary_len(b), dtype=np.float64)
        self._L.so_get_boundary_volumes(self._h, b, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def clear_boundary_forces(self, b):
        self._L.so_clear_boundary_forces(self._h, b)


def kernel_w(r, h, f64=False):
    return lib().so_kernel_w_f64(r, h) if f64 else lib().so_kernel_w(r, h)


def kernel_dw(r, h, f64=False):
    return lib().so_kernel_dw_f64(r, h) if f64 else lib().so_kernel_dw(r, h)
