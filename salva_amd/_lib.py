"""ctypes binding of `libsalva_hip.so` (C ABI: include/salva_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises — there is no Python,
PyTorch or CPU fallback for the fluid step.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsalva_hip.so")
if os.environ.get("SALVA_HIP_LIB_VARIANT"):  # kernel experiments only (tools/variant_probe.py): e.g. "t3" -> libsalva_hip_t3.so
    LIB_PATH = os.path.join(CSRC, "libsalva_hip_%s.so" % os.environ["SALVA_HIP_LIB_VARIANT"])

OK, E_HIP, E_INVALID, E_NUMERIC, E_CAPACITY = 0, -1, -2, -3, -4
SOLVER_DFSPH, SOLVER_IISPH = 0, 1
FORCE_XSPH, FORCE_ARTIFICIAL, FORCE_AKINCI2013, FORCE_DFSPH_VISCOSITY, FORCE_HE2014, FORCE_WCSPH_TENSION, FORCE_CUSTOM = 1, 2, 3, 4, 5, 6, 7
DIRTY_POSITIONS, DIRTY_VELOCITIES, DIRTY_VOLUMES, DIRTY_ACCELERATIONS, DIRTY_ALL = 1, 2, 4, 8, 15
(FIELD_DENSITY, FIELD_ALPHA, FIELD_NUM_FLUID_CONTACTS, FIELD_NUM_BOUNDARY_CONTACTS, FIELD_VELOCITY_CHANGE,
 FIELD_PRESSURE, FIELD_VOLUME, FIELD_ACCELERATION) = range(8)

# every symbol include/salva_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = [
    "salva_hip_default_params", "salva_hip_create", "salva_hip_destroy", "salva_hip_h", "salva_hip_set_fluid",
    "salva_hip_set_fluid_forces", "salva_hip_remove_fluid", "salva_hip_set_boundary", "salva_hip_remove_boundary",
    "salva_hip_num_fluids", "salva_hip_num_boundaries", "salva_hip_fluid_len", "salva_hip_boundary_len",
    "salva_hip_step", "salva_hip_get_fluid", "salva_hip_get_fluid_field", "salva_hip_get_boundary",
    "salva_hip_clear_boundary_forces", "salva_hip_device_bytes", "salva_hip_time_pred_density",
    "salva_hip_last_error", "salva_hip_version", "salva_hip_comm_rccl_unique_id", "salva_hip_comm_rccl_create",
    "salva_hip_comm_loopback_create", "salva_hip_comm_destroy", "salva_hip_set_domain", "salva_hip_get_owned",
    "salva_hip_get_force_stats", "salva_hip_get_fluid_contacts", "salva_hip_add_particles", "salva_hip_delete_particles",
    "salva_hip_particles_intersecting_aabb", "salva_hip_set_boundary_sampling", "salva_hip_update_boundary_pose",
    "salva_hip_get_boundary_particles", "salva_hip_get_boundary_wrench", "salva_hip_set_force_callback",
    "salva_hip_force_get_state", "salva_hip_force_add_accelerations", "salva_hip_set_fluid_field", "salva_hip_get_timestep",
    "salva_hip_set_timestep", "salva_hip_get_counters", "salva_hip_time_kernel", "salva_hip_particles_intersecting_shape", "salva_hip_rebalance",
    "salva_hip_set_boundary_dynamic_sampling", "salva_hip_get_boundary_sources", "salva_hip_set_boundary_dynamic_sampling_host",
    "salva_hip_delete_owned", "salva_hip_enable_counters", "salva_hip_comm_peer_begin", "salva_hip_comm_peer_connect",
    "salva_hip_comm_peer_abort", "salva_hip_comm_selftest", "salva_hip_comm_time", "salva_hip_clear_boundary_sampling", "salva_hip_get_fluid_async", "salva_hip_wait_download",
    "salva_hip_host_alloc", "salva_hip_host_free", "salva_hip_host_register", "salva_hip_host_unregister",
    "salva_hip_set_cfl", "salva_hip_get_substeps", "salva_hip_particles_intersecting_host_shape",
    "salva_hip_set_coupling_callback",
    "salva_hip_get_dist_timing", "salva_hip_local_len", "salva_hip_get_local", "salva_hip_get_local_contacts", "salva_hip_force_add_local_accelerations",
]


class Params(C.Structure):
    _fields_ = [
        ("particle_radius", C.c_float),
        ("smoothing_factor", C.c_float),
        ("solver", C.c_int32),
        ("min_pressure_iter", C.c_int32),
        ("max_pressure_iter", C.c_int32),
        ("max_density_error", C.c_float),
        ("min_divergence_iter", C.c_int32),
        ("max_divergence_iter", C.c_int32),
        ("max_divergence_error", C.c_float),
        ("device", C.c_int32),
        ("enable_timers", C.c_int32),
        ("kernel_density", C.c_int32),
        ("kernel_gradient", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


class ForceDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", C.c_float * 7)]


# SalvaHipForceCallback (include/salva_hip.h)
FORCE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float)
# SalvaHipCouplingCallback: (user, world, phase, dt)
COUPLING_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_float)


class RigidPose(C.Structure):
    """SalvaHipRigidPose (include/salva_hip.h)."""
    _fields_ = [("translation", C.c_float * 3), ("rotation", C.c_float * 4), ("linvel", C.c_float * 3),
                ("angvel", C.c_float * 3), ("world_com", C.c_float * 3), ("has_body", C.c_int32), ("is_dynamic", C.c_int32)]


class StepStats(C.Structure):
    _fields_ = [
        ("n_divergence_iters", C.c_int32),
        ("n_pressure_iters", C.c_int32),
        ("divergence_error", C.c_float),
        ("density_error", C.c_float),
        ("ncontacts", C.c_uint64),
        ("nparticles", C.c_uint64),
        ("grid_ms", C.c_float),
        ("solver_ms", C.c_float),
        ("step_ms", C.c_float),
        ("reserved", C.c_float * 5),
    ]


class _StagesCounters(C.Structure):
    _fields_ = [("collision_detection_time", C.c_double), ("solver_time", C.c_double)]


class _CollisionDetectionCounters(C.Structure):
    _fields_ = [("ncontacts", C.c_uint64), ("boundary_update_time", C.c_double), ("grid_insertion_time", C.c_double),
                ("neighborhood_search_time", C.c_double), ("contact_sorting_time", C.c_double)]


class _SolverCounters(C.Structure):
    _fields_ = [("non_pressure_resolution_time", C.c_double), ("pressure_resolution_time", C.c_double)]


class CountersStruct(C.Structure):
    """SalvaHipCounters (include/salva_hip.h) = the reference's Counters tree (counters/mod.rs:17-30)."""
    _fields_ = [("nsubsteps", C.c_uint64), ("step_time", C.c_double), ("custom", C.c_double), ("stages", _StagesCounters),
                ("cd", _CollisionDetectionCounters), ("solver", _SolverCounters), ("n_divergence_iters", C.c_int32),
                ("n_pressure_iters", C.c_int32), ("speculative_passes", C.c_uint64), ("discarded_passes", C.c_uint64),
                ("chained_passes", C.c_uint64), ("chain_breaks", C.c_uint64), ("pregrid_adopted", C.c_uint64), ("pregrid_dropped", C.c_uint64),
                ("light_class_passes", C.c_uint64), ("sparse_class_passes", C.c_uint64)]


class Shape(C.Structure):
    """SalvaHipShape (include/salva_hip.h)."""
    _fields_ = [("kind", C.c_int32), ("params", C.c_float * 3)]


KERNEL_CUBIC_SPLINE, KERNEL_POLY6, KERNEL_SPIKY, KERNEL_VISCOSITY = 0, 1, 2, 3
SHAPE_BALL, SHAPE_CUBOID, SHAPE_CAPSULE, SHAPE_CYLINDER = 1, 2, 3, 4

HOST_AABB_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float))
HOST_PROJECT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8))


HOST_DISTANCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float))


class HostQueryShape(C.Structure):
    """SalvaHipHostQueryShape (include/salva_hip.h): a query shape whose geometry stays with the host."""
    _fields_ = [("aabb", HOST_AABB_FN), ("distance", HOST_DISTANCE_FN), ("user", C.c_void_p)]


class HostShape(C.Structure):
    """SalvaHipHostShape (include/salva_hip.h): a collider whose geometry stays with the host."""
    _fields_ = [("aabb", HOST_AABB_FN), ("project", HOST_PROJECT_FN), ("user", C.c_void_p)]


class SalvaHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"salva_hip error {code}: {message}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 with the committed Makefile (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(salva_amd has no fallback path)")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
    fp = C.POINTER(C.c_float)
    L.salva_hip_default_params.argtypes = [C.POINTER(Params)]
    L.salva_hip_default_params.restype = None
    L.salva_hip_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.salva_hip_destroy.argtypes = [vp]
    L.salva_hip_destroy.restype = None
    L.salva_hip_h.argtypes = [vp]
    L.salva_hip_h.restype = f32
    L.salva_hip_set_fluid.argtypes = [vp, u32, u64, fp, fp, fp, fp, fp, f32, u32, u32, u32]
    L.salva_hip_set_fluid_forces.argtypes = [vp, u32, C.POINTER(ForceDesc), u32]
    L.salva_hip_remove_fluid.argtypes = [vp, u32]
    L.salva_hip_set_boundary.argtypes = [vp, u32, u64, fp, fp, u32, u32, i32]
    L.salva_hip_remove_boundary.argtypes = [vp, u32]
    L.salva_hip_num_fluids.argtypes = [vp]
    L.salva_hip_num_fluids.restype = u32
    L.salva_hip_num_boundaries.argtypes = [vp]
    L.salva_hip_num_boundaries.restype = u32
    L.salva_hip_fluid_len.argtypes = [vp, u32]
    L.salva_hip_fluid_len.restype = u64
    L.salva_hip_boundary_len.argtypes = [vp, u32]
    L.salva_hip_boundary_len.restype = u64
    L.salva_hip_step.argtypes = [vp, f32, fp, C.POINTER(StepStats)]
    L.salva_hip_get_fluid.argtypes = [vp, u32, fp, fp]
    L.salva_hip_get_fluid_field.argtypes = [vp, u32, i32, fp]
    L.salva_hip_get_boundary.argtypes = [vp, u32, fp, fp]
    L.salva_hip_clear_boundary_forces.argtypes = [vp, u32]
    L.salva_hip_set_boundary_sampling.argtypes = [vp, u32, u64, fp, u32, u32]
    L.salva_hip_update_boundary_pose.argtypes = [vp, u32, C.POINTER(RigidPose)]
    L.salva_hip_set_boundary_dynamic_sampling.argtypes = [vp, u32, C.POINTER(Shape), u32, u32]
    L.salva_hip_set_boundary_dynamic_sampling_host.argtypes = [vp, u32, C.POINTER(HostShape), u32, u32]
    L.salva_hip_clear_boundary_sampling.argtypes = [vp, u32]
    L.salva_hip_get_fluid_async.argtypes = [vp, u32, fp, fp]
    L.salva_hip_get_dist_timing.argtypes = [vp, C.POINTER(C.c_double)]
    L.salva_hip_local_len.argtypes = [vp]
    L.salva_hip_local_len.restype = u64
    L.salva_hip_get_local.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint8), fp, fp, fp, fp]
    L.salva_hip_get_local_contacts.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), u64]
    L.salva_hip_get_local_contacts.restype = C.c_int64
    L.salva_hip_force_add_local_accelerations.argtypes = [vp, fp]
    L.salva_hip_wait_download.argtypes = [vp]
    L.salva_hip_host_alloc.argtypes = [vp, C.c_uint64]
    L.salva_hip_host_alloc.restype = C.c_void_p
    L.salva_hip_host_free.argtypes = [C.c_void_p]
    L.salva_hip_host_register.argtypes = [vp, C.c_void_p, C.c_uint64]
    L.salva_hip_host_unregister.argtypes = [C.c_void_p]
    L.salva_hip_get_boundary_sources.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32)]
    L.salva_hip_get_boundary_particles.argtypes = [vp, u32, fp, fp]
    L.salva_hip_get_boundary_wrench.argtypes = [vp, u32, fp, fp, fp]
    L.salva_hip_set_force_callback.argtypes = [vp, FORCE_CALLBACK, vp]
    L.salva_hip_set_coupling_callback.argtypes = [vp, COUPLING_CALLBACK, vp]
    L.salva_hip_force_get_state.argtypes = [vp, u32, fp, fp, fp]
    L.salva_hip_force_add_accelerations.argtypes = [vp, u32, fp]
    L.salva_hip_set_fluid_field.argtypes = [vp, u32, i32, fp]
    L.salva_hip_get_timestep.argtypes = [vp, fp, fp]
    L.salva_hip_set_timestep.argtypes = [vp, f32, f32]
    L.salva_hip_device_bytes.argtypes = [vp]
    L.salva_hip_device_bytes.restype = u64
    L.salva_hip_time_pred_density.argtypes = [vp, i32]
    L.salva_hip_time_pred_density.restype = f32
    L.salva_hip_particles_intersecting_shape.argtypes = [vp, fp, fp, C.POINTER(Shape), u64, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.salva_hip_particles_intersecting_shape.restype = C.c_int64
    L.salva_hip_particles_intersecting_host_shape.argtypes = [vp, C.POINTER(HostQueryShape), u64, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.salva_hip_particles_intersecting_host_shape.restype = C.c_int64
    L.salva_hip_delete_owned.argtypes = [vp, u32, C.POINTER(u32)]
    L.salva_hip_delete_owned.restype = C.c_int64
    L.salva_hip_rebalance.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.salva_hip_time_kernel.argtypes = [vp, i32, i32]
    L.salva_hip_time_kernel.restype = f32
    L.salva_hip_get_counters.argtypes = [vp, C.POINTER(CountersStruct)]
    L.salva_hip_set_cfl.argtypes = [vp, i32, C.c_float, i32, i32]
    L.salva_hip_get_substeps.restype = C.c_int64
    L.salva_hip_get_substeps.argtypes = [vp, C.POINTER(C.c_float), u64]
    L.salva_hip_enable_counters.argtypes = [vp, i32]
    if hasattr(L, "salva_hip_time_variant"):  # the kernel-development build only (SALVA_HIP_LIB_VARIANT=diag)
        L.salva_hip_time_variant.argtypes = [vp, i32, u32, i32, C.POINTER(u64)]
        L.salva_hip_time_variant.restype = f32
    ubp = C.POINTER(C.c_ubyte)
    L.salva_hip_comm_rccl_unique_id.argtypes = [ubp]
    L.salva_hip_comm_rccl_create.argtypes = [i32, i32, ubp, i32, C.POINTER(vp)]
    L.salva_hip_comm_loopback_create.argtypes = [i32, C.POINTER(vp)]
    L.salva_hip_comm_peer_begin.argtypes = [i32, i32, i32, u64, ubp, C.POINTER(vp)]
    L.salva_hip_comm_peer_connect.argtypes = [vp, ubp, C.POINTER(vp)]
    L.salva_hip_comm_peer_abort.argtypes = [vp]
    L.salva_hip_comm_peer_abort.restype = None
    L.salva_hip_comm_selftest.argtypes = [vp, u64, i32]
    L.salva_hip_comm_time.argtypes = [vp, u64, i32, C.POINTER(f32), C.POINTER(f32)]
    L.salva_hip_comm_destroy.argtypes = [vp]
    L.salva_hip_comm_destroy.restype = None
    L.salva_hip_set_domain.argtypes = [vp, vp, i32, i32, u32]
    L.salva_hip_get_owned.argtypes = [vp, u32, C.POINTER(u32), fp, fp, C.POINTER(u32)]
    L.salva_hip_get_owned.restype = C.c_int64
    L.salva_hip_particles_intersecting_aabb.argtypes = [vp, fp, fp, u64, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.salva_hip_particles_intersecting_aabb.restype = C.c_int64
    L.salva_hip_add_particles.argtypes = [vp, u32, u64, fp, fp]
    L.salva_hip_delete_particles.argtypes = [vp, u32, C.POINTER(C.c_uint8)]
    L.salva_hip_delete_particles.restype = C.c_int64
    L.salva_hip_get_fluid_contacts.argtypes = [vp, u32, i32, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), u64]
    L.salva_hip_get_fluid_contacts.restype = C.c_int64
    L.salva_hip_get_force_stats.argtypes = [vp, u32, u32, C.POINTER(i32), fp]
    L.salva_hip_last_error.restype = C.c_char_p
    L.salva_hip_version.restype = C.c_char_p
    _lib = L
    return L


def check(code: int):
    if code != OK:
        raise SalvaHipError(code, lib().salva_hip_last_error().decode("utf-8", "replace"))
