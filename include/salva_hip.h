/* salva_hip.h — C ABI of libsalva_hip.so: an MI355X-native replacement for the body of
 * salva3d's `LiquidWorld::step_with_coupling` (coupling = `()`).
 *
 * The reference (dimforge/salva, pure Rust) has no FFI for this path; its seams are Rust structs and
 * traits.  Neighbour search runs inside `LiquidWorld::step_with_coupling` before the `PressureSolver`
 * trait object is called (/root/reference/src/liquid_world.rs:88-128), so the drop-in unit is the whole
 * step: a salva3d-compatible `LiquidWorld` whose `step` forwards to `salva_hip_step`.  Every entry point
 * below names the reference interface it replaces.  The Rust binding a maintainer would add is shown in
 * INTEGRATION.md; include/salva_hip.hpp and salva_amd/world.py are the C++ / Python mirrors used here
 * (there is no cargo/rustc in this image).
 *
 * Conventions
 *  - plain pointers and sizes only; host arrays are caller-owned and borrowed for the duration of a call;
 *    3-vectors are AoS [x,y,z] f32 = the memory layout of Vec<Point3<f32>> / Vec<Vector3<f32>>.
 *  - return 0 on success, negative on error (SALVA_HIP_E_*); message via salva_hip_last_error().
 *    The reference panics (assert!/unwrap, e.g. dfsph_solver.rs:92,145,662); this ABI never unwinds.
 *  - a world may be used from any one thread at a time (LiquidWorld: Send + Sync, liquid_world.rs:283-287);
 *    every entry point selects the world's device itself.
 *  - there is no CPU fallback: without a usable HIP device salva_hip_create fails with SALVA_HIP_E_HIP.
 */
#ifndef SALVA_HIP_H
#define SALVA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SalvaHipWorld SalvaHipWorld;

enum {
    SALVA_HIP_OK = 0,
    SALVA_HIP_E_HIP = -1,       /* a HIP runtime call failed / no device */
    SALVA_HIP_E_INVALID = -2,   /* invalid argument */
    SALVA_HIP_E_NUMERIC = -3,   /* zero density / boundary denominator / NaN — the reference's assert! cases */
    SALVA_HIP_E_CAPACITY = -4   /* grid or neighbour list exceeds addressable size */
};

/* which pressure solver: solver::DFSPHSolver (dfsph_solver.rs) or solver::IISPHSolver (iisph_solver.rs) */
enum { SALVA_HIP_SOLVER_DFSPH = 0, SALVA_HIP_SOLVER_IISPH = 1 };
/* The `KernelDensity` / `KernelGradient` type parameters of DFSPHSolver<..> / IISPHSolver<..> (dfsph_solver.rs:17-20,
 * iisph_solver.rs:17-20; src/kernel/{cubic_spline,poly6,spiky,viscosity}_kernel.rs).  Every contact's weight comes from the
 * first and its gradient from the second (solver/helper.rs:9-63), for the solver and for every NonPressureForce. */
enum {
    SALVA_HIP_KERNEL_CUBIC_SPLINE = 0, /* the default of both parameters, and what every example and benchmark uses */
    SALVA_HIP_KERNEL_POLY6 = 1,
    SALVA_HIP_KERNEL_SPIKY = 2,
    SALVA_HIP_KERNEL_VISCOSITY = 3
};

/* Mirrors `LiquidWorld::new(solver, particle_radius, smoothing_factor)` (liquid_world.rs:39-57) plus the
 * pub tuning fields of DFSPHSolver (dfsph_solver.rs:21-38, defaults :54-70) / IISPHSolver (iisph_solver.rs:21-30,
 * defaults :48-64). */
typedef struct SalvaHipParams {
    float particle_radius;
    float smoothing_factor;          /* h = particle_radius * smoothing_factor * 2 */
    int32_t solver;                  /* SALVA_HIP_SOLVER_* */
    int32_t min_pressure_iter;       /* 1 */
    int32_t max_pressure_iter;       /* 50 */
    float max_density_error;         /* 0.05 */
    int32_t min_divergence_iter;     /* 1   (DFSPH only) */
    int32_t max_divergence_iter;     /* 50  (DFSPH only) */
    float max_divergence_error;      /* 0.1 (DFSPH only) */
    int32_t device;                  /* HIP device ordinal */
    int32_t enable_timers;           /* fill the *_ms fields of SalvaHipStepStats (Counters, counters/mod.rs:17-72) */
    int32_t kernel_density;          /* SALVA_HIP_KERNEL_* (0 = CubicSplineKernel) */
    int32_t kernel_gradient;         /* SALVA_HIP_KERNEL_* (0 = CubicSplineKernel) */
    int32_t reserved[5];
} SalvaHipParams;

/* Built-in `NonPressureForce` implementations that run on the device.  A `Fluid` holds a list of them
 * (`fluid.nonpressure_forces`, object/fluid.rs:14); they are applied in list order by `predict_advection`
 * (dfsph_solver.rs:565-604). */
enum {
    SALVA_HIP_FORCE_XSPH = 1,        /* solver::XSPHViscosity::new(fluid_coeff, boundary_coeff), xsph_viscosity.rs:22-28 */
    SALVA_HIP_FORCE_ARTIFICIAL = 2,  /* solver::ArtificialViscosity::new(fluid_coeff, boundary_coeff), artificial_viscosity.rs:29-37 */
    SALVA_HIP_FORCE_AKINCI2013 = 3,  /* solver::Akinci2013SurfaceTension::new(tension, adhesion), akinci2013_surface_tension.rs:29-35 */
    SALVA_HIP_FORCE_DFSPH_VISCOSITY = 4, /* solver::DFSPHViscosity::new(viscosity_coefficient), dfsph_viscosity.rs:102-118 */
    SALVA_HIP_FORCE_HE2014 = 5,      /* solver::He2014SurfaceTension::new(fluid_tension, boundary_tension), he2014_surface_tension.rs:21-29 */
    SALVA_HIP_FORCE_WCSPH_TENSION = 6, /* solver::WCSPHSurfaceTension::new(fluid_tension, boundary_tension), wcsph_surface_tension.rs:22-28 */
    SALVA_HIP_FORCE_CUSTOM = 7       /* any other `impl NonPressureForce` (nonpressure_force.rs:10-30): runs on the host through
                                        the callback of salva_hip_set_force_callback, at its place in the list */
};
typedef struct SalvaHipForceDesc {
    int32_t kind;
    /* XSPH:       p[0] fluid_viscosity_coefficient, p[1] boundary_viscosity_coefficient
     * ARTIFICIAL: p[0] fluid coeff, p[1] boundary coeff, p[2] alpha (1), p[3] beta (0), p[4] speed_of_sound (10)
     * AKINCI2013: p[0] fluid_tension_coefficient, p[1] boundary_adhesion_coefficient
     * DFSPH_VISCOSITY: p[0] viscosity_coefficient (0..1), p[1] min_viscosity_iter (1), p[2] max_viscosity_iter (50),
     *                  p[3] max_viscosity_error (0.01)          — the pub fields of DFSPHViscosity, dfsph_viscosity.rs:89-99
     * HE2014:     p[0] fluid_tension_coefficient, p[1] boundary_tension_coefficient
     * WCSPH_TENSION: p[0] fluid_tension_coefficient, p[1] boundary_tension_coefficient — must be 0: the reference's boundary
     *                  loop (wcsph_surface_tension.rs:66-83) indexes the boundaries with fluid-fluid contacts and panics */
    float p[7];
} SalvaHipForceDesc;

/* dirty_mask bits of salva_hip_set_fluid: which host arrays changed since the last call */
enum {
    SALVA_HIP_DIRTY_POSITIONS = 1,
    SALVA_HIP_DIRTY_VELOCITIES = 2,
    SALVA_HIP_DIRTY_VOLUMES = 4,
    SALVA_HIP_DIRTY_ACCELERATIONS = 8,
    SALVA_HIP_DIRTY_ALL = 15
};

/* Per-step report; replaces the `Counters` the reference fills (counters/mod.rs:17-72, liquid_world.rs:73-156). */
typedef struct SalvaHipStepStats {
    int32_t n_divergence_iters;   /* compute_velocity_changes_for_divergence applications (dfsph_solver.rs:474-502) */
    int32_t n_pressure_iters;     /* compute_velocity_changes applications (:439-463) / IISPH Jacobi iterations */
    float divergence_error;       /* last evaluated average divergence error */
    float density_error;          /* last evaluated average density error */
    uint64_t ncontacts;           /* counters.cd.ncontacts (liquid_world.rs:119): ff + fb + bb directed contacts */
    uint64_t nparticles;          /* fluid particles stepped */
    float grid_ms;                /* counters.cd.grid_insertion_time + neighborhood_search_time equivalents */
    float solver_ms;              /* counters.stages.solver_time */
    float step_ms;                /* counters.step_time */
    float reserved[5];
} SalvaHipStepStats;

/* `LiquidWorld::counters` — the reference's `Counters` tree, field for field (counters/mod.rs:17-30,
 * stages_counters.rs:6-11, collision_detection_counters.rs:6-17, solver_counters.rs:6-11), as the testbed plugins read it
 * (testbed_plugin.rs:508-510).  Times are milliseconds like the reference's `Timer::time()` (instant::now() is in ms),
 * measured with HIP events on the world's stream (NaN = an interval that could not be read), and only filled when
 * SalvaHipParams::enable_timers is set
 * (`Counters::enable`); the counts are always filled.  Filled by salva_hip_step, read with salva_hip_get_counters. */
typedef struct SalvaHipCounters {
    uint64_t nsubsteps;                       /* liquid_world.rs:86 — 1 per step (0 for dt <= eps): the reference never sub-steps;
                                                 more only with salva_hip_set_cfl */
    double step_time;                         /* liquid_world.rs:74,156 */
    double custom;                            /* dfsph_solver.rs:492-501: the divergence solve */
    struct {
        double collision_detection_time;      /* liquid_world.rs:88-120 */
        double solver_time;                   /* liquid_world.rs:122-147 */
    } stages;
    struct {
        uint64_t ncontacts;                   /* liquid_world.rs:119 */
        double boundary_update_time;          /* liquid_world.rs:94-103: coupling.update_boundaries — the DynamicContactSampling
                                                 pass inside the step; statically sampled boundaries are posed by the caller's
                                                 salva_hip_update_boundary_pose before the step and add nothing here */
        double grid_insertion_time;           /* liquid_world.rs:89-92,105-107: cell sort of fluids (+ boundaries when changed) */
        double neighborhood_search_time;      /* contacts.rs:154-252 via update_contacts: tile tables + neighbour lists */
        double contact_sorting_time;          /* never started in the reference: 0 */
    } cd;
    struct {
        double non_pressure_resolution_time;  /* never started in the reference: 0 */
        double pressure_resolution_time;      /* dfsph_solver.rs:677-707 / iisph_solver.rs:664-710: the whole solver.step */
    } solver;
    /* --- not in the reference: what this implementation adds to a step report --- */
    int32_t n_divergence_iters, n_pressure_iters;
    uint64_t speculative_passes;              /* steps whose table sizes were predicted from the previous step (no mid-step read-back) */
    uint64_t discarded_passes;                /* passes discarded and repeated: a failed prediction, or a neighbour list longer than
                                                 the capacity it was built with (checked at the end of the step) */
    uint64_t chained_passes;                  /* DFSPH passes whose solves and everything behind them were enqueued without a host
                                                 wait in between and whose chain held (one wait per step: its last read-back) */
    uint64_t chain_breaks;                    /* ... whose chain broke: a solve needed more iterations than the batch enqueued for it,
                                                 the kernels behind it returned at once and the host continued the classic way */
    uint64_t pregrid_adopted;                 /* steps that found their grid part (keys, cell sort, tile tables) on the device already,
                                                 enqueued by the end of the step before */
    uint64_t pregrid_dropped;                 /* ... that found one they could not use (the cell box moved, the host edited the world) */
    uint64_t light_class_passes;              /* passes whose tile kernels ran the slots with small halos in launches of their own, on the
                                                 three-workgroups-per-CU layouts, beside slots whose halos are beyond those */
    uint64_t sparse_class_passes;             /* passes whose tile kernels ran the sparse slots (a stray particle or a few, alone in
                                                 their tile) in launches of their own: 64 threads, a few KB of LDS */
} SalvaHipCounters;

/* fields of salva_hip_get_fluid_field (solver scratch the reference keeps private; exposed for parity tests) */
enum {
    SALVA_HIP_FIELD_DENSITY = 0,            /* densities            f32 x n */
    SALVA_HIP_FIELD_ALPHA = 1,              /* alphas (DFSPH)       f32 x n */
    SALVA_HIP_FIELD_NUM_FLUID_CONTACTS = 2, /* len of fluid_fluid_contacts[i]    (as f32) x n */
    SALVA_HIP_FIELD_NUM_BOUNDARY_CONTACTS = 3, /* len of fluid_boundary_contacts[i] (as f32) x n */
    SALVA_HIP_FIELD_VELOCITY_CHANGE = 4,    /* velocity_changes     f32 x 3n */
    SALVA_HIP_FIELD_PRESSURE = 5,           /* pressures (IISPH)    f32 x n */
    SALVA_HIP_FIELD_VOLUME = 6,             /* volumes              f32 x n */
    SALVA_HIP_FIELD_ACCELERATION = 7        /* accelerations        f32 x 3n */
};

void salva_hip_default_params(SalvaHipParams* out);

/* LiquidWorld::new — liquid_world.rs:39-57 */
int salva_hip_create(const SalvaHipParams* params, SalvaHipWorld** out);
void salva_hip_destroy(SalvaHipWorld* world);

/* LiquidWorld::h / particle_radius — liquid_world.rs:201-208 */
float salva_hip_h(const SalvaHipWorld* world);

/* LiquidWorld::add_fluid (liquid_world.rs:161-163) when slot == current number of fluids, otherwise the
 * upload half of `fluids_mut().get_mut(handle)` edits (pub fields of Fluid, object/fluid.rs:12-34).
 * `slot` is the dense index of the FluidSet (object/contiguous_arena.rs).  velocities / volumes / accelerations
 * may be NULL (zeros / Fluid::particle_volume = 0.8 (2r)^3, fluid.rs:110-120 / zeros).  When `n` differs from
 * the stored count every array passed replaces the old one and the solver's velocity_changes of that fluid
 * restart from zero (or from `velocity_changes` if non-NULL — used to carry init_with_fluids' compaction,
 * dfsph_solver.rs:526-561). */
int salva_hip_set_fluid(SalvaHipWorld* world, uint32_t slot, uint64_t n,
                        const float* positions_xyz, const float* velocities_xyz, const float* volumes,
                        const float* accelerations_xyz, const float* velocity_changes_xyz,
                        float density0, uint32_t memberships, uint32_t filter, uint32_t dirty_mask);

/* `fluid.nonpressure_forces` — object/fluid.rs:14 */
int salva_hip_set_fluid_forces(SalvaHipWorld* world, uint32_t slot, const SalvaHipForceDesc* forces, uint32_t nforces);

/* LiquidWorld::remove_fluid (liquid_world.rs:171-173): swap-remove, like ContiguousArena::remove */
int salva_hip_remove_fluid(SalvaHipWorld* world, uint32_t slot);

/* LiquidWorld::add_boundary (liquid_world.rs:166-168) / edits of Boundary's pub fields (object/boundary.rs:11-24).
 * velocities may be NULL (zeros).  wants_forces != 0 <=> `boundary.forces = Some(..)`. */
int salva_hip_set_boundary(SalvaHipWorld* world, uint32_t slot, uint64_t n,
                           const float* positions_xyz, const float* velocities_xyz,
                           uint32_t memberships, uint32_t filter, int32_t wants_forces);
int salva_hip_remove_boundary(SalvaHipWorld* world, uint32_t slot);

uint32_t salva_hip_num_fluids(const SalvaHipWorld* world);
uint32_t salva_hip_num_boundaries(const SalvaHipWorld* world);
uint64_t salva_hip_fluid_len(const SalvaHipWorld* world, uint32_t slot);
uint64_t salva_hip_boundary_len(const SalvaHipWorld* world, uint32_t slot);

/* LiquidWorld::step(dt, gravity) — liquid_world.rs:62-158 with the no-op `()` CouplingManager. */
int salva_hip_step(SalvaHipWorld* world, float dt, const float gravity[3], SalvaHipStepStats* stats_or_null);

/* Download half of `fluids().get(handle)`: positions / velocities after the step (any pointer may be NULL). */
int salva_hip_get_fluid(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz);
int salva_hip_get_fluid_field(SalvaHipWorld* world, uint32_t slot, int32_t field, float* out);

/* ---- Rigid-body coupling on the device: the StaticSampling arm of salva's rapier integration
 * (src/integrations/rapier/fluids_pipeline.rs).  The rigid-body engine stays on the host; per step it hands over one pose
 * per coupled collider and receives one wrench back, instead of re-uploading every boundary particle and downloading
 * every force. */
typedef struct SalvaHipRigidPose {
    float translation[3];   /* collider.position().translation */
    float rotation[4];      /* collider.position().rotation as a unit quaternion (i, j, k, w) — nalgebra's storage order */
    float linvel[3];        /* body.linvel() */
    float angvel[3];        /* body.angvel() */
    float world_com[3];     /* body.center_of_mass() (world space) */
    int32_t has_body;       /* collider.parent() is Some: 0 -> velocities are zero and `forces` is left as it is */
    int32_t is_dynamic;     /* body.is_dynamic(): the boundary receives forces iff set (fluids_pipeline.rs:163-171) */
} SalvaHipRigidPose;

/* ColliderCouplingSet::register_coupling(boundary, collider, ColliderSampling::StaticSampling(points))
 * (fluids_pipeline.rs:36-41, 96-114): creates or resizes boundary `slot` (like salva_hip_set_boundary) and keeps the
 * collider-local sample points on the device.  Until the first pose the boundary sits at the identity pose. */
int salva_hip_set_boundary_sampling(SalvaHipWorld* world, uint32_t slot, uint64_t n, const float* local_points_xyz,
                                    uint32_t memberships, uint32_t filter);
/* ColliderCouplingManager::update_boundaries, StaticSampling arm (fluids_pipeline.rs:160-193, 262), as one kernel:
 * positions[i] = pose * points[i]; velocities[i] = body.velocity_at_point(points[i]) = linvel + angvel x (points[i] -
 * world_com) — the reference passes the LOCAL point there (:183), reproduced as written; forces are cleared.
 * Call before salva_hip_step (the reference calls it at the top of the substep, liquid_world.rs:93-101). */
int salva_hip_update_boundary_pose(SalvaHipWorld* world, uint32_t slot, const SalvaHipRigidPose* pose);
/* Reads `boundary.positions` / `boundary.velocities` (object/boundary.rs:13-15) back, e.g. after a pose update.  Any pointer may be NULL. */
int salva_hip_get_boundary_particles(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz);
/* ColliderCouplingManager::transmit_forces (fluids_pipeline.rs:266-287) applies `force_i * dt` at `position_i` for every
 * boundary particle; this returns the two sums that is equivalent to, reduced on the device:
 *   force = sum_i f_i,  torque = sum_i (x_i - point) x f_i   ->  body.apply_impulse(force*dt), apply_torque_impulse(torque*dt)
 * with point = body.center_of_mass().  Zero when the boundary does not receive forces. */
int salva_hip_get_boundary_wrench(SalvaHipWorld* world, uint32_t slot, const float point[3], float force[3], float torque[3]);

/* ---- User-defined `NonPressureForce`s (solver/nonpressure_force.rs:10-30; examples3d/custom_forces3.rs:67-90).
 * A SALVA_HIP_FORCE_CUSTOM entry in a fluid's force list makes salva_hip_step call `cb` in the middle of the substep, at the
 * point where `predict_advection` would call the force's `solve` (dfsph_solver.rs:580-603): after gravity and the forces
 * listed before it were accumulated, with the contacts, densities and (divergence-corrected) velocities of this substep.
 * `dt` / `inv_dt` are what `timestep.dt()` / `inv_dt()` return at that point (the previous substep's).
 * Inside the callback — and only there — the host may call salva_hip_force_get_state, salva_hip_get_fluid_contacts (both
 * kinds), salva_hip_get_boundary_particles / salva_hip_get_boundary (volumes) and salva_hip_force_add_accelerations; other
 * entry points fail with SALVA_HIP_E_INVALID.  A non-zero return aborts the step with SALVA_HIP_E_INVALID.
 * This is the slow path by construction (the state crosses PCIe twice per step); the built-in kinds never leave the device. */
typedef int (*SalvaHipForceCallback)(void* user, SalvaHipWorld* world, uint32_t fluid_slot, uint32_t force_index, float dt,
                                     float inv_dt);
int salva_hip_set_force_callback(SalvaHipWorld* world, SalvaHipForceCallback cb, void* user);
/* `fluid.positions`, `fluid.velocities`, `densities` as `NonPressureForce::solve` receives them (host order; NULL = skip). */
int salva_hip_force_get_state(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz, float* densities);
/* `fluid.accelerations[i] += acc[i]` */
int salva_hip_force_add_accelerations(SalvaHipWorld* world, uint32_t slot, const float* accelerations_xyz);

/* ---- `CouplingManager` inside the substep loop (coupling_manager.rs:9-28; liquid_world.rs:85-147).
 * `LiquidWorld::step_with_coupling` calls `coupling.update_boundaries(&timestep, h, r, &hgrid, fluids, boundaries)` at the top of
 * EVERY substep (:94-103) and `coupling.transmit_forces(&timestep, boundaries)` at its end (:146); with one substep per step — the
 * reference as it runs — a caller can do both around salva_hip_step (salva_hip_update_boundary_pose before, salva_hip_get_boundary_wrench
 * after).  With salva_hip_set_cfl the substeps are chosen inside the solver, and the rapier manager applies each substep's impulse
 * (force * timestep.dt(), fluids_pipeline.rs:266-287) to its bodies before the next substep samples their velocities
 * (:160-193): one wrench per step cannot carry that.  A coupling callback puts the host back where the reference has it:
 *   phase 0  = update_boundaries: at the top of every substep, before anything of the substep has run; `dt` = timestep.dt(), the
 *              previous substep's length.  Call salva_hip_update_boundary_pose (which clears the boundary's forces, :262) here.
 *   phase 1  = transmit_forces: the substep is complete, boundary forces are final; `dt` = the length of THIS substep.  Call
 *              salva_hip_get_boundary_wrench (or salva_hip_get_boundary) and apply force * dt to the body.
 * Inside either phase the pose / wrench / boundary read-back entry points are available; the fluids and the object sets must not
 * be edited.  A non-zero return aborts the step with SALVA_HIP_E_INVALID.  With a callback registered the combination
 * "CFL sub-stepping + coupled boundary that wants forces" is accepted; without one it stays refused.  NULL unregisters. */
typedef int (*SalvaHipCouplingCallback)(void* user, SalvaHipWorld* world, int32_t phase, float dt);
int salva_hip_set_coupling_callback(SalvaHipWorld* world, SalvaHipCouplingCallback cb, void* user);

/* ---- Checkpoint / restart.  The state `LiquidWorld::step` carries from one call to the next is:
 * positions, velocities and volumes of every fluid (salva_hip_get_fluid / get_fluid_field(VOLUME)), the solver's
 * `velocity_changes` (dfsph_solver.rs:41, applied at the top of the next step) and, for IISPH, the `pressures` its Jacobi
 * loop warm-starts from (iisph_solver.rs:35, :428-431) — readable with salva_hip_get_fluid_field and writable with
 * salva_hip_set_fluid_field (fields VELOCITY_CHANGE and PRESSURE only) — plus the TimestepManager's `dt` / `inv_dt`
 * (timestep_manager.rs:23-34), which the next step reads before advancing (the dt lag of divergence_solve and of the
 * non-pressure forces).  A world rebuilt from these continues the run: same contacts and iteration counts, state equal up to
 * f32 summation order (the order of particles within a cell restarts from host order). */
int salva_hip_set_fluid_field(SalvaHipWorld* world, uint32_t slot, int32_t field, const float* data);
int salva_hip_get_timestep(const SalvaHipWorld* world, float* dt, float* inv_dt);
int salva_hip_set_timestep(SalvaHipWorld* world, float dt, float inv_dt);

/* `boundary.volumes` (recomputed every substep, dfsph_solver.rs:72-96) and `boundary.forces`
 * (accumulated by Boundary::apply_force, boundary.rs:62-67).  Any pointer may be NULL. */
int salva_hip_get_boundary(SalvaHipWorld* world, uint32_t slot, float* volumes, float* forces_xyz);
/* Boundary::clear_forces — boundary.rs:70-82 */
int salva_hip_clear_boundary_forces(SalvaHipWorld* world, uint32_t slot);

/* Device residency helpers for benchmarks: bytes of HBM held by the world; algorithmic byte model of the last
 * step evaluated with the measured mean contact count K and iteration counts. */
uint64_t salva_hip_device_bytes(const SalvaHipWorld* world);

/* Timing hook for bench.py's roofline leg: re-launches the last step's k_pred_density kernel `reps` times on
 * the world's stream between two hipEvents and returns the average launch duration in microseconds
 * (negative on error).  State is not modified (the kernel rewrites the same outputs from the same inputs). */
float salva_hip_time_pred_density(SalvaHipWorld* world, int32_t reps);
/* The same for the other neighbour-sum kernels bench.py reports a roofline for: `kernel` = 0 k_pred_density (DFSPH, N (4K + 52)
 * algorithmic bytes), 1 k_divergence (N (4K + 48)), 2 k_iisph_next_pressure (IISPH, N (4K + 60)), 3 k_iisph_dij_pj (N (4K + 36))
 * — SURVEY.md §8d; 4 k_nbr_tile (the neighbour-list build, N (12 + 4K)), 6 k_divergence_apply (DFSPH, N (4K + 44); runs on a copy
 * of w).  The IISPH kernels rewrite scratch only (next pressures into the spare buffer). */
float salva_hip_time_kernel(SalvaHipWorld* world, int32_t kernel, int32_t reps);
/* `world.counters` after the last step — counters/mod.rs:17-72 */
int salva_hip_get_counters(const SalvaHipWorld* world, SalvaHipCounters* out);
/* `Counters::enable()` / `disable()` (counters/mod.rs:56-72): switches the timers (SalvaHipParams::enable_timers) from the next step
 * on.  Off — the reference's default, `Timer::new` (counters/timer.rs:11-18) — a step records no events and the *_ms / *_time
 * fields read 0; on, a step costs ~20 us more of host time (ten event records and their read-out). */
int salva_hip_enable_counters(SalvaHipWorld* world, int32_t enabled);
/* Opt-in CFL sub-stepping — SURVEY.md row f4.  The reference's TimestepManager carries cfl_coeff = 0.4 and 1..10 substeps
 * (timestep_manager.rs:23-34) and `max_substep` (:36-46), but `compute_substep` returns the whole step and leaves the clamp
 * commented out below a FIXME (:87-94): every `step` is ONE substep.  That stays the default here (mode 0).
 *   mode 1: the commented code, literally — in every pass of the substep loop (liquid_world.rs:85) the solver's `timestep.advance`
 *           (dfsph_solver.rs:702, iisph_solver.rs:662) takes
 *               substep = clamp(particle_radius * 2 / sqrt(max_i |v_i + a_i * remaining_time|^2) * cfl_coeff,
 *                               dt / max_num_substeps, dt / min_num_substeps)
 *           (the last substep may step past the end of the step: the commented code does not cut it);
 *   mode 2: the same, cut at the remaining time; a remainder below 1e-4 dt is taken along with the substep before it (no last substep
 *           of float residue).
 * `counters.nsubsteps` counts the passes, the timers add up over them, the iteration counts and errors of SalvaHipStepStats are
 * the last pass's.  Plain boundaries accumulate reaction forces over the substeps as the reference's do; a COUPLED boundary that
 * wants forces needs salva_hip_set_coupling_callback (the reference transmits its impulse per substep, which one wrench per step
 * cannot carry) and is refused with SALVA_HIP_E_INVALID without one.
 * In a decomposed run every rank must make the same call: the substep is chosen from an all-reduced maximum. */
int salva_hip_set_cfl(SalvaHipWorld* world, int32_t mode, float cfl_coeff, int32_t min_num_substeps, int32_t max_num_substeps);
/* Substep lengths of the last salva_hip_step (`TimestepManager::dt` of every pass): writes min(count, capacity) values, returns the
 * count (= counters.nsubsteps), or a negative error code. */
int64_t salva_hip_get_substeps(const SalvaHipWorld* world, float* out, uint64_t capacity);
/* ---- multi-GPU: one process and one world per GPU, the domain cut into slabs of grid-cell planes along x.
 * No counterpart in the reference (single process).  A world owns the particles whose cell x = floor(x / h) lies in
 * [cell_lo, cell_hi] (the first / last rank also keep whatever lies beyond their open end); every step it migrates
 * leavers to rank-1 / rank+1, mirrors the two cell planes at each face there as ghosts, refreshes the ghosts' fields
 * once per solver iteration and all-reduces the convergence sums, so iteration counts are global (a slab must span at
 * least four planes).  Each rank uploads only its own
 * particles with salva_hip_set_fluid (same fluid slots everywhere) plus the boundary particles within three cells of its
 * slab; `gid_offset` + upload index is the particle's global id.  After the first step per-fluid host-order access
 * (salva_hip_get_fluid) is replaced by salva_hip_get_owned. */
typedef struct SalvaHipComm SalvaHipComm;
/* RCCL over xGMI: rank 0 creates the 128-byte id and distributes it (torch.distributed, MPI, a file ...) */
int salva_hip_comm_rccl_unique_id(unsigned char* out128);
int salva_hip_comm_rccl_create(int32_t rank, int32_t size, const unsigned char* id128, int32_t device, SalvaHipComm** out);
/* xGMI peer-direct, for the ranks of ONE node (one process per rank): every exchange is a flagged store into a window of the
 * neighbour's memory mapped through hipIpc, the convergence all-reduce a flagged store into every rank's window — a few
 * microseconds instead of a collective-library round trip per solver iteration.  Two phases, because the windows' IPC handles
 * must travel between the processes: _begin allocates this rank's window (4 receive slots of `slot_bytes`; longer messages
 * go in rounds) and returns its 64-byte handle; the caller gathers the handles of all ranks in rank order (torch.distributed,
 * MPI, a file ...) and passes the size x 64 bytes to _connect, which consumes the setup object (also when it fails; _abort
 * frees a setup that is never connected).  At most 64 ranks.  Two ranks may share a GPU (how single-GPU boxes test it).
 * A rank whose neighbour does not show up within 30 s gets SALVA_HIP_E_HIP from its next call instead of a hung kernel. */
#define SALVA_HIP_PEER_HANDLE_BYTES 64
typedef struct SalvaHipPeerSetup SalvaHipPeerSetup;
int salva_hip_comm_peer_begin(int32_t rank, int32_t size, int32_t device, uint64_t slot_bytes, unsigned char* handle64,
                              SalvaHipPeerSetup** out);
int salva_hip_comm_peer_connect(SalvaHipPeerSetup* setup, const unsigned char* handles, SalvaHipComm** out);
void salva_hip_comm_peer_abort(SalvaHipPeerSetup* setup);
/* in-process loopback for tests: `size` communicators sharing a mailbox; drive each from its own host thread */
int salva_hip_comm_loopback_create(int32_t size, SalvaHipComm** out_ranks);
void salva_hip_comm_destroy(SalvaHipComm* comm);
/* Collective self-test of a communicator, any transport: `rounds` patterned exchanges of different (odd, empty, up to
 * `max_bytes`) lengths with both neighbours, a count exchange and both all-reduces per round, on a stream of its own.
 * SALVA_HIP_OK, or SALVA_HIP_E_HIP with the first mismatch in salva_hip_last_error(). */
int salva_hip_comm_selftest(SalvaHipComm* comm, uint64_t max_bytes, int32_t rounds);
/* Collective: what one solver iteration of a decomposed run adds — average microseconds (host clock over `iters` back-to-back
 * calls between two stream synchronisations) of one exchange of `bytes` bytes each way with both neighbours, and of one
 * all-reduce of four floats. */
int salva_hip_comm_time(SalvaHipComm* comm, uint64_t bytes, int32_t iters, float* us_exchange, float* us_allreduce);
int salva_hip_set_domain(SalvaHipWorld* world, SalvaHipComm* comm, int32_t cell_lo, int32_t cell_hi, uint32_t gid_offset);
/* Load balancing, collective (every rank calls it between two steps, after at least one step): the slabs are re-cut at cell
 * planes so that every rank owns about the same number of particles (all-reduced histogram over the planes; a cut stays
 * between the old cuts on either side, so a particle changes owner by one rank at most, and slabs stay four planes thick).
 * The particles follow in the next step's migration phase.  Returns this rank's new [cell_lo, cell_hi]; the caller re-uploads
 * the boundary particles its new slab needs (salva_hip_set_boundary), as it chose them for the old one. */
int salva_hip_rebalance(SalvaHipWorld* world, int32_t* cell_lo, int32_t* cell_hi);
/* particles currently owned by this rank, in no particular order; returns their number (negative on error) */
int64_t salva_hip_get_owned(SalvaHipWorld* world, uint32_t capacity, uint32_t* gids, float* positions_xyz,
                            float* velocities_xyz, uint32_t* fluid_slots);

/* Creation and removal of particles in a RUNNING decomposed world (faucet3.rs:69-104-style emitters and sinks).  Both are
 * collective: every rank calls them between the same two steps — with n = 0 where it has nothing to add or delete — because
 * the particle counts the solvers' error averages divide by are global, and so are the ids.
 *  - salva_hip_add_particles (below) appends to THIS rank: the positions must lie in its slab or the adjacent one (the next
 *    step's migration hands them over); the new particles get the ids following the largest id in the run, rank by rank;
 *  - salva_hip_delete_owned removes the particles of `gids` that this rank owns (ids owned elsewhere are ignored, so every rank
 *    may pass the same list); like `Fluid::delete_particle_at_next_timestep` (fluid.rs:71-86) they are gone from the next step
 *    on.  Returns the number of particles this rank still owns (negative on error).
 * Before the first step of a decomposed world the ordinary host-order calls apply (the upload index is the id). */
int64_t salva_hip_delete_owned(SalvaHipWorld* world, uint32_t n, const uint32_t* gids);

/* `LiquidWorld::particles_intersecting_aabb(aabb)` (liquid_world.rs:210-243): the particles whose distance to the box
 * [mins, maxs] is below the particle radius, as (kind, slot, index) triples sorted by kind (0 = ParticleId::FluidParticle,
 * 1 = BoundaryParticle), slot and index.  Returns how many there are (negative on error); at most `capacity` are written.
 * The reference filters the cells of the grid of its last step; this tests current positions, so it also finds particles
 * that entered the box's cells since then. */
int64_t salva_hip_particles_intersecting_aabb(SalvaHipWorld* world, const float mins[3], const float maxs[3], uint64_t capacity,
                                             uint32_t* kinds, uint32_t* slots, uint32_t* indices);

/* `LiquidWorld::particles_intersecting_shape(pos, shape)` (liquid_world.rs:245-280) for the analytic shapes the examples use,
 * entirely on the device (every other shape: salva_hip_particles_intersecting_host_shape below): the particles in the grid cells the posed shape's AABB touches whose distance to the
 * (solid) shape is <= the particle radius.  `rotation_ijkw` is the unit quaternion of the isometry.  Output and return value
 * as salva_hip_particles_intersecting_aabb; like it, current positions are tested. */
enum {
    SALVA_HIP_SHAPE_BALL = 1,      /* params[0] = radius */
    SALVA_HIP_SHAPE_CUBOID = 2,    /* params = half extents */
    SALVA_HIP_SHAPE_CAPSULE = 3,   /* parry Capsule::new_y: params[0] = half height of the segment along the local y axis, params[1] = radius */
    SALVA_HIP_SHAPE_CYLINDER = 4,  /* parry Cylinder (axis = local y): params[0] = half height, params[1] = radius */
    SALVA_HIP_SHAPE_HOST = 100     /* any other shape: its geometry stays with the host (SalvaHipHostShape below); never passed in a SalvaHipShape */
};
typedef struct SalvaHipShape {
    int32_t kind;
    float params[3];
} SalvaHipShape;
int64_t salva_hip_particles_intersecting_shape(SalvaHipWorld* world, const float translation[3], const float rotation_ijkw[4],
                                              const SalvaHipShape* shape, uint64_t capacity, uint32_t* kinds, uint32_t* slots,
                                              uint32_t* indices);

/* The same query for ANY other shape (the reference's is generic over parry's `Shape`, liquid_world.rs:247-250): the geometry
 * stays with the host.  The library calls `aabb` once (`shape.compute_aabb(pos)`), collects the particles of the grid cells that
 * box touches on the device (hgrid.cells_intersecting_aabb, hgrid.rs:122-133), calls `distance` once for all of them
 * (`shape.distance_to_point(pos, &pt, true)` per point) and reports those within the particle radius — one PCIe round trip of
 * 16 bytes per candidate.  Both callbacks run on the calling thread, inside this call; neither may call back into the world. */
typedef void (*SalvaHipHostAabbFn)(void* user, float* mins_xyz, float* maxs_xyz);
typedef void (*SalvaHipHostDistanceFn)(void* user, uint32_t n, const float* points_xyz, float* distances);
typedef struct SalvaHipHostQueryShape {
    SalvaHipHostAabbFn aabb;
    SalvaHipHostDistanceFn distance;
    void* user;
} SalvaHipHostQueryShape;
int64_t salva_hip_particles_intersecting_host_shape(SalvaHipWorld* world, const SalvaHipHostQueryShape* shape, uint64_t capacity,
                                                    uint32_t* kinds, uint32_t* slots, uint32_t* indices);

/* ---- Rigid-body coupling, the DynamicContactSampling arm (src/integrations/rapier/fluids_pipeline.rs:42-43, 193-259) for
 * ball, cuboid, capsule (y) and cylinder (y) colliders: no sample points are kept; inside every salva_hip_step — after the fluids went into the grid and
 * before the boundaries do, where `coupling.update_boundaries` runs (liquid_world.rs:94-103) — each fluid particle whose
 * predicted position x + v*dt lies in the collider's AABB loosened by 1.5 h is projected onto the shape
 * (`project_point_and_get_feature`); particles inside the shape are pushed out by depth + 0.1 r and lose their inward normal
 * velocity; one boundary particle per projection within 1.5 h is emitted (velocity = body.velocity_at_point(projection)).
 * As in the reference the grid is not rebuilt after the push-out: a pushed particle is searched from the cell of its old
 * position for that substep.  The collider's pose is handed over with salva_hip_update_boundary_pose before the step.
 * Registers boundary `slot` (created empty if slot == number of boundaries).
 * Decomposed (multi-GPU) worlds: COLLECTIVE — every rank registers the same collider in the same slot and hands over the same
 * pose.  Each rank emits for the fluid particles it owns, and every rank ends up with every rank's points, in rank order
 * (two all-reduces per collider and step); the force accumulator of a rank holds what that rank's own fluid exerts, so the
 * wrench on the collider is the sum of the ranks' salva_hip_get_boundary_wrench. */
int salva_hip_set_boundary_dynamic_sampling(SalvaHipWorld* world, uint32_t slot, const SalvaHipShape* collider_shape,
                                            uint32_t memberships, uint32_t filter);
/* The same arm for every other parry shape (triangle mesh, height field, convex polyhedron, compound, ...): the loop is the
 * same, the two calls into parry stay on the host.  Inside every salva_hip_step, on the calling thread and at the same point
 * of the step, the library calls `aabb` once (`collider.shape().compute_aabb(collider.position())`), copies the predicted
 * positions of the fluid particles that pass the reference's two box tests (fluids_pipeline.rs:204-211) to the host, calls
 * `project` once for all of them (`collider.shape().project_point_and_get_feature(collider.position(), &pt)` per point:
 * world-space projection and `proj.is_inside`), and finishes the loop body — push-out, reach test, emission — on the device
 * as for the built-in shapes.  Cost: two small PCIe round trips per step and collider.  The pose given to
 * salva_hip_update_boundary_pose is used for `velocity_at_point` only.  Neither callback may call back into the world. */
typedef void (*SalvaHipHostProjectFn)(void* user, uint32_t n, const float* points_xyz, float* projections_xyz, uint8_t* is_inside);
typedef struct SalvaHipHostShape {
    SalvaHipHostAabbFn aabb;
    SalvaHipHostProjectFn project;
    void* user;
} SalvaHipHostShape;
int salva_hip_set_boundary_dynamic_sampling_host(SalvaHipWorld* world, uint32_t slot, const SalvaHipHostShape* collider_shape,
                                                 uint32_t memberships, uint32_t filter);
/* (salva_hip_boundary_len reports what the last step emitted.) */
/* `ColliderCouplingSet::unregister_coupling` (fluids_pipeline.rs:116-125) as the library sees it: forget the sampling method of
 * boundary `slot` — static sample points, collider shape, host callbacks and their `user` pointer.  The boundary keeps the
 * particles it holds at that moment and is a plain boundary from then on (the reference's boundary object likewise stays in
 * the world with its last particles once its coupling entry is gone).  MUST be called before the memory behind a
 * SalvaHipHostShape's callbacks or `user` is released: the library calls them in every step while they are registered.
 * No-op for a boundary without a sampling method. */
int salva_hip_clear_boundary_sampling(SalvaHipWorld* world, uint32_t slot);
/* For a dynamically sampled boundary: (fluid slot, particle index) of the fluid particle behind each of its points, in the
 * order of salva_hip_get_boundary_particles (the order itself is unspecified, as the reference's hash-grid walk is).
 * Decomposed worlds: `indices` receives the particle's global id (the `ids` of salva_hip_get_local / _get_owned) — it may be
 * held by another rank. */
int salva_hip_get_boundary_sources(SalvaHipWorld* world, uint32_t slot, uint32_t* fluid_slots, uint32_t* indices);

/* Decomposed runs with the stage timers on (salva_hip_enable_counters): what the exchanges of the LAST step cost on this rank, from
 * HIP event pairs around them on the world's stream — out4 = {ms in ghost refreshes (gather -> exchange with the neighbours ->
 * scatter), number of refreshes, ms in all-reduced convergence tests (sum -> all-reduce -> decide), number of tests}.  The
 * times include waiting for the neighbour's data: they are what an exchange costs inside a real step, not the transport alone
 * (salva_hip_comm_time).  Zeros without a domain or with the timers off. */
int salva_hip_get_dist_timing(const SalvaHipWorld* world, double out4[4]);

/* ---- The working set as it is ("local view"): every fluid particle this world holds — in a decomposed run the particles the
 * rank owns AND its ghosts — in the order of the last step's cell sort, with global ids.  This is the per-rank form of what
 * salva_hip_get_fluid / salva_hip_get_fluid_contacts / salva_hip_force_get_state give a single-domain world in host order
 * (host order does not exist on a rank: particles migrate), and it is what a user-defined `NonPressureForce`
 * (nonpressure_force.rs:10-30, custom_forces3.rs:67-90) works on in a decomposed run: the callback of
 * salva_hip_set_force_callback runs on every rank, at its place in the force list, reads its rank's particles and contacts
 * here and returns accelerations in the same order.  Valid after a completed step and inside a force callback (where
 * velocities are w = v + dv, like salva_hip_force_get_state); the order changes with every step.
 *   salva_hip_local_len                   particles in the working set
 *   salva_hip_get_local                   ids (single domain: the particle's index over all fluids in slot order; decomposed: its
 *                                         global id), fluid slots, ghost flags (1 = not owned by this rank), positions, velocities,
 *                                         densities of the last density pass, particle volumes; any pointer may be NULL
 *   salva_hip_get_local_contacts          ParticlesContacts (contacts.rs:57-131) of ALL local particles as CSR: offsets has
 *                                         local_len + 1 entries; an entry is (j_model, j) with j a LOCAL index (fluid-fluid) or the
 *                                         index inside boundary j_model's arrays as this rank uploaded them (fluid-boundary).
 *                                         Returns the total; entries are written only if `capacity` holds them all.  The lists of
 *                                         ghost particles are complete for the inner ghost plane only.
 *   salva_hip_force_add_local_accelerations   inside a force callback: accelerations += a, local order, local_len x 3; what is
 *                                         added to a ghost is ignored (its owner computes it). */
uint64_t salva_hip_local_len(const SalvaHipWorld* world);
int salva_hip_get_local(SalvaHipWorld* world, uint32_t* ids, uint32_t* fluid_slots, uint8_t* is_ghost, float* positions_xyz,
                        float* velocities_xyz, float* densities, float* volumes);
int64_t salva_hip_get_local_contacts(SalvaHipWorld* world, int32_t boundary_contacts, uint64_t* offsets, uint32_t* j_model, uint32_t* j,
                                     uint64_t capacity);
int salva_hip_force_add_local_accelerations(SalvaHipWorld* world, const float* accelerations_xyz);

/* ---- Asynchronous read-back.  The reference's users read `fluid.positions` / `velocities` after every step
 * (src/integrations/rapier/testbed_plugin.rs:361-367: the renderer walks them each frame).  salva_hip_get_fluid does that
 * synchronously and serially with the step (0.6 ms for the 24 MB of 10^6 particles).  This pair takes the
 * copy off the critical path: salva_hip_get_fluid_async enqueues the read-back of the state AS IT IS NOW (after the last completed
 * step, host order, AoS [x,y,z] like salva_hip_get_fluid; either pointer may be NULL) and returns at once; the copy runs on the
 * world's copy stream while the next salva_hip_step is already computing; salva_hip_wait_download blocks until the arrays are
 * complete.  The destinations must stay valid and untouched until then.  One read-back is in flight at a time (a second
 * salva_hip_get_fluid_async waits for the first).  A destination in pinned memory — from salva_hip_host_alloc, or the caller's own
 * array after salva_hip_host_register (a Rust Vec / numpy array that is not reallocated) — is written by DMA directly at PCIe
 * speed; any other destination is served through the library's own pinned buffers plus one memcpy inside the wait.
 * Not available in a decomposed run (salva_hip_get_owned). */
int salva_hip_get_fluid_async(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz);
int salva_hip_wait_download(SalvaHipWorld* world);
/* Pinned host memory on the world's device context: NULL on failure (salva_hip_last_error).  Release with salva_hip_host_free. */
void* salva_hip_host_alloc(SalvaHipWorld* world, uint64_t bytes);
int salva_hip_host_free(void* p);
/* Pin a caller-owned array in place (hipHostRegister) / undo it.  The array must not move or be freed while registered. */
int salva_hip_host_register(SalvaHipWorld* world, void* p, uint64_t bytes);
int salva_hip_host_unregister(void* p);

/* `Fluid::add_particles(positions, velocities)` (object/fluid.rs:126-150): append to the fluid on the device — default
 * volume, zero acceleration and velocity change — without re-uploading the particles it already holds.
 * velocities_xyz may be NULL (zeros).  In a running decomposed world: collective, see salva_hip_delete_owned above. */
int salva_hip_add_particles(SalvaHipWorld* world, uint32_t slot, uint64_t n_add, const float* positions_xyz,
                            const float* velocities_xyz);
/* `Fluid::delete_particle_at_next_timestep` + `apply_particles_removal` (fluid.rs:71-98) and the solver's matching
 * compaction of its velocity_changes (dfsph_solver.rs:550-560): stable compaction of every per-particle array of the
 * fluid on the device.  deleted_mask has fluid_len bytes, non-zero = delete.  Returns the remaining count (negative on
 * error); the survivors keep their order, so the host compacts its copies with the same mask. */
int64_t salva_hip_delete_particles(SalvaHipWorld* world, uint32_t slot, const uint8_t* deleted_mask);

/* `ContactManager::fluid_fluid_contacts[slot]` / `fluid_boundary_contacts[slot]` (liquid_world.rs:26, geometry/contacts.rs:57-131)
 * of the last step as a CSR structure in host order: offsets has fluid_len + 1 entries, contact k of particle i is
 * (j_model[offsets[i] + k], j[offsets[i] + k]) — the model (fluid or boundary slot) and the index inside that model's host
 * arrays; the self contact is included, as in the reference.  Returns the number of contacts (negative on error); entries
 * are written only when `capacity` holds them all, so call once with capacity 0 to size the arrays.  This is what a host
 * fallback for user-defined NonPressureForce implementations iterates. */
int64_t salva_hip_get_fluid_contacts(SalvaHipWorld* world, uint32_t slot, int32_t boundary_contacts, uint64_t* offsets,
                                     uint32_t* j_model, uint32_t* j, uint64_t capacity);

/* Iterations and last average error of an iterative NonPressureForce (DFSPHViscosity's solve loop, dfsph_viscosity.rs:307-323)
 * in the last step; 0 / 0 for the other kinds.  `force` indexes the list given to salva_hip_set_fluid_forces. */
int salva_hip_get_force_stats(SalvaHipWorld* world, uint32_t slot, uint32_t force, int32_t* iters, float* error);

const char* salva_hip_last_error(void);
const char* salva_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SALVA_HIP_H */
