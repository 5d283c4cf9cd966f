//! `LiquidWorld` with salva3d's surface (reference: src/liquid_world.rs:17-280) whose `step` runs on the device.
//!
//! SOURCE ONLY — never compiled (no Rust toolchain where it was written).  Design notes:
//! * the host `FluidSet` / `BoundarySet` are salva3d's own; the device world mirrors them slot by slot (dense indices =
//!   `as_slice()` order, which is what `ContiguousArena::remove`'s swap-remove preserves on both sides);
//! * `fluids_mut()` / `boundaries_mut()` hand out `&mut` access, so everything is re-uploaded before the next step after
//!   either was called; a run that only calls `step` uploads nothing;
//! * after a step the new positions / velocities are downloaded into the host set (`auto_sync`, on by default: the
//!   reference's users read `fluid.positions` every frame); headless runs switch it off and call `sync()` when they look.
use crate::ffi;
use na::Vector3;
use nalgebra as na;
use salva3d::counters::Counters;
use salva3d::math::Real;
use salva3d::object::{Boundary, BoundaryHandle, BoundarySet, Fluid, FluidHandle, FluidSet};
use salva3d::solver::{DFSPHSolver, IISPHSolver};
use std::ffi::CStr;

#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

pub(crate) fn check(code: i32) -> Result<(), Error> {
    if code == ffi::SALVA_HIP_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(ffi::salva_hip_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code, message })
}

/// The two pressure solvers the device implements, recognised by their public tuning fields
/// (dfsph_solver.rs:21-38, iisph_solver.rs:21-30).
pub trait GpuPressureSolver {
    fn params(&self, particle_radius: Real, smoothing_factor: Real) -> ffi::SalvaHipParams;
}

fn default_params() -> ffi::SalvaHipParams {
    let mut p = std::mem::MaybeUninit::<ffi::SalvaHipParams>::uninit();
    unsafe {
        ffi::salva_hip_default_params(p.as_mut_ptr());
        p.assume_init()
    }
}

/// The reference's kernels as the device knows them (`SALVA_HIP_KERNEL_*`): the solvers' `KernelDensity` / `KernelGradient`
/// type parameters (dfsph_solver.rs:17-20, iisph_solver.rs:17-20) carry over unchanged.
pub trait GpuKernel: salva3d::kernel::Kernel {
    const KIND: i32;
}
impl GpuKernel for salva3d::kernel::CubicSplineKernel {
    const KIND: i32 = ffi::SALVA_HIP_KERNEL_CUBIC_SPLINE;
}
impl GpuKernel for salva3d::kernel::Poly6Kernel {
    const KIND: i32 = ffi::SALVA_HIP_KERNEL_POLY6;
}
impl GpuKernel for salva3d::kernel::SpikyKernel {
    const KIND: i32 = ffi::SALVA_HIP_KERNEL_SPIKY;
}
impl GpuKernel for salva3d::kernel::ViscosityKernel {
    const KIND: i32 = ffi::SALVA_HIP_KERNEL_VISCOSITY;
}

impl<KD: GpuKernel, KG: GpuKernel> GpuPressureSolver for DFSPHSolver<KD, KG> {
    fn params(&self, particle_radius: Real, smoothing_factor: Real) -> ffi::SalvaHipParams {
        let mut p = default_params();
        p.particle_radius = particle_radius;
        p.smoothing_factor = smoothing_factor;
        p.solver = ffi::SALVA_HIP_SOLVER_DFSPH;
        p.kernel_density = KD::KIND;
        p.kernel_gradient = KG::KIND;
        p.min_pressure_iter = self.min_pressure_iter as i32;
        p.max_pressure_iter = self.max_pressure_iter as i32;
        p.max_density_error = self.max_density_error;
        p.min_divergence_iter = self.min_divergence_iter as i32;
        p.max_divergence_iter = self.max_divergence_iter as i32;
        p.max_divergence_error = self.max_divergence_error;
        p
    }
}

impl<KD: GpuKernel, KG: GpuKernel> GpuPressureSolver for IISPHSolver<KD, KG> {
    fn params(&self, particle_radius: Real, smoothing_factor: Real) -> ffi::SalvaHipParams {
        let mut p = default_params();
        p.particle_radius = particle_radius;
        p.smoothing_factor = smoothing_factor;
        p.solver = ffi::SALVA_HIP_SOLVER_IISPH;
        p.kernel_density = KD::KIND;
        p.kernel_gradient = KG::KIND;
        p.min_pressure_iter = self.min_pressure_iter as i32;
        p.max_pressure_iter = self.max_pressure_iter as i32;
        p.max_density_error = self.max_density_error;
        p
    }
}

pub struct LiquidWorld {
    /// `world.counters` as the reference's plugins read it (counters/mod.rs:17-72); refreshed by every step.
    pub counters: Counters,
    raw: *mut ffi::SalvaHipWorld,
    particle_radius: Real,
    h: Real,
    fluids: FluidSet,
    boundaries: BoundarySet,
    host_dirty: bool,    // fluids_mut() / boundaries_mut() / add_* was called since the last upload
    device_newer: bool,  // a step ran since the last download
    auto_sync: bool,
    /// (address, bytes) of the host arrays pinned in place for the read-back (`salva_hip_host_register`).  A registration never
    /// outlives a point at which its `Vec` could be reallocated or freed: `fluids_mut()` (which hands out `&mut FluidSet`),
    /// and `remove_fluid` release every range first (`unpin_all`) and the next `sync` registers afresh (`add_fluid` only moves
    /// `Fluid` headers, not the heap blocks of their `Vec`s) — page-locked
    /// memory is never left behind in freed heap, and `hipHostUnregister` never runs on an address that is no longer mapped.
    pinned: Vec<(usize, usize)>,
    decomposed: bool,    // set_domain was called (dist.rs): particles are read with owned()
    last_stats: ffi::SalvaHipStepStats,
    cfl_mode: i32,       // what set_cfl_substepping was last called with (coupling.rs: the manager's calls move into the substep loop)
}

// Send: the C side keeps no thread-affine state — every entry point selects the world's device and stream itself, and the
// last-error string is thread-local.
// Sync (round 6): every C entry point that takes a world holds that world's lock for its duration (a recursive mutex in
// `SalvaHipWorld`, salva_amd/csrc/capi.hip — the C side reuses scratch buffers, one stream and lazily refreshed staging arrays per
// world), so the `&self` methods here, which touch nothing but the C handle, may be called from several threads at once; the
// methods that touch the Rust-side mirrors take `&mut self` and the borrow checker does the rest.  The reference pins the same two
// bounds at compile time (src/liquid_world.rs:283-287); tests/cpp/two_threads.cpp is the run-time half of it on this side.
unsafe impl Send for LiquidWorld {}
unsafe impl Sync for LiquidWorld {}

#[cfg(test)]
mod send_sync {
    // the reference's own test, liquid_world.rs:283-287
    #[test]
    fn world_is_send_and_sync() {
        fn check<T: Send + Sync>() {}
        check::<super::LiquidWorld>();
    }
}

impl Drop for LiquidWorld {
    fn drop(&mut self) {
        unsafe {
            ffi::salva_hip_wait_download(self.raw);
            for (p, _) in self.pinned.drain(..) {
                ffi::salva_hip_host_unregister(p as *mut std::ffi::c_void);
            }
            ffi::salva_hip_destroy(self.raw)
        }
    }
}

impl LiquidWorld {
    /// Release every in-place registration (see `pinned`); called before anything that may move or free a fluid's `Vec`s.
    fn unpin_all(&mut self) {
        unsafe {
            ffi::salva_hip_wait_download(self.raw); // (no DMA may still target the ranges)
            for (p, _) in self.pinned.drain(..) {
                ffi::salva_hip_host_unregister(p as *mut std::ffi::c_void);
            }
        }
    }

    /// `LiquidWorld::new(solver, particle_radius, smoothing_factor)` (liquid_world.rs:39-57).
    pub fn new(solver: impl GpuPressureSolver, particle_radius: Real, smoothing_factor: Real) -> Result<Self, Error> {
        let mut params = solver.params(particle_radius, smoothing_factor);
        params.enable_timers = 0; // `Counters::new`: timers disabled until `enable_counters(true)` (counters/timer.rs:11-18)
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::salva_hip_create(&params, &mut raw) })?;
        Ok(Self {
            counters: Counters::new(),
            raw,
            particle_radius,
            h: unsafe { ffi::salva_hip_h(raw) },
            fluids: FluidSet::new(),
            boundaries: BoundarySet::new(),
            host_dirty: true,
            device_newer: false,
            auto_sync: true,
            pinned: Vec::new(),
            decomposed: false,
            last_stats: unsafe { std::mem::zeroed() },
            cfl_mode: 0,
        })
    }

    pub fn set_auto_sync(&mut self, on: bool) {
        self.auto_sync = on;
    }

    pub fn h(&self) -> Real {
        self.h
    }
    pub fn particle_radius(&self) -> Real {
        self.particle_radius
    }

    pub fn add_fluid(&mut self, fluid: Fluid) -> FluidHandle {
        self.host_dirty = true;
        self.fluids.insert(fluid)
    }
    pub fn add_boundary(&mut self, boundary: Boundary) -> BoundaryHandle {
        self.host_dirty = true;
        self.boundaries.insert(boundary)
    }

    /// Swap-remove on both sides (liquid_world.rs:171-173; the solver's per-slot buffers stay positional there and here).
    pub fn remove_fluid(&mut self, handle: FluidHandle) -> Result<Option<Fluid>, Error> {
        self.sync()?;
        self.unpin_all(); // (the removed fluid's Vecs are about to leave the world, the last fluid's to change slot)
        let slot = self.fluids.iter().position(|(h, _)| h == handle);
        if let Some(slot) = slot {
            if (slot as u32) < unsafe { ffi::salva_hip_num_fluids(self.raw) } {
                check(unsafe { ffi::salva_hip_remove_fluid(self.raw, slot as u32) })?;
            }
        }
        self.host_dirty = true;
        Ok(self.fluids.remove(handle))
    }
    pub fn remove_boundary(&mut self, handle: BoundaryHandle) -> Result<Option<Boundary>, Error> {
        let slot = self.boundaries.iter().position(|(h, _)| h == handle);
        if let Some(slot) = slot {
            if (slot as u32) < unsafe { ffi::salva_hip_num_boundaries(self.raw) } {
                check(unsafe { ffi::salva_hip_remove_boundary(self.raw, slot as u32) })?;
            }
        }
        self.host_dirty = true;
        Ok(self.boundaries.remove(handle))
    }

    pub fn fluids(&self) -> &FluidSet {
        &self.fluids
    }
    /// Mutable access: the device state is downloaded first (the caller sees the current particles) and everything is
    /// uploaded again before the next step.
    pub fn fluids_mut(&mut self) -> Result<&mut FluidSet, Error> {
        self.sync()?;
        self.unpin_all(); // (the caller may push particles, replace or drop the Vecs: nothing stays page-locked behind its back)
        self.host_dirty = true;
        Ok(&mut self.fluids)
    }
    pub fn boundaries(&self) -> &BoundarySet {
        &self.boundaries
    }
    pub fn boundaries_mut(&mut self) -> &mut BoundarySet {
        self.host_dirty = true;
        &mut self.boundaries
    }

    /// Positions / velocities of every fluid, boundary volumes and forces: device -> host, if a step ran since the last time.
    pub fn sync(&mut self) -> Result<(), Error> {
        if !self.device_newer {
            return Ok(());
        }
        let nfluids = if self.decomposed { 0 } else { self.fluids.len() };  // host-order read-back does not exist in a decomposed run
        for (slot, fluid) in self.fluids.as_mut_slice().iter_mut().enumerate().take(nfluids) {
            if fluid.num_particles() == 0 {
                continue;
            }
            // Vec<Point3<f32>> / Vec<Vector3<f32>> are [x, y, z] f32 in memory.  The two Vecs are pinned in place (once per
            // allocation) so that the read-back is one scatter kernel + a DMA at PCIe speed (55 GB/s measured: 24 MB for 10^6
            // particles in ~0.45 ms) instead of the un-sort + unpack + copy of salva_hip_get_fluid (salva_hip_get_fluid_async).
            let bytes = fluid.num_particles() * 3 * std::mem::size_of::<f32>();
            let (pp, vp) = (fluid.positions.as_mut_ptr() as *mut f32, fluid.velocities.as_mut_ptr() as *mut f32);
            for ptr in [pp as usize, vp as usize] {
                if !self.pinned.iter().any(|&(p, b)| p == ptr && b == bytes) {
                    // (every path that can move a Vec went through unpin_all: an entry that overlaps without being equal cannot
                    // exist — the defensive sweep stays, it costs nothing)
                    self.pinned.retain(|&(p, b)| {
                        let overlap = p < ptr + bytes && ptr < p + b;
                        if overlap {
                            unsafe { ffi::salva_hip_host_unregister(p as *mut std::ffi::c_void) };
                        }
                        !overlap
                    });
                    if unsafe { ffi::salva_hip_host_register(self.raw, ptr as *mut std::ffi::c_void, bytes as u64) } == ffi::SALVA_HIP_OK {
                        self.pinned.push((ptr, bytes));
                    } // (else: the library falls back to its own pinned buffers for this array)
                }
            }
            check(unsafe { ffi::salva_hip_get_fluid_async(self.raw, slot as u32, pp, vp) })?;
            check(unsafe { ffi::salva_hip_wait_download(self.raw) })?;
        }
        for (slot, b) in self.boundaries.as_mut_slice().iter_mut().enumerate() {
            let n = b.num_particles();
            if n == 0 {
                continue;
            }
            let forces_ptr = match &mut b.forces {
                Some(lock) => {
                    let f = lock.get_mut().unwrap();
                    f.resize(n, Vector3::zeros());
                    f.as_mut_ptr() as *mut f32
                }
                None => std::ptr::null_mut(),
            };
            check(unsafe { ffi::salva_hip_get_boundary(self.raw, slot as u32, b.volumes.as_mut_ptr(), forces_ptr) })?;
        }
        self.device_newer = false;
        Ok(())
    }

    fn upload(&mut self) -> Result<(), Error> {
        if !self.host_dirty {
            return Ok(());
        }
        // in a decomposed run the fluids live on the device and change owner (dist.rs): only boundaries travel again
        let nfluids = if self.decomposed { 0 } else { self.fluids.len() };
        for (slot, fluid) in self.fluids.as_mut_slice().iter_mut().enumerate().take(nfluids) {
            // host-side half of apply_particles_removal + init_with_fluids (fluid.rs:88-98, dfsph_solver.rs:526-561): the
            // survivors' velocity_changes are fetched, filtered with the same mask and sent back with the particles
            let deleted: Vec<bool> = fluid.deleted_particles_mask().to_vec();
            if fluid.num_deleted_particles() != 0 {
                let mask: Vec<u8> = deleted.iter().map(|d| *d as u8).collect();
                if (slot as u32) < unsafe { ffi::salva_hip_num_fluids(self.raw) } {
                    let left = unsafe { ffi::salva_hip_delete_particles(self.raw, slot as u32, mask.as_ptr()) };
                    if left < 0 {
                        return check(left as i32);
                    }
                }
                // (Fluid::apply_particles_removal is pub(crate) upstream: the wrapper crate needs it `pub`, see README.md)
                fluid.apply_particles_removal();
            }
            let descs: Vec<ffi::SalvaHipForceDesc> = fluid
                .nonpressure_forces
                .iter()
                .map(|f| {
                    // `gpu_desc` is the one method the upstream trait gains (README.md): Some((kind, params)) for the built-ins,
                    // None for user forces, which then run on the host through the force callback (kind 7) at their place
                    // in the list
                    let (kind, p) = f.gpu_desc().unwrap_or((ffi::SALVA_HIP_FORCE_CUSTOM, [0.0; 7]));
                    ffi::SalvaHipForceDesc { kind, p }
                })
                .collect();
            let n = fluid.num_particles() as u64;
            check(unsafe {
                ffi::salva_hip_set_fluid(
                    self.raw,
                    slot as u32,
                    n,
                    fluid.positions.as_ptr() as *const f32,
                    fluid.velocities.as_ptr() as *const f32,
                    fluid.volumes.as_ptr(),
                    fluid.accelerations.as_ptr() as *const f32,
                    std::ptr::null(), // velocity_changes stay where they are on the device
                    fluid.density0,
                    fluid.interaction_groups.memberships.bits(),
                    fluid.interaction_groups.filter.bits(),
                    ffi::SALVA_HIP_DIRTY_ALL as u32,
                )
            })?;
            check(unsafe { ffi::salva_hip_set_fluid_forces(self.raw, slot as u32, descs.as_ptr(), descs.len() as u32) })?;
        }
        for (slot, b) in self.boundaries.as_slice().iter().enumerate() {
            check(unsafe {
                ffi::salva_hip_set_boundary(
                    self.raw,
                    slot as u32,
                    b.num_particles() as u64,
                    b.positions.as_ptr() as *const f32,
                    b.velocities.as_ptr() as *const f32,
                    b.interaction_groups.memberships.bits(),
                    b.interaction_groups.filter.bits(),
                    b.forces.is_some() as i32,
                )
            })?;
        }
        self.host_dirty = false;
        Ok(())
    }

    /// What `set_domain` (dist.rs) needs: everything the host holds goes up once; afterwards only boundaries are re-uploaded.
    pub(crate) fn upload_for_domain(&mut self) -> Result<(), Error> {
        self.upload()?;
        self.decomposed = true;
        Ok(())
    }

    /// `LiquidWorld::step(dt, gravity)` (liquid_world.rs:62-158 with the `()` coupling manager).
    pub fn step(&mut self, dt: Real, gravity: &Vector3<Real>) -> Result<(), Error> {
        self.upload()?;
        let g = [gravity.x, gravity.y, gravity.z];
        check(unsafe { ffi::salva_hip_step(self.raw, dt, g.as_ptr(), &mut self.last_stats) })?;
        self.device_newer = true;
        self.refresh_counters()?;
        if self.auto_sync {
            self.sync()?;
        }
        Ok(())
    }

    /// `world.counters.enable()` / `.disable()` (counters/mod.rs:56-72): the device library has to know, because the timers are
    /// HIP events it records inside the step.
    pub fn enable_counters(&mut self, enabled: bool) -> Result<(), Error> {
        if enabled {
            self.counters.enable();
        } else {
            self.counters.disable();
        }
        check(unsafe { ffi::salva_hip_enable_counters(self.raw, enabled as i32) })
    }

    /// Opt-in CFL sub-stepping: `TimestepManager::max_substep` (timestep_manager.rs:36-46) with the clamp the reference left
    /// commented out in `compute_substep` (:90-93).  `mode` 0 = off (one substep per step, the reference as it runs), 1 = the
    /// commented code literally, 2 = the same, cut at the remaining time.  Defaults of `TimestepManager::new`: 0.4, 1, 10.
    pub fn set_cfl_substepping(&mut self, mode: i32, cfl_coeff: Real, min_num_substeps: i32, max_num_substeps: i32) -> Result<(), Error> {
        check(unsafe { ffi::salva_hip_set_cfl(self.raw, mode, cfl_coeff, min_num_substeps, max_num_substeps) })?;
        self.cfl_mode = mode;
        Ok(())
    }

    /// 0 = one substep per step (the reference as it runs); see `set_cfl_substepping`.
    pub fn cfl_mode(&self) -> i32 {
        self.cfl_mode
    }

    /// Substep lengths of the last `step` (`counters.nsubsteps` of them).
    pub fn substeps(&self) -> Result<Vec<Real>, Error> {
        // (the count first: max_num_substeps is the caller's to choose)
        let n = unsafe { ffi::salva_hip_get_substeps(self.raw, std::ptr::null_mut(), 0) };
        if n < 0 {
            check(n as i32)?;
        }
        let mut v = vec![0.0 as Real; n as usize];
        if n > 0 {
            let m = unsafe { ffi::salva_hip_get_substeps(self.raw, v.as_mut_ptr(), v.len() as u64) };
            if m < 0 {
                check(m as i32)?;
            }
            v.truncate((m as usize).min(v.len()));
        }
        Ok(v)
    }

    fn refresh_counters(&mut self) -> Result<(), Error> {
        let mut c = std::mem::MaybeUninit::<ffi::SalvaHipCounters>::uninit();
        check(unsafe { ffi::salva_hip_get_counters(self.raw, c.as_mut_ptr()) })?;
        let c = unsafe { c.assume_init() };
        // (the reference's timers have no setter: the wrapper crate needs `Timer::set_time(ms)` or pub fields, README.md)
        self.counters.nsubsteps = c.nsubsteps as usize;
        self.counters.cd.ncontacts = c.cd.ncontacts as usize;
        self.counters.step_time.set_time(c.step_time);
        self.counters.custom.set_time(c.custom);
        self.counters.stages.collision_detection_time.set_time(c.stages.collision_detection_time);
        self.counters.stages.solver_time.set_time(c.stages.solver_time);
        self.counters.cd.boundary_update_time.set_time(c.cd.boundary_update_time);
        self.counters.cd.grid_insertion_time.set_time(c.cd.grid_insertion_time);
        self.counters.cd.neighborhood_search_time.set_time(c.cd.neighborhood_search_time);
        self.counters.solver.pressure_resolution_time.set_time(c.solver.pressure_resolution_time);
        Ok(())
    }

    /// Iteration counts, errors and contacts of the last step (not in the reference's API).
    pub fn last_step(&self) -> &ffi::SalvaHipStepStats {
        &self.last_stats
    }

    /// `LiquidWorld::particles_intersecting_shape` (liquid_world.rs:245-280), generic over parry's `Shape` like the reference's:
    /// `shape.compute_aabb(pos)` and `shape.distance_to_point(pos, &pt, true)` stay on the host (the two callbacks of
    /// `salva_hip_particles_intersecting_host_shape`), the cell filter and the particle scan run on the device.
    #[cfg(feature = "parry")]
    pub fn particles_intersecting_shape<S: ?Sized + parry3d::shape::Shape>(
        &mut self,
        pos: &na::Isometry3<Real>,
        shape: &S,
    ) -> Result<Vec<salva3d::object::ParticleId>, Error> {
        use parry3d::query::PointQuery;
        struct Ctx<'a, S: ?Sized> {
            pos: &'a na::Isometry3<Real>,
            shape: &'a S,
        }
        unsafe extern "C" fn aabb_cb<S: ?Sized + parry3d::shape::Shape>(user: *mut std::ffi::c_void, mins: *mut f32, maxs: *mut f32) {
            let c = &*(user as *const Ctx<S>);
            // (a panic must not unwind into C: an empty box makes the library refuse the query)
            let b = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| c.shape.compute_aabb(c.pos)));
            for k in 0..3 {
                *mins.add(k) = b.as_ref().map(|b| b.mins[k]).unwrap_or(f32::NAN);
                *maxs.add(k) = b.as_ref().map(|b| b.maxs[k]).unwrap_or(f32::NAN);
            }
        }
        unsafe extern "C" fn dist_cb<S: ?Sized + parry3d::shape::Shape>(user: *mut std::ffi::c_void, n: u32, pts: *const f32, out: *mut f32) {
            let c = &*(user as *const Ctx<S>);
            for k in 0..n as usize {
                let pt = na::Point3::new(*pts.add(3 * k), *pts.add(3 * k + 1), *pts.add(3 * k + 2));
                let d = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| c.shape.distance_to_point(c.pos, &pt, true)));
                *out.add(k) = d.unwrap_or(f32::INFINITY);
            }
        }
        let ctx = Ctx { pos, shape };
        let sh = ffi::SalvaHipHostQueryShape {
            aabb: Some(aabb_cb::<S>),
            distance: Some(dist_cb::<S>),
            user: &ctx as *const Ctx<S> as *mut std::ffi::c_void,
        };
        let mut cap = 1024usize;
        loop {
            let (mut k, mut s, mut i) = (vec![0u32; cap], vec![0u32; cap], vec![0u32; cap]);
            let total = unsafe {
                ffi::salva_hip_particles_intersecting_host_shape(self.raw, &sh, cap as u64, k.as_mut_ptr(), s.as_mut_ptr(), i.as_mut_ptr())
            };
            if total < 0 {
                check(total as i32)?;
            }
            let total = total as usize;
            if total > cap {
                cap = total;
                continue;
            }
            // dense slot -> handle, as the reference maps its grid entries (`get_from_contiguous_index`, liquid_world.rs:257, :266)
            return Ok((0..total)
                .filter_map(|j| {
                    if k[j] == 0 {
                        let (_, h) = self.fluids.get_from_contiguous_index(s[j] as usize)?;
                        Some(salva3d::object::ParticleId::FluidParticle(h, i[j] as usize))
                    } else {
                        let (_, h) = self.boundaries.get_from_contiguous_index(s[j] as usize)?;
                        Some(salva3d::object::ParticleId::BoundaryParticle(h, i[j] as usize))
                    }
                })
                .collect());
        }
    }

    /// The raw handle, for the entry points this wrapper does not cover (coupling, AABB / analytic-shape queries, multi-GPU).
    pub fn raw(&mut self) -> *mut ffi::SalvaHipWorld {
        self.raw
    }
}
