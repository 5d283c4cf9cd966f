//! The rapier coupling (reference: src/integrations/rapier/fluids_pipeline.rs:18-288) over the pose / wrench entry points:
//! per step ONE `SalvaHipRigidPose` goes to the device per coupled collider and ONE wrench comes back, instead of every
//! boundary particle up and every force down.  `FluidsPipeline`, `ColliderCouplingSet`, `ColliderSampling` keep the
//! reference's names and meaning.  SOURCE ONLY — never compiled.
use crate::ffi;
use crate::{Error, LiquidWorld};
use nalgebra as na;
use rapier3d::dynamics::RigidBodySet;
use rapier3d::geometry::{ColliderHandle, ColliderSet};
use salva3d::math::{Point, Real, Vector};
use salva3d::object::BoundaryHandle;
use salva3d::solver::DFSPHSolver;
use std::collections::HashMap;

/// fluids_pipeline.rs:36-43.
pub enum ColliderSampling {
    /// Collider-local sample points (kept on the device: `salva_hip_set_boundary_sampling`).
    StaticSampling(Vec<Point<Real>>),
    /// Boundary particles = projections of the nearby fluid particles onto the collider, recomputed inside every step
    /// (`salva_hip_set_boundary_dynamic_sampling` for Ball, Cuboid, Capsule (y) and Cylinder colliders — geometry on the
    /// device; every other parry shape through `salva_hip_set_boundary_dynamic_sampling_host`: the loop on the device, the
    /// shape's `compute_aabb` / `project_point_and_get_feature` called back here once per step).
    DynamicContactSampling,
}

/// What the host-shape callbacks see: the collider's shape and its pose as of this step's `update_boundaries`.
struct HostShapeCtx {
    shape: rapier3d::geometry::SharedShape,
    position: na::Isometry3<Real>,
    panicked: std::sync::atomic::AtomicBool,
}

// A panic must not unwind across `extern "C"` (UB): both callbacks run their bodies under catch_unwind.  A panicking aabb
// hands the library a NaN box, which fails the step with SALVA_HIP_E_INVALID; a panicking projection leaves the points where
// they are and outside the shape, and sets `panicked`, which `FluidsPipeline::step` turns into an `Err` after the step.
unsafe extern "C" fn host_aabb(user: *mut std::ffi::c_void, mins: *mut f32, maxs: *mut f32) {
    let ctx = &*(user as *const HostShapeCtx);
    let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| ctx.shape.compute_aabb(&ctx.position))); // fluids_pipeline.rs:196-198 (the library loosens it)
    for a in 0..3 {
        match &r {
            Ok(aabb) => {
                *mins.add(a) = aabb.mins[a];
                *maxs.add(a) = aabb.maxs[a];
            }
            Err(_) => {
                *mins.add(a) = f32::NAN;
                *maxs.add(a) = f32::NAN;
            }
        }
    }
    if r.is_err() {
        ctx.panicked.store(true, std::sync::atomic::Ordering::Relaxed);
    }
}

unsafe extern "C" fn host_project(user: *mut std::ffi::c_void, n: u32, points: *const f32, projections: *mut f32, is_inside: *mut u8) {
    let ctx = &*(user as *const HostShapeCtx);
    for k in 0..n as usize {
        let pt = Point::new(*points.add(3 * k), *points.add(3 * k + 1), *points.add(3 * k + 2));
        let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| ctx.shape.project_point_and_get_feature(&ctx.position, &pt))); // :213-217
        match r {
            Ok((proj, _feature)) => {
                for a in 0..3 {
                    *projections.add(3 * k + a) = proj.point[a];
                }
                *is_inside.add(k) = proj.is_inside as u8;
            }
            Err(_) => {
                for a in 0..3 {
                    *projections.add(3 * k + a) = pt[a];
                }
                *is_inside.add(k) = 0;
                ctx.panicked.store(true, std::sync::atomic::Ordering::Relaxed);
            }
        }
    }
}

struct Entry {
    sampling: ColliderSampling,
    boundary: BoundaryHandle,
    uploaded: bool,
    /// `Some` for a collider whose shape is projected on the host; boxed so that the pointer the library holds stays valid
    host_shape: Option<Box<HostShapeCtx>>,
}

/// fluids_pipeline.rs:64-136.
///
/// The library keeps the sampling method of an uploaded entry — for a host shape: two function pointers and the address of
/// the entry's `HostShapeCtx` — and calls it in every step until `salva_hip_clear_boundary_sampling` detaches it.  An entry that
/// is unregistered or replaced after its upload therefore moves to `retired`: its context stays alive there until the next
/// `FluidsPipeline::step` (the first place that can reach the world) has detached the boundary.
#[derive(Default)]
pub struct ColliderCouplingSet {
    entries: HashMap<ColliderHandle, Entry>,
    retired: Vec<Entry>,
}

impl ColliderCouplingSet {
    pub fn new() -> Self {
        Self::default()
    }
    pub fn register_coupling(&mut self, boundary: BoundaryHandle, collider: ColliderHandle, sampling: ColliderSampling) -> Option<BoundaryHandle> {
        let old = self.entries.insert(collider, Entry { sampling, boundary, uploaded: false, host_shape: None });
        old.map(|e| self.retire(e))
    }
    pub fn unregister_coupling(&mut self, collider: ColliderHandle) -> Option<BoundaryHandle> {
        let old = self.entries.remove(&collider);
        old.map(|e| self.retire(e))
    }
    fn retire(&mut self, e: Entry) -> BoundaryHandle {
        let b = e.boundary;
        if e.uploaded {
            self.retired.push(e); // (keeps the host-shape context alive until the library has forgotten it)
        }
        b
    }
}

fn check(code: i32) -> Result<(), Error> {
    if code == ffi::SALVA_HIP_OK {
        Ok(())
    } else {
        let m = unsafe { std::ffi::CStr::from_ptr(ffi::salva_hip_last_error()) }.to_string_lossy().into_owned();
        Err(Error { code, message: m })
    }
}

/// a coupled collider's boundary slot, the pose it was last given, and its parent body
type Pose = (u32, ffi::SalvaHipRigidPose, Option<rapier3d::dynamics::RigidBodyHandle>);

/// fluids_pipeline.rs:18-61: the liquid world (DFSPH, :35) and the couplings.
pub struct FluidsPipeline {
    pub liquid_world: LiquidWorld,
    pub coupling: ColliderCouplingSet,
}

impl FluidsPipeline {
    pub fn new(particle_radius: Real, smoothing_factor: Real) -> Result<Self, Error> {
        let dfsph: DFSPHSolver = DFSPHSolver::new();
        Ok(Self { liquid_world: LiquidWorld::new(dfsph, particle_radius, smoothing_factor)?, coupling: ColliderCouplingSet::new() })
    }

    /// `step(gravity, dt, colliders, bodies)` (fluids_pipeline.rs:48-60) = update_boundaries -> the substep -> transmit_forces.
    pub fn step(&mut self, gravity: &Vector<Real>, dt: Real, colliders: &ColliderSet, bodies: &mut RigidBodySet) -> Result<(), Error> {
        // ---- entries unregistered or replaced since the last step: detach their boundaries in the library FIRST (it would call
        // a retired host shape's callbacks otherwise), then let their contexts go.  The boundary keeps its last particles, as the
        // reference's does after unregister_coupling (fluids_pipeline.rs:116-125).
        while let Some(e) = self.coupling.retired.last() {
            // (an entry leaves `retired` only after its detach has succeeded: an early return keeps its context alive)
            let still_coupled = self.coupling.entries.values().any(|o| o.boundary == e.boundary && o.uploaded);
            if let (false, Some(slot)) = (still_coupled, self.liquid_world.boundaries().iter().position(|(h, _)| h == e.boundary)) {
                check(unsafe { ffi::salva_hip_clear_boundary_sampling(self.liquid_world.raw(), slot as u32) })?;
            }
            self.coupling.retired.pop();
        }
        if self.liquid_world.cfl_mode() != 0 {
            return self.step_substepping(gravity, dt, colliders, bodies);
        }
        let poses = self.update_boundaries(colliders, bodies)?;
        // ---- the substep
        let stepped = self.liquid_world.step(dt, &na::Vector3::new(gravity.x, gravity.y, gravity.z));
        self.check_host_shapes()?;
        stepped?;
        self.transmit_forces(&poses, bodies, dt)
    }

    /// CFL sub-stepping (`LiquidWorld::set_cfl_substepping`): the reference's manager is called INSIDE the substep loop
    /// (liquid_world.rs:94-103 `update_boundaries`, :146 `transmit_forces`), so that each substep's impulse (force * timestep.dt(),
    /// fluids_pipeline.rs:266-287) reaches the bodies before the next substep samples their velocities.  The library calls back at
    /// both points (`salva_hip_set_coupling_callback`).
    fn step_substepping(&mut self, gravity: &Vector<Real>, dt: Real, colliders: &ColliderSet, bodies: &mut RigidBodySet) -> Result<(), Error> {
        struct Ctx<'a> {
            pipeline: *mut FluidsPipeline,
            colliders: &'a ColliderSet,
            bodies: *mut RigidBodySet,
            poses: Vec<Pose>,
            error: Option<Error>,
        }
        unsafe extern "C" fn cb(user: *mut std::ffi::c_void, _world: *mut ffi::SalvaHipWorld, phase: i32, sub_dt: f32) -> i32 {
            let c = &mut *(user as *mut Ctx);
            // (a panic must not unwind into C)
            let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
                let p = &mut *c.pipeline;
                if phase == 0 {
                    p.update_boundaries(c.colliders, &*c.bodies).map(|poses| c.poses = poses)
                } else {
                    p.transmit_forces(&c.poses, &mut *c.bodies, sub_dt)
                }
            }));
            match r {
                Ok(Ok(())) => 0,
                Ok(Err(e)) => {
                    c.error = Some(e);
                    1
                }
                Err(_) => {
                    c.error = Some(Error { code: ffi::SALVA_HIP_E_INVALID, message: "the coupling callback panicked".into() });
                    1
                }
            }
        }
        let mut ctx = Ctx { pipeline: self as *mut FluidsPipeline, colliders, bodies: bodies as *mut RigidBodySet, poses: Vec::new(), error: None };
        let raw = self.liquid_world.raw();
        check(unsafe { ffi::salva_hip_set_coupling_callback(raw, Some(cb), &mut ctx as *mut Ctx as *mut std::ffi::c_void) })?;
        // (the callback reaches `self` through the raw pointer while `step` holds `&mut self.liquid_world`: the library calls it on
        // this thread, inside salva_hip_step, and the step touches nothing of the Rust-side mirrors until it returns)
        let stepped = self.liquid_world.step(dt, &na::Vector3::new(gravity.x, gravity.y, gravity.z));
        unsafe { ffi::salva_hip_set_coupling_callback(raw, None, std::ptr::null_mut()) };
        self.check_host_shapes()?;
        if let Some(e) = ctx.error.take() {
            return Err(e);
        }
        stepped
    }

    fn check_host_shapes(&self) -> Result<(), Error> {
        for entry in self.coupling.entries.values() {
            if let Some(ctx) = entry.host_shape.as_ref() {
                if ctx.panicked.swap(false, std::sync::atomic::Ordering::Relaxed) {
                    return Err(Error { code: ffi::SALVA_HIP_E_INVALID, message: "a host-shape callback (compute_aabb / project_point) panicked during the step".into() });
                }
            }
        }
        Ok(())
    }

    /// `update_boundaries` (fluids_pipeline.rs:146-264): one pose per collider
    fn update_boundaries(&mut self, colliders: &ColliderSet, bodies: &RigidBodySet) -> Result<Vec<Pose>, Error> {
        let mut poses: Vec<Pose> = Vec::new();
        for (co_handle, entry) in self.coupling.entries.iter_mut() {
            let (Some(collider), Some(slot)) = (colliders.get(*co_handle), self.liquid_world.boundaries().iter().position(|(h, _)| h == entry.boundary)) else {
                continue;
            };
            let slot = slot as u32;
            let raw = self.liquid_world.raw();
            if !entry.uploaded {
                let groups = self.liquid_world.boundaries().get(entry.boundary).unwrap().interaction_groups;
                match &entry.sampling {
                    ColliderSampling::StaticSampling(points) => check(unsafe {
                        ffi::salva_hip_set_boundary_sampling(raw, slot, points.len() as u64, points.as_ptr() as *const f32, groups.memberships.bits(), groups.filter.bits())
                    })?,
                    ColliderSampling::DynamicContactSampling => {
                        let builtin = if let Some(b) = collider.shape().as_ball() {
                            Some(ffi::SalvaHipShape { kind: ffi::SALVA_HIP_SHAPE_BALL, params: [b.radius, 0.0, 0.0] })
                        } else if let Some(c) = collider.shape().as_cuboid() {
                            Some(ffi::SalvaHipShape { kind: ffi::SALVA_HIP_SHAPE_CUBOID, params: [c.half_extents.x, c.half_extents.y, c.half_extents.z] })
                        } else if let Some(c) = collider.shape().as_capsule().filter(|c| c.segment.a.x == 0.0 && c.segment.a.z == 0.0 && c.segment.b == -c.segment.a) {
                            // Capsule::new_y; a capsule about another axis goes the host way below
                            Some(ffi::SalvaHipShape { kind: ffi::SALVA_HIP_SHAPE_CAPSULE, params: [c.segment.b.y, c.radius, 0.0] })
                        } else if let Some(c) = collider.shape().as_cylinder() {
                            Some(ffi::SalvaHipShape { kind: ffi::SALVA_HIP_SHAPE_CYLINDER, params: [c.half_height, c.radius, 0.0] })
                        } else {
                            None
                        };
                        match builtin {
                            Some(shape) => check(unsafe { ffi::salva_hip_set_boundary_dynamic_sampling(raw, slot, &shape, groups.memberships.bits(), groups.filter.bits()) })?,
                            None => {
                                // any other parry shape: the two parry calls of the loop stay here
                                let ctx = Box::new(HostShapeCtx { shape: collider.shared_shape().clone(), position: *collider.position(), panicked: Default::default() });
                                let host = ffi::SalvaHipHostShape {
                                    aabb: Some(host_aabb),
                                    project: Some(host_project),
                                    user: &*ctx as *const HostShapeCtx as *mut std::ffi::c_void,
                                };
                                check(unsafe { ffi::salva_hip_set_boundary_dynamic_sampling_host(raw, slot, &host, groups.memberships.bits(), groups.filter.bits()) })?;
                                entry.host_shape = Some(ctx);
                            }
                        }
                    }
                }
                entry.uploaded = true;
            }
            let iso = collider.position();
            if let Some(ctx) = entry.host_shape.as_mut() {
                ctx.position = *iso; // what host_aabb / host_project apply during the step below
            }
            let q = iso.rotation.coords; // (i, j, k, w): nalgebra's storage order
            let body = collider.parent().and_then(|p| bodies.get(p).map(|b| (p, b)));
            let mut pose = ffi::SalvaHipRigidPose {
                translation: [iso.translation.x, iso.translation.y, iso.translation.z],
                rotation: [q.x, q.y, q.z, q.w],
                linvel: [0.0; 3],
                angvel: [0.0; 3],
                world_com: [0.0; 3],
                has_body: 0,
                is_dynamic: 0,
            };
            if let Some((_, b)) = body {
                let (lv, av, com) = (b.linvel(), b.angvel(), b.center_of_mass());
                pose.linvel = [lv.x, lv.y, lv.z];
                pose.angvel = [av.x, av.y, av.z];
                pose.world_com = [com.x, com.y, com.z];
                pose.has_body = 1;
                pose.is_dynamic = b.is_dynamic() as i32;
            }
            check(unsafe { ffi::salva_hip_update_boundary_pose(raw, slot, &pose) })?;
            poses.push((slot, pose, body.map(|(p, _)| p)));
        }
        Ok(poses)
    }

    /// `transmit_forces` (fluids_pipeline.rs:266-287): sum_i apply_impulse_at_point(f_i dt, x_i) = apply_impulse(F dt) + apply_torque_impulse(T dt)
    fn transmit_forces(&mut self, poses: &[Pose], bodies: &mut RigidBodySet, dt: Real) -> Result<(), Error> {
        for (slot, pose, parent) in poses.iter().copied() {
            let Some(parent) = parent else { continue };
            if pose.is_dynamic == 0 {
                continue;
            }
            let (mut f, mut t) = ([0.0f32; 3], [0.0f32; 3]);
            check(unsafe { ffi::salva_hip_get_boundary_wrench(self.liquid_world.raw(), slot, pose.world_com.as_ptr(), f.as_mut_ptr(), t.as_mut_ptr()) })?;
            if let Some(body) = bodies.get_mut(parent) {
                body.apply_impulse(Vector::new(f[0], f[1], f[2]) * dt, true);
                body.apply_torque_impulse(Vector::new(t[0], t[1], t[2]) * dt, true);
            }
        }
        Ok(())
    }
}
