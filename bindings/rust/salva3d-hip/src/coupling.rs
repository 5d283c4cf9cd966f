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
}

unsafe extern "C" fn host_aabb(user: *mut std::ffi::c_void, mins: *mut f32, maxs: *mut f32) {
    let ctx = &*(user as *const HostShapeCtx);
    let aabb = ctx.shape.compute_aabb(&ctx.position); // fluids_pipeline.rs:196-198 (the library loosens it)
    for a in 0..3 {
        *mins.add(a) = aabb.mins[a];
        *maxs.add(a) = aabb.maxs[a];
    }
}

unsafe extern "C" fn host_project(user: *mut std::ffi::c_void, n: u32, points: *const f32, projections: *mut f32, is_inside: *mut u8) {
    let ctx = &*(user as *const HostShapeCtx);
    for k in 0..n as usize {
        let pt = Point::new(*points.add(3 * k), *points.add(3 * k + 1), *points.add(3 * k + 2));
        let (proj, _feature) = ctx.shape.project_point_and_get_feature(&ctx.position, &pt); // :213-217
        for a in 0..3 {
            *projections.add(3 * k + a) = proj.point[a];
        }
        *is_inside.add(k) = proj.is_inside as u8;
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
#[derive(Default)]
pub struct ColliderCouplingSet {
    entries: HashMap<ColliderHandle, Entry>,
}

impl ColliderCouplingSet {
    pub fn new() -> Self {
        Self::default()
    }
    pub fn register_coupling(&mut self, boundary: BoundaryHandle, collider: ColliderHandle, sampling: ColliderSampling) -> Option<BoundaryHandle> {
        self.entries.insert(collider, Entry { sampling, boundary, uploaded: false, host_shape: None }).map(|e| e.boundary)
    }
    pub fn unregister_coupling(&mut self, collider: ColliderHandle) -> Option<BoundaryHandle> {
        self.entries.remove(&collider).map(|e| e.boundary)
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
        // ---- update_boundaries (:146-264): one pose per collider
        let mut poses: Vec<(u32, ffi::SalvaHipRigidPose, Option<rapier3d::dynamics::RigidBodyHandle>)> = Vec::new();
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
                                let ctx = Box::new(HostShapeCtx { shape: collider.shared_shape().clone(), position: *collider.position() });
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
        // ---- the substep
        self.liquid_world.step(dt, &na::Vector3::new(gravity.x, gravity.y, gravity.z))?;
        // ---- transmit_forces (:266-287): sum_i apply_impulse_at_point(f_i dt, x_i) = apply_impulse(F dt) + apply_torque_impulse(T dt)
        for (slot, pose, parent) in poses {
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
