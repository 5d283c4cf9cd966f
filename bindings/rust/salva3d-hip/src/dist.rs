//! Multi-GPU: one process and one `LiquidWorld` per GPU, the domain cut into slabs of grid-cell planes along x
//! (include/salva_hip.h, "multi-GPU").  No counterpart in salva3d, which is single-process.
//!
//! SOURCE ONLY — never compiled (no Rust toolchain where it was written); mirrors `salva::Comm` of include/salva_hip.hpp and
//! `salva_amd/dist.py`, which run on hardware (tests/test_dist_gpu.py, tests/test_peer_transport_gpu.py).
//!
//! ```ignore
//! // every rank (process), after adding ITS particles to ITS world:
//! let id = if rank == 0 { Comm::rccl_unique_id()? } else { [0u8; 128] };
//! let id = mpi_broadcast(id);                                   // any side channel
//! let mut comm = Comm::rccl(rank, size, &id, device)?;          // or Comm::peer(rank, size, device, slot, |h| mpi_allgather(h))
//! world.set_domain(&mut comm, cell_lo, cell_hi, gid_offset)?;
//! loop { world.step(dt, &gravity)?; let mine = world.owned()?; /* render / write mine */ }
//! ```
use crate::ffi;
use crate::liquid_world::{check, Error, LiquidWorld};
use na::{Point3, Vector3};
use nalgebra as na;
use salva3d::math::Real;
use salva3d::object::FluidHandle;

/// A rank's handle on the slab exchange transport.  Destroy it after the worlds that use it.
pub struct Comm {
    raw: *mut ffi::SalvaHipComm,
    rank: i32,
    size: i32,
}

// the transport's calls are issued by the world that owns it, from whichever thread steps that world
unsafe impl Send for Comm {}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { ffi::salva_hip_comm_destroy(self.raw) }
    }
}

pub const PEER_HANDLE_BYTES: usize = ffi::SALVA_HIP_PEER_HANDLE_BYTES as usize;

impl Comm {
    pub fn rank(&self) -> i32 {
        self.rank
    }
    pub fn size(&self) -> i32 {
        self.size
    }

    /// RCCL over xGMI (the default transport): rank 0 creates the id, the caller distributes it.
    pub fn rccl_unique_id() -> Result<[u8; 128], Error> {
        let mut id = [0u8; 128];
        check(unsafe { ffi::salva_hip_comm_rccl_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    pub fn rccl(rank: i32, size: i32, id: &[u8; 128], device: i32) -> Result<Self, Error> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::salva_hip_comm_rccl_create(rank, size, id.as_ptr(), device, &mut raw) })?;
        Ok(Self { raw, rank, size })
    }

    /// xGMI peer-direct, the ranks of one node: flagged stores into hipIpc-mapped windows instead of collective-library calls.
    /// `all_gather(mine)` returns every rank's 64-byte handle in rank order (the caller's all-gather; it doubles as the barrier
    /// that makes every window exist before anybody writes to it).
    pub fn peer(
        rank: i32,
        size: i32,
        device: i32,
        slot_bytes: u64,
        all_gather: impl FnOnce([u8; PEER_HANDLE_BYTES]) -> Vec<[u8; PEER_HANDLE_BYTES]>,
    ) -> Result<Self, Error> {
        let mut mine = [0u8; PEER_HANDLE_BYTES];
        let mut setup = std::ptr::null_mut();
        check(unsafe { ffi::salva_hip_comm_peer_begin(rank, size, device, slot_bytes, mine.as_mut_ptr(), &mut setup) })?;
        let all = all_gather(mine);
        if all.len() != size as usize {
            unsafe { ffi::salva_hip_comm_peer_abort(setup) };
            return Err(Error { code: ffi::SALVA_HIP_E_INVALID, message: "peer transport: one handle per rank, in rank order".into() });
        }
        let flat: Vec<u8> = all.iter().flat_map(|h| h.iter().copied()).collect();
        let mut raw = std::ptr::null_mut();
        // consumes `setup`, also when it fails
        check(unsafe { ffi::salva_hip_comm_peer_connect(setup, flat.as_ptr(), &mut raw) })?;
        Ok(Self { raw, rank, size })
    }

    /// In-process loopback for tests: drive each rank from its own host thread.
    pub fn loopback(size: i32) -> Result<Vec<Self>, Error> {
        let mut raws = vec![std::ptr::null_mut(); size as usize];
        check(unsafe { ffi::salva_hip_comm_loopback_create(size, raws.as_mut_ptr()) })?;
        Ok(raws.into_iter().enumerate().map(|(r, raw)| Self { raw, rank: r as i32, size }).collect())
    }

    /// Collective check of the transport itself: patterned exchanges of `rounds` different lengths, the count exchange and both
    /// all-reduces.
    pub fn selftest(&mut self, max_bytes: u64, rounds: i32) -> Result<(), Error> {
        check(unsafe { ffi::salva_hip_comm_selftest(self.raw, max_bytes, rounds) })
    }
    /// Collective: (µs per exchange of `bytes` each way with both neighbours, µs per four-float all-reduce) — what one solver
    /// iteration of a decomposed run adds.
    pub fn time(&mut self, bytes: u64, iters: i32) -> Result<(f32, f32), Error> {
        let (mut a, mut b) = (0.0f32, 0.0f32);
        check(unsafe { ffi::salva_hip_comm_time(self.raw, bytes, iters, &mut a, &mut b) })?;
        Ok((a, b))
    }
}

/// The particles a rank owns after a step of a decomposed run (`salva_hip_get_owned`), in no particular order.
/// `LiquidWorld::local_view`
pub struct LocalView {
    pub ids: Vec<u32>,
    pub fluid_slots: Vec<u32>,
    pub is_ghost: Vec<u8>,
    pub positions: Vec<Point3<Real>>,
    pub velocities: Vec<Vector3<Real>>,
    pub densities: Vec<Real>,
    pub volumes: Vec<Real>,
}

pub struct OwnedParticles {
    /// global ids: `gid_offset` + upload index of the rank that created the particle
    pub gids: Vec<u32>,
    /// dense fluid slot (`fluids().as_slice()` order, the same on every rank)
    pub fluid_slots: Vec<u32>,
    pub positions: Vec<Point3<Real>>,
    pub velocities: Vec<Vector3<Real>>,
}

impl LiquidWorld {
    /// This world becomes the slab of cell planes `[cell_lo, cell_hi]` (cell = floor(x / h)) of a domain cut along x.  Call after
    /// adding this rank's fluids (the same fluid slots on every rank) and the boundary particles within three cells of its slab,
    /// before the first step.  From then on the host `FluidSet` is no longer refreshed (particles migrate between ranks): read
    /// them with `owned()`.
    pub fn set_domain(&mut self, comm: &mut Comm, cell_lo: i32, cell_hi: i32, gid_offset: u32) -> Result<(), Error> {
        self.upload_for_domain()?;
        check(unsafe { ffi::salva_hip_set_domain(self.raw(), comm.raw, cell_lo, cell_hi, gid_offset) })?;
        self.set_auto_sync(false);
        Ok(())
    }

    /// The working set as it is (`salva_hip_get_local`): every particle this world holds — on a rank of a decomposed run the
    /// owned particles and the ghosts — in the order of the last step's cell sort.  Also valid inside a force callback, where
    /// `velocities` are v + dv; that is what a user `NonPressureForce` works on in a decomposed run (`local_contacts`,
    /// `force_add_local_accelerations`).
    pub fn local_view(&mut self) -> Result<LocalView, Error> {
        let n = unsafe { ffi::salva_hip_local_len(self.raw()) } as usize;
        let mut v = LocalView {
            ids: vec![0; n],
            fluid_slots: vec![0; n],
            is_ghost: vec![0; n],
            positions: vec![Point3::origin(); n],
            velocities: vec![Vector3::zeros(); n],
            densities: vec![0.0; n],
            volumes: vec![0.0; n],
        };
        check(unsafe {
            ffi::salva_hip_get_local(
                self.raw(),
                v.ids.as_mut_ptr(),
                v.fluid_slots.as_mut_ptr(),
                v.is_ghost.as_mut_ptr(),
                v.positions.as_mut_ptr() as *mut f32,
                v.velocities.as_mut_ptr() as *mut f32,
                v.densities.as_mut_ptr(),
                v.volumes.as_mut_ptr(),
            )
        })?;
        Ok(v)
    }

    /// `ParticlesContacts` of every local particle as CSR (`salva_hip_get_local_contacts`): (offsets, j_model, j) with `j` a
    /// local index (fluid-fluid) or an index into boundary `j_model`'s arrays as this rank uploaded them.
    pub fn local_contacts(&mut self, boundary: bool) -> Result<(Vec<u64>, Vec<u32>, Vec<u32>), Error> {
        let n = unsafe { ffi::salva_hip_local_len(self.raw()) } as usize;
        let mut offsets = vec![0u64; n + 1];
        let total = unsafe { ffi::salva_hip_get_local_contacts(self.raw(), boundary as i32, offsets.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), 0) };
        if total < 0 {
            check(total as i32)?;
        }
        let (mut jm, mut j) = (vec![0u32; total as usize], vec![0u32; total as usize]);
        if total > 0 {
            let rc = unsafe { ffi::salva_hip_get_local_contacts(self.raw(), boundary as i32, offsets.as_mut_ptr(), jm.as_mut_ptr(), j.as_mut_ptr(), total as u64) };
            if rc < 0 {
                check(rc as i32)?;
            }
        }
        Ok((offsets, jm, j))
    }

    /// Inside a force callback: accelerations += `acc` (local order; what lands on a ghost is ignored).
    pub fn force_add_local_accelerations(&mut self, acc: &[Vector3<Real>]) -> Result<(), Error> {
        check(unsafe { ffi::salva_hip_force_add_local_accelerations(self.raw(), acc.as_ptr() as *const f32) })
    }

    pub fn owned(&mut self) -> Result<OwnedParticles, Error> {
        let mut cap = self.last_step().nparticles as usize + 1024;
        loop {
            let mut o = OwnedParticles {
                gids: vec![0; cap],
                fluid_slots: vec![0; cap],
                positions: vec![Point3::origin(); cap],
                velocities: vec![Vector3::zeros(); cap],
            };
            let m = unsafe {
                ffi::salva_hip_get_owned(
                    self.raw(),
                    cap as u32,
                    o.gids.as_mut_ptr(),
                    o.positions.as_mut_ptr() as *mut f32,
                    o.velocities.as_mut_ptr() as *mut f32,
                    o.fluid_slots.as_mut_ptr(),
                )
            };
            if m < 0 {
                check(m as i32)?;
            }
            let m = m as usize;
            if m <= cap {
                o.gids.truncate(m);
                o.fluid_slots.truncate(m);
                o.positions.truncate(m);
                o.velocities.truncate(m);
                return Ok(o);
            }
            cap = m;
        }
    }

    /// Collective re-cut of the slabs for equal particle counts; returns this rank's new `(cell_lo, cell_hi)` — the caller
    /// re-uploads the boundary particles the new slab needs.
    pub fn rebalance(&mut self) -> Result<(i32, i32), Error> {
        let (mut lo, mut hi) = (0i32, 0i32);
        check(unsafe { ffi::salva_hip_rebalance(self.raw(), &mut lo, &mut hi) })?;
        Ok((lo, hi))
    }

    /// Collective, between the same two steps on every rank (an empty slice where there is nothing to add): particles appended
    /// to this rank of a running decomposed world get the next free global ids (faucet3.rs:69-104-style emitters).
    pub fn add_owned(&mut self, fluid: FluidHandle, positions: &[Point3<Real>], velocities: Option<&[Vector3<Real>]>) -> Result<(), Error> {
        if let Some(v) = velocities {
            assert_eq!(v.len(), positions.len(), "The provided positions and velocities arrays must have the same length.");
        }
        let slot = self.fluids().iter().position(|(h, _)| h == fluid).expect("unknown fluid handle") as u32;
        let vel = match velocities {
            Some(v) if !positions.is_empty() => v.as_ptr() as *const f32,
            _ => std::ptr::null(),
        };
        let pos = if positions.is_empty() { std::ptr::null() } else { positions.as_ptr() as *const f32 };
        check(unsafe { ffi::salva_hip_add_particles(self.raw(), slot, positions.len() as u64, pos, vel) })
    }

    /// Collective: the listed particles this rank owns are gone from the next step on (ids owned elsewhere are ignored, so every
    /// rank may pass the same list).  Returns how many particles the rank still owns.
    pub fn delete_owned(&mut self, gids: &[u32]) -> Result<u64, Error> {
        let p = if gids.is_empty() { std::ptr::null() } else { gids.as_ptr() };
        let m = unsafe { ffi::salva_hip_delete_owned(self.raw(), gids.len() as u32, p) };
        if m < 0 {
            check(m as i32)?;
        }
        Ok(m as u64)
    }
}
