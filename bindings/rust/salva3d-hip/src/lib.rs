//! `salva3d-hip`: the public surface of `salva3d::LiquidWorld` (liquid_world.rs:39-280) over the C ABI of libsalva_hip.so
//! (include/salva_hip.h).  `examples3d/*.rs` switch with one line: `use salva3d_hip::LiquidWorld;`.
//!
//! Nothing here computes: `Fluid`, `Boundary`, the solver and force structs are salva3d's own; this crate uploads what the
//! host changed, calls `salva_hip_step`, and reads back what the host looks at.  There is no CPU fallback — without the
//! library or a device `LiquidWorld::new` returns the ABI's error.
pub mod ffi;
pub mod dist;
mod liquid_world;
#[cfg(feature = "rapier")]
pub mod coupling;

pub use dist::{Comm, OwnedParticles};
pub use liquid_world::{Error, GpuPressureSolver, LiquidWorld};
