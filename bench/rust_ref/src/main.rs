//! salva_rust_ref — the reference's own CPU path on the benchmark scene and on every golden scene.
//!
//!   cargo run --release -- [--side 100] [--steps 50] [--warmup 5] [--dump DIR]        (the bench.py tank, built here)
//!   cargo run --release -- --scene scenes/NAME.scene --dump DIR                        (a scene written by
//!       tests/golden/export_scenes.py: the seven golden scenes are committed under scenes/; `--bench 100` adds config 2)
//!
//! Scene mode dumps, after the first step (prefix s1_) and after the last (no prefix): pos_F.f32 / vel_F.f32 per fluid
//! (xyz, f32 LE), bvol_B.f32 per boundary, bforce_B.f32 for boundaries that receive forces, and ncontacts.txt (one line per
//! step).  tests/golden/compare_rust_dump.py --scene NAME DIR compares them with tests/golden/NAME.npz.
//!
//! Scene = bench.py `build_scene`: side^3 lattice block (spacing 2r, r = 0.025, the
//! positions of examples3d/helper.rs `cube_fluid`), jittered by +-0.1 r with the Numerical-Recipes LCG (seed 42),
//! resting in an open tank of lattice boundary particles (floor + 4 walls, one spacing outside the block),
//! rho0 = 1000, XSPHViscosity(0.5, 0), DFSPH defaults, dt = 1/200, g = -9.81 y.
//! Prints one JSON line with particle-steps/s; with --dump writes positions/velocities (f32 LE, xyz) after the run.
use nalgebra::{Point3, Vector3};
use salva3d::object::interaction_groups::{Group, InteractionGroups};
use salva3d::object::{Boundary, BoundaryHandle, Fluid, FluidHandle};
use salva3d::solver::{
    Akinci2013SurfaceTension, ArtificialViscosity, DFSPHSolver, DFSPHViscosity, He2014SurfaceTension, IISPHSolver, NonPressureForce,
    WCSPHSurfaceTension, XSPHViscosity,
};
use salva3d::LiquidWorld;
use std::io::Write;
use std::time::Instant;

const R: f32 = 0.025;

/// x_{n+1} = 1664525 x_n + 1013904223 (mod 2^32); uniform in [0, 1) from the top 24 bits — salva_amd/scenes.py `lcg_uniform`.
struct Lcg(u32);
impl Lcg {
    fn next(&mut self) -> f32 {
        self.0 = self.0.wrapping_mul(1664525).wrapping_add(1013904223);
        (self.0 >> 8) as f32 * (1.0 / 16777216.0)
    }
}

fn cube_fluid(ni: usize, nj: usize, nk: usize) -> Vec<Point3<f32>> {
    let half = Vector3::new(ni as f32 * R, nj as f32 * R, nk as f32 * R);
    let mut pts = Vec::with_capacity(ni * nj * nk);
    for i in 0..ni {
        for j in 0..nj {
            for k in 0..nk {
                let x = (i as f32) * R * 2.0;
                let y = (j as f32) * R * 2.0;
                let z = (k as f32) * R * 2.0;
                pts.push(Point3::new(x + R, y + R, z + R) - half);
            }
        }
    }
    pts
}

/// Lattice shell on the faces "xXyzZ" of [mins, maxs] (salva_amd/scenes.py `box_shell`), shared edges emitted once.
fn tank(side: usize) -> (Vec<Point3<f32>>, Vec<Point3<f32>>) {
    let d = 2.0 * R;
    let fluid = cube_fluid(side, side, side);
    let fmin = -(side as f32) * R + R;
    let fmax = fmin + (side as f32 - 1.0) * d;
    let mins = [fmin - d, fmin - d, fmin - d];
    let mut maxs = [fmax + d, fmax + d, fmax + d];
    maxs[1] += (std::cmp::max(side / 2, 4) as f32) * d;
    let n: Vec<i64> = (0..3).map(|a| (((maxs[a] - mins[a]) / d).round() as i64).max(1) + 1).collect();
    let mut seen = std::collections::HashSet::new();
    let mut shell = Vec::new();
    for (axis, hi) in [(0usize, false), (0, true), (1, false), (2, false), (2, true)] {
        let (u, v) = match axis { 0 => (1, 2), 1 => (0, 2), _ => (0, 1) };
        for a in 0..n[u] {
            for b in 0..n[v] {
                let mut q = [0i64; 3];
                q[axis] = if hi { n[axis] - 1 } else { 0 };
                q[u] = a;
                q[v] = b;
                if seen.insert(q) {
                    shell.push(Point3::new(mins[0] + q[0] as f32 * d, mins[1] + q[1] as f32 * d, mins[2] + q[2] as f32 * d));
                }
            }
        }
    }
    (fluid, shell)
}

// ------------------------------------------------------------------------------------------------ scene files
struct Reader {
    buf: Vec<u8>,
    at: usize,
}
impl Reader {
    fn u32(&mut self) -> u32 {
        let v = u32::from_le_bytes(self.buf[self.at..self.at + 4].try_into().unwrap());
        self.at += 4;
        v
    }
    fn f32(&mut self) -> f32 {
        f32::from_bits(self.u32())
    }
    fn f32s(&mut self, n: usize) -> Vec<f32> {
        (0..n).map(|_| self.f32()).collect()
    }
    fn points(&mut self, n: usize) -> Vec<Point3<f32>> {
        (0..n).map(|_| { let x = self.f32(); let y = self.f32(); let z = self.f32(); Point3::new(x, y, z) }).collect()
    }
    fn vectors(&mut self, n: usize) -> Vec<Vector3<f32>> {
        (0..n).map(|_| { let x = self.f32(); let y = self.f32(); let z = self.f32(); Vector3::new(x, y, z) }).collect()
    }
}

fn build_force(kind: u32, p: &[f32]) -> Box<dyn NonPressureForce> {
    match kind {
        1 => Box::new(XSPHViscosity::new(p[0], p[1])),
        2 => {
            let mut v = ArtificialViscosity::new(p[0], p[1]);
            v.alpha = p[2];
            v.beta = p[3];
            v.speed_of_sound = p[4];
            Box::new(v)
        }
        3 => Box::new(Akinci2013SurfaceTension::new(p[0], p[1])),
        4 => {
            let mut v = DFSPHViscosity::new(p[0]);
            v.min_viscosity_iter = p[1] as usize;
            v.max_viscosity_iter = p[2] as usize;
            v.max_viscosity_error = p[3];
            Box::new(v)
        }
        5 => Box::new(He2014SurfaceTension::new(p[0], p[1])),
        6 => Box::new(WCSPHSurfaceTension::new(p[0], p[1])),
        k => panic!("unknown force kind {}", k),
    }
}

fn write_f32(path: String, it: impl Iterator<Item = f32>) {
    let mut w = std::io::BufWriter::new(std::fs::File::create(path).unwrap());
    for v in it {
        w.write_all(&v.to_le_bytes()).unwrap();
    }
}

fn dump_state(world: &LiquidWorld, fluids: &[FluidHandle], bounds: &[BoundaryHandle], dir: &str, prefix: &str) {
    for (k, h) in fluids.iter().enumerate() {
        let f = world.fluids().get(*h).unwrap();
        write_f32(format!("{}/{}pos_{}.f32", dir, prefix, k), f.positions.iter().flat_map(|p| [p.x, p.y, p.z]));
        write_f32(format!("{}/{}vel_{}.f32", dir, prefix, k), f.velocities.iter().flat_map(|v| [v.x, v.y, v.z]));
    }
    for (k, h) in bounds.iter().enumerate() {
        let b = world.boundaries().get(*h).unwrap();
        write_f32(format!("{}/{}bvol_{}.f32", dir, prefix, k), b.volumes.iter().cloned());
        if let Some(forces) = &b.forces {
            let forces = forces.read().unwrap();
            write_f32(format!("{}/{}bforce_{}.f32", dir, prefix, k), forces.iter().flat_map(|v| [v.x, v.y, v.z]));
        }
    }
}

fn run_scene(path: &str, dump: Option<String>) {
    let mut r = Reader { buf: std::fs::read(path).expect("scene file"), at: 0 };
    assert_eq!(&r.buf[0..8], b"SLVSCN01", "not a scene file");
    r.at = 8;
    let radius = r.f32();
    let smoothing = r.f32();
    let solver_kind = r.u32();
    let (min_p, max_p, max_derr) = (r.u32() as usize, r.u32() as usize, r.f32());
    let (min_d, max_d, max_diverr) = (r.u32() as usize, r.u32() as usize, r.f32());
    let nsteps = r.u32() as usize;
    let dt = r.f32();
    let g = Vector3::new(r.f32(), r.f32(), r.f32());
    let mut world = if solver_kind == 0 {
        let mut s: DFSPHSolver = DFSPHSolver::new();
        s.min_pressure_iter = min_p;
        s.max_pressure_iter = max_p;
        s.max_density_error = max_derr;
        s.min_divergence_iter = min_d;
        s.max_divergence_iter = max_d;
        s.max_divergence_error = max_diverr;
        LiquidWorld::new(s, radius, smoothing)
    } else {
        let mut s: IISPHSolver = IISPHSolver::new();
        s.min_pressure_iter = min_p;
        s.max_pressure_iter = max_p;
        s.max_density_error = max_derr;
        LiquidWorld::new(s, radius, smoothing)
    };
    let mut fluids = Vec::new();
    let mut nparticles = 0usize;
    for _ in 0..r.u32() {
        let n = r.u32() as usize;
        let density0 = r.f32();
        let groups = InteractionGroups::new(Group::from_bits_retain(r.u32()), Group::from_bits_retain(r.u32()));
        let (has_vel, has_vol, nforces) = (r.u32() != 0, r.u32() != 0, r.u32());
        let mut forces = Vec::new();
        for _ in 0..nforces {
            let kind = r.u32();
            let np = r.u32() as usize;
            let params = r.f32s(np);
            forces.push(build_force(kind, &params));
        }
        let mut fluid = Fluid::new(r.points(n), radius, density0, groups);
        if has_vel {
            fluid.velocities = r.vectors(n);
        }
        if has_vol {
            fluid.volumes = r.f32s(n);
        }
        fluid.nonpressure_forces = forces;
        nparticles += n;
        fluids.push(world.add_fluid(fluid));
    }
    let mut bounds = Vec::new();
    for _ in 0..r.u32() {
        let n = r.u32() as usize;
        let groups = InteractionGroups::new(Group::from_bits_retain(r.u32()), Group::from_bits_retain(r.u32()));
        let (wants_forces, has_vel) = (r.u32() != 0, r.u32() != 0);
        let mut b = Boundary::new(r.points(n), groups);
        if has_vel {
            b.velocities = r.vectors(n);
        }
        if wants_forces {
            b.forces = Some(std::sync::RwLock::new(Vec::new()));  // as the coupling does for dynamic bodies (fluids_pipeline.rs:168-169)
            b.clear_forces(true);
        }
        bounds.push(world.add_boundary(b));
    }
    assert_eq!(r.at, r.buf.len(), "trailing bytes in the scene file");
    if let Some(dir) = &dump {
        std::fs::create_dir_all(dir).unwrap();
    }
    let mut ncontacts = Vec::new();
    let t0 = Instant::now();
    for step in 0..nsteps {
        // (boundary.forces is only ever cleared by a coupling manager, fluids_pipeline.rs:169 / :258: with the `()` manager of
        // LiquidWorld::step it accumulates over the steps, and the golden fixtures hold that sum)
        world.step(dt, &g);
        ncontacts.push(world.counters.cd.ncontacts);
        if step == 0 {
            if let Some(dir) = &dump {
                dump_state(&world, &fluids, &bounds, dir, "s1_");
            }
        }
    }
    let el = t0.elapsed().as_secs_f64();
    if let Some(dir) = &dump {
        dump_state(&world, &fluids, &bounds, dir, "");
        let mut w = std::fs::File::create(format!("{}/ncontacts.txt", dir)).unwrap();
        for c in &ncontacts {
            writeln!(w, "{}", c).unwrap();
        }
    }
    println!(
        "{{\"scene\": \"{}\", \"value\": {:.1}, \"unit\": \"particle-steps/s\", \"kind\": \"reference\", \"threads\": {}, \"particles\": {}, \"steps\": {}, \"ms_per_step\": {:.3}}}",
        path,
        nparticles as f64 * nsteps as f64 / el,
        std::thread::available_parallelism().map(|v| v.get()).unwrap_or(1),
        nparticles, nsteps, el / nsteps as f64 * 1e3
    );
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if let Some(path) = args.iter().position(|a| a == "--scene").and_then(|i| args.get(i + 1)).cloned() {
        let dump = args.iter().position(|a| a == "--dump").and_then(|i| args.get(i + 1)).cloned();
        run_scene(&path, dump);
        return;
    }
    let get = |name: &str, default: usize| -> usize {
        args.iter().position(|a| a == name).and_then(|i| args.get(i + 1)).and_then(|v| v.parse().ok()).unwrap_or(default)
    };
    let side = get("--side", 100);
    let steps = get("--steps", 50);
    let warmup = get("--warmup", 5);
    let dump = args.iter().position(|a| a == "--dump").and_then(|i| args.get(i + 1)).cloned();

    let (mut fluid_pts, shell) = tank(side);
    let mut lcg = Lcg(42);
    for p in fluid_pts.iter_mut() {
        // same order as numpy: u has the shape of `positions` (row-major: x, y, z of particle 0, then particle 1 ...)
        for a in 0..3 {
            p[a] += (lcg.next() * 2.0 - 1.0) * (0.1 * R);
        }
    }
    let n = fluid_pts.len();
    let solver: DFSPHSolver = DFSPHSolver::new();  // default kernels (CubicSplineKernel), as fluids_pipeline.rs:35
    let mut world = LiquidWorld::new(solver, R, 2.0);
    let mut fluid = Fluid::new(fluid_pts, R, 1000.0, InteractionGroups::default());
    fluid.nonpressure_forces.push(Box::new(XSPHViscosity::new(0.5, 0.0)));
    let fh = world.add_fluid(fluid);
    world.add_boundary(Boundary::new(shell.clone(), InteractionGroups::default()));

    let g = Vector3::new(0.0, -9.81, 0.0);
    let dt = 1.0 / 200.0;
    for _ in 0..warmup {
        world.step(dt, &g);
    }
    let t0 = Instant::now();
    for _ in 0..steps {
        world.step(dt, &g);
    }
    let el = t0.elapsed().as_secs_f64();
    println!(
        "{{\"metric\": \"particle-steps/sec (3D DFSPH)\", \"value\": {:.1}, \"unit\": \"particle-steps/s\", \"kind\": \"reference\", \
         \"threads\": {}, \"particles\": {}, \"boundary_particles\": {}, \"steps\": {}, \"warmup\": {}, \"ms_per_step\": {:.3}}}",
        n as f64 * steps as f64 / el,
        std::thread::available_parallelism().map(|v| v.get()).unwrap_or(1),
        n, shell.len(), steps, warmup, el / steps as f64 * 1e3
    );
    if let Some(dir) = dump {
        let f = world.fluids().get(fh).unwrap();
        std::fs::create_dir_all(&dir).unwrap();
        let mut w = std::fs::File::create(format!("{}/positions.f32", dir)).unwrap();
        for p in &f.positions { for a in 0..3 { w.write_all(&p[a].to_le_bytes()).unwrap(); } }
        let mut w = std::fs::File::create(format!("{}/velocities.f32", dir)).unwrap();
        for v in &f.velocities { for a in 0..3 { w.write_all(&v[a].to_le_bytes()).unwrap(); } }
    }
}
