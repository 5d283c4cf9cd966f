"""Synthetic scene generators (numpy, float32 arithmetic).

`cube_fluid` restates /root/reference/examples3d/helper.rs:4-20 operation for operation; the boundary
generators replace the reference's parry ray-casting sampler (src/sampling/ray_sampling.rs, out of scope)
with plain lattice shells at the same 2r spacing.  The LCG generators implement the seeds named in
SURVEY.md §8(d) (42 for position jitter, 12345 for velocities).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def cube_fluid_positions(ni: int, nj: int, nk: int, particle_rad: float) -> np.ndarray:
    """examples3d/helper.rs:4-20 — ni*nj*nk lattice, spacing 2r, centred on the origin, i-major / k-minor."""
    r = F(particle_rad)
    i = np.arange(ni, dtype=F)[:, None, None]
    j = np.arange(nj, dtype=F)[None, :, None]
    k = np.arange(nk, dtype=F)[None, None, :]
    half = np.array([F(ni) * r, F(nj) * r, F(nk) * r], dtype=F)
    x = (i * r) * F(2.0)
    y = (j * r) * F(2.0)
    z = (k * r) * F(2.0)
    pts = np.empty((ni, nj, nk, 3), dtype=F)
    pts[..., 0] = (x + r) - half[0]
    pts[..., 1] = (y + r) - half[1]
    pts[..., 2] = (z + r) - half[2]
    return pts.reshape(-1, 3)


def lcg_uniform(n: int, seed: int) -> np.ndarray:
    """Deterministic uniforms in [0, 1): Numerical-Recipes 32-bit LCG, top 24 bits.

    State n is a closed form of state 0 (a^n * s + c * (a^n - 1)/(a - 1) mod 2^32) evaluated with a
    vectorised doubling scan, so 10^7 values cost milliseconds and the sequence is a property of
    (seed, index) only.
    """
    a, c = np.uint64(1664525), np.uint64(1013904223)
    mask = np.uint64(0xFFFFFFFF)
    # affine maps x -> A x + B composed by doubling: (A, B) for 2^k steps
    idx = np.arange(1, n + 1, dtype=np.uint64)
    A = np.ones(n, dtype=np.uint64)
    B = np.zeros(n, dtype=np.uint64)
    pa, pb = a, c
    bit = 0
    with np.errstate(over="ignore"):
        while (1 << bit) <= n:
            sel = ((idx >> np.uint64(bit)) & np.uint64(1)).astype(bool)
            # apply (pa, pb) after the current map on the selected lanes
            A[sel] = (A[sel] * pa) & mask
            B[sel] = (B[sel] * pa + pb) & mask
            pb = (pb * pa + pb) & mask
            pa = (pa * pa) & mask
            bit += 1
        x = (A * np.uint64(seed & 0xFFFFFFFF) + B) & mask
    return ((x >> np.uint64(8)).astype(np.float64) / float(1 << 24)).astype(F)


def jitter(positions: np.ndarray, amplitude: float, seed: int = 42) -> np.ndarray:
    u = lcg_uniform(positions.size, seed).reshape(positions.shape)
    return (positions + (u * F(2.0) - F(1.0)) * F(amplitude)).astype(F)


def random_velocities(n: int, amplitude: float, seed: int = 12345) -> np.ndarray:
    u = lcg_uniform(3 * n, seed).reshape(n, 3)
    return ((u * F(2.0) - F(1.0)) * F(amplitude)).astype(F)


def box_shell(mins, maxs, particle_rad: float, faces: str = "xXyYzZ", layers: int = 1) -> np.ndarray:
    """Boundary particles on the faces of an axis-aligned box, lattice spacing 2r.

    `faces` selects which faces are sampled (x = min-x face, X = max-x face, ...).  With `layers` > 1
    further layers are stacked outwards.  Shared edges are emitted once.
    """
    r = F(particle_rad)
    d = F(2.0) * r
    mins = np.asarray(mins, dtype=F)
    maxs = np.asarray(maxs, dtype=F)
    n = np.maximum(np.round((maxs - mins) / d).astype(np.int64), 1) + 1
    pts = set()
    out = []
    for face in faces:
        axis = "xyz".index(face.lower())
        hi = face.isupper()
        u, v = [a for a in range(3) if a != axis]
        for layer in range(layers):
            for a in range(int(n[u])):
                for b in range(int(n[v])):
                    q = [0, 0, 0]
                    q[axis] = (int(n[axis]) - 1 + layer) if hi else -layer
                    q[u], q[v] = a, b
                    q = tuple(q)
                    if q in pts:
                        continue
                    pts.add(q)
                    out.append(q)
    q = np.asarray(out, dtype=F)
    return (mins[None, :] + q * d).astype(F)


def plane_lattice(nx: int, nz: int, y: float, particle_rad: float, x0: float, z0: float, layers: int = 1) -> np.ndarray:
    """nx*nz boundary particles at height y (and layers below), spacing 2r — a vectorised floor."""
    r = F(particle_rad)
    d = F(2.0) * r
    i = np.arange(nx, dtype=F)[:, None, None]
    k = np.arange(nz, dtype=F)[None, :, None]
    l = np.arange(layers, dtype=F)[None, None, :]
    pts = np.empty((nx, nz, layers, 3), dtype=F)
    pts[..., 0] = F(x0) + i * d
    pts[..., 1] = F(y) - l * d
    pts[..., 2] = F(z0) + k * d
    return pts.reshape(-1, 3)


def tank(nx: int, ny: int, nz: int, particle_rad: float, wall_cells: int = 0):
    """A fluid block of nx*ny*nz particles resting in an open-top lattice tank.

    Returns (fluid_positions, boundary_positions).  The tank floor/walls sit one lattice spacing outside
    the fluid block; `wall_cells` extra spacings of head-room are left on +x so a dam-break has somewhere to go.
    """
    r = F(particle_rad)
    d = F(2.0) * r
    fluid = cube_fluid_positions(nx, ny, nz, particle_rad)
    fmin = fluid.min(axis=0)
    fmax = fluid.max(axis=0)
    mins = fmin - d
    maxs = fmax + d
    maxs[0] += F(wall_cells) * d
    maxs[1] += F(max(ny // 2, 4)) * d
    shell = box_shell(mins, maxs, particle_rad, faces="xXyzZ")
    return fluid, shell


def cuboid_surface_ray_sample(half_extents, particle_rad: float) -> np.ndarray:
    """`shape_surface_ray_sample(&Cuboid::new(half_extents), particle_rad)` of /root/reference/src/sampling/ray_sampling.rs
    (:9-15, surface_ray_sample :27-88, quantize_point :209-231, unquantize_points :187-207) for an axis-aligned cuboid, with
    parry's ray cast (out of scope here) replaced by its closed form for a box: a ray along axis i on a lattice line that
    crosses the box enters at -he_i and leaves at +he_i.

    The sampler's lattice has spacing s = 2r and origin (aabb.mins - s) + s/2 per axis; an entry impact is quantised with
    ceil, an exit impact with floor, the other two coordinates with round — so the samples are the outer shell of the
    lattice origin + idx * s, idx = 1..N_i, i.e. they sit half a spacing INSIDE the faces.  Returned in lexicographic index
    order (the reference's order is a HashSet's: unspecified).  All arithmetic in f32 as in the reference.
    """
    he = np.asarray(half_extents, dtype=F)
    s = F(particle_rad) * F(2.0)
    mins = (-he) - s                       # Aabb::loosened(subdivision_size)
    maxs = he + s
    origin = (mins + s / F(2.0)).astype(F)
    pts = set()
    # lattice lines of each axis pair: curr starts at origin and advances by s while curr < maxs (:62-76)
    coords = []
    for a in range(3):
        c, line = origin[a], []
        while c < maxs[a]:
            line.append(c)
            c = F(c + s)
        coords.append(line)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        for cj in coords[j]:
            for ck in coords[k]:
                if not (abs(cj) <= he[j] and abs(ck) <= he[k]):
                    continue  # the ray misses the box
                # round() of the two transverse coordinates of the impact (:222-224)
                qj = int(np.round(F(F(cj - origin[j]) / s)))
                qk = int(np.round(F(F(ck - origin[k]) / s)))
                q_in = int(np.ceil(F(F(-he[i] - origin[i]) / s)))    # entry point: ceil (:218-219)
                q_out = int(np.floor(F(F(he[i] - origin[i]) / s)))   # exit point: floor (:220-221)
                for qi in (q_in, q_out):
                    q = [0, 0, 0]
                    q[i], q[j], q[k] = qi, qj, qk
                    pts.add(tuple(q))
    q = np.asarray(sorted(pts), dtype=np.float64)
    return (origin[None, :] + (q.astype(F) * s)).astype(F)   # unquantize_points (:198-204)


def quat_from_scaled_axis(v) -> np.ndarray:
    """nalgebra `UnitQuaternion::from_scaled_axis` in f32, as (i, j, k, w): the rotation part of `Isometry3::new(t, v)`."""
    v = np.asarray(v, dtype=F)
    ang = F(np.sqrt(F(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
    if ang == 0:
        return np.array([0, 0, 0, 1], dtype=F)
    half = F(ang * F(0.5))
    sn, cs = F(np.sin(half)), F(np.cos(half))
    return np.array([v[0] / ang * sn, v[1] / ang * sn, v[2] / ang * sn, cs], dtype=F)


def basic3(nparticles: int = 15, particle_rad: float = 0.05):
    """The literal scene of /root/reference/examples3d/basic3.rs:16-99: a cube of nparticles^3 fluid particles (helper.rs:4-20)
    lifted by ground_thickness + nparticles * r, and five fixed cuboid colliders — four walls and the ground — each sampled
    by `shape_surface_ray_sample` in its local frame and coupled by `ColliderSampling::StaticSampling`.

    Returns (fluid_positions, [(local_samples, translation, rotation_quaternion_ijkw), ...]) in registration order
    (walls first, ground last, basic3.rs:63-96).  BASELINE config[0]; forces: ArtificialViscosity(1.0, 0.0), DFSPH,
    gravity (0, -9.81, 0), dt = 1/200 (basic3.rs:22,36-44,118).
    """
    r = F(particle_rad)
    ground_thickness, ground_half_width, ground_half_height = F(0.2), F(2.5), F(0.7)
    fluid = cube_fluid_positions(nparticles, nparticles, nparticles, particle_rad)
    fluid = (fluid + np.array([0.0, ground_thickness + F(nparticles) * r, 0.0], dtype=F)[None, :]).astype(F)  # transform_by
    wall = cuboid_surface_ray_sample([ground_thickness, ground_half_height, ground_half_width], particle_rad)
    ground = cuboid_surface_ray_sample([ground_half_width, ground_thickness, ground_half_width], particle_rad)
    half_pi_y = quat_from_scaled_axis([0.0, F(np.pi) / F(2.0), 0.0])
    ident = np.array([0, 0, 0, 1], dtype=F)
    colliders = [
        (wall, np.array([0.0, ground_half_height, ground_half_width], dtype=F), half_pi_y),
        (wall, np.array([0.0, ground_half_height, -ground_half_width], dtype=F), half_pi_y),
        (wall, np.array([ground_half_width, ground_half_height, 0.0], dtype=F), ident),
        (wall, np.array([-ground_half_width, ground_half_height, 0.0], dtype=F), ident),
        (ground, np.zeros(3, dtype=F), ident),
    ]
    return fluid, colliders


def surface_tension3():
    """The literal scene of /root/reference/examples3d/surface_tension3.rs:18-92: a 1 cm^3 droplet in decimetre units —
    cube_fluid(7, 7, 7, r = 0.005, 1000) lifted by 0.08, Akinci2013SurfaceTension(1.0, 0.0) then ArtificialViscosity(0.01, 0.01),
    gravity (0, -0.981, 0), dt = 1/200 — over a fixed cuboid ground (half extents 0.15 x 0.02 x 0.15 at the origin) coupled by
    ColliderSampling::DynamicContactSampling to an initially empty boundary.

    Returns dict(radius, fluid, forces, gravity, ground_half_extents)."""
    r = 0.005
    fluid = cube_fluid_positions(7, 7, 7, r)
    fluid = (fluid + np.array([0.0, 0.08, 0.0], dtype=F)[None, :]).astype(F)  # transform_by(Isometry3::translation(0, 0.08, 0))
    return dict(radius=r, fluid=fluid, forces=[("akinci", 1.0, 0.0), ("artificial", 0.01, 0.01)], gravity=(0.0, -0.981, 0.0),
                ground_half_extents=(0.15, 0.02, 0.15))


def ball_surface_ray_sample(radius: float, particle_rad: float) -> np.ndarray:
    """`shape_surface_ray_sample(&Ball::new(radius), particle_rad)` (/root/reference/src/sampling/ray_sampling.rs:9-88, 187-231)
    with parry's ray cast replaced by its closed form for a ball at the origin: a ray along axis i through (c_j, c_k) with
    c_j^2 + c_k^2 < R^2 enters at -sqrt(R^2 - c_j^2 - c_k^2) and leaves at +sqrt(..).  Same lattice, quantisation (entry: ceil,
    exit: floor, transverse: round) and f32 arithmetic as `cuboid_surface_ray_sample`; lexicographic index order."""
    R = F(radius)
    s = F(particle_rad) * F(2.0)
    mins = np.full(3, -R - s, dtype=F)
    maxs = np.full(3, R + s, dtype=F)
    origin = (mins + s / F(2.0)).astype(F)
    coords = []
    for a in range(3):
        c, line = origin[a], []
        while c < maxs[a]:
            line.append(c)
            c = F(c + s)
        coords.append(line)
    pts = set()
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        for cj in coords[j]:
            for ck in coords[k]:
                d2 = F(R * R) - F(F(cj * cj) + F(ck * ck))
                if not d2 > 0:
                    continue  # the ray misses (or grazes) the ball
                half = F(np.sqrt(d2))
                qj = int(np.round(F(F(cj - origin[j]) / s)))
                qk = int(np.round(F(F(ck - origin[k]) / s)))
                q_in = int(np.ceil(F(F(-half - origin[i]) / s)))
                q_out = int(np.floor(F(F(half - origin[i]) / s)))
                for qi in (q_in, q_out):
                    q = [0, 0, 0]
                    q[i], q[j], q[k] = qi, qj, qk
                    pts.add(tuple(q))
    q = np.asarray(sorted(pts), dtype=np.float64)
    return (origin[None, :] + (q.astype(F) * s)).astype(F)


def faucet3():
    """The literal scene of /root/reference/examples3d/faucet3.rs:19-109: an initially EMPTY fluid (r = 0.0125, rho0 = 1000,
    XSPHViscosity(0.5, 0) then Akinci2013SurfaceTension(1, 10)) fed by a callback that adds a 10 x 10 sheet of particles at
    height 0.6 whenever 0.06 s have passed (:84-103: positions (i d - 10 r, 0.6, j d - 10 r), zero velocity) and deletes what
    fell below y = -2 (:69-73); a fixed ball of radius 0.15 at the origin, ray-sampled and coupled with StaticSampling;
    gravity (0, -9.81, 0), dt = 1/200.

    Returns dict(radius, sheet, ball_samples, period, forces, gravity)."""
    r = 0.025 / 2.0
    d, shift = F(r) * F(2.0), F(-10.0) * F(r)
    sheet = np.array([[F(i) * d + shift, F(0.6), F(j) * d + shift] for i in range(10) for j in range(10)], dtype=F)
    return dict(radius=r, sheet=sheet, ball_samples=ball_surface_ray_sample(0.15, r), period=0.06,
                forces=[("xsph", 0.5, 0.0), ("akinci", 1.0, 10.0)], gravity=(0.0, -9.81, 0.0))
