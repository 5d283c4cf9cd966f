// dcs.hip — ColliderSampling::DynamicContactSampling on the device for ball, cuboid, capsule and cylinder colliders
// (integrations/rapier/fluids_pipeline.rs:193-259).
//
// The reference walks the cells of the fluid grid that the collider's loosened AABB touches, projects every fluid particle
// whose PREDICTED position (x + v dt) lies in that box onto the shape, pushes particles that are inside out of it, and
// emits one boundary particle per projection.  All of that is a per-particle predicate, so here it is one pass over the
// fluid particles (k_dcs_project) followed by a stable compaction of the emitted points (DeviceSelect) and one pass that
// writes them into the boundary's rows (k_dcs_emit).  The grid is not rebuilt after the push-out in the reference — a
// pushed particle stays registered in the cell of its old position for this substep — so the pass runs between the cell-key
// kernel and the sort, and reads each particle's cell back from its key.
//
// parry3d 0.18 (an un-vendored dependency of the reference) supplies the geometry; restated from its published source and
// followed operation by operation:
//   Ball::compute_aabb(pos)   = [t - r, t + r]; Cuboid::compute_aabb(pos) = t -+ |R| half_extents with
//                               R = UnitQuaternion::to_rotation_matrix; Aabb::loosened(m) = [mins - m, maxs + m];
//                               Aabb::contains_local_point: mins <= p <= maxs on every axis
//   project_point_and_get_feature(m, pt) = project_local(m^-1 pt) carried back by m, solid = false:
//     Ball:   inside = |p|^2 <= r^2; proj = p * (r / |p|)
//     Cuboid: shift = sup(mins - p, 0) - sup(p - maxs, 0); outside iff shift != 0 -> p + shift; inside -> the nearest face
//             (the largest of mins - p, p - maxs over the axes)
//     Capsule (new_y: segment a = (0, -hh, 0), b = (0, hh, 0)): s = the segment's closest point (ab.ap <= 0 -> a; >= |ab|^2 -> b;
//             else a + ab (ab.ap / |ab|^2)); (dir, dist) = try_new_and_get(p - s, eps): inside = dist <= r, proj = s + dir r;
//             a point ON the segment: s + (1, 0, 0) r (orthonormal_basis of the axis), inside
//     Cylinder (axis y): planar = |(x, z)|, dir2 = (x, z) / planar ((1, 0) if planar <= eps); inside (|y| <= hh and planar <= r):
//             the nearest of top / bottom / side (strict <, side wins ties); outside: clamp y to the caps and the plane
//             point to the circle
//   Capsule::aabb(pos) = [inf(A, B) - r, sup(A, B) + r] with A, B the posed segment ends; Cylinder::aabb(pos) = t -+ |R| (r, hh, r)
// This file is compiled with -ffp-contract=off (see Makefile): Rust never fuses a*b+c, and the emitted points feed the
// exact d^2 <= h^2 contact test.
#include "dcs.h"
#include "dist.h"
#include "tile.h"
#include <cmath>

namespace salva {

// nalgebra UnitQuaternion * Vector3: t = 2 q.vec x v; v' = t w + q.vec x t + v
__device__ __forceinline__ void quat_rot(float qx, float qy, float qz, float qw, float vx, float vy, float vz, float& ox, float& oy,
                                         float& oz) {
    const float tx = (qy * vz - qz * vy) * 2.0f, ty = (qz * vx - qx * vz) * 2.0f, tz = (qx * vy - qy * vx) * 2.0f;
    const float cx = qy * tz - qz * ty, cy = qz * tx - qx * tz, cz = qx * ty - qy * tx;
    ox = (tx * qw + cx) + vx;
    oy = (ty * qw + cy) + vy;
    oz = (tz * qw + cz) + vz;
}

// the cell the particle was inserted under (hgrid.rs:122-133 filters CELLS by the box, then :211 tests the prediction)
// On a FOLDED grid (device_types.h TileGrid; round 6: worlds with dynamically sampled colliders fold too) the key names the cell modulo
// the axis' period: the image is the one nearest to where the particle is now — an earlier collider of this pass may have pushed it, by
// a fraction of a cell; the periods are at least 64 cells.
__device__ __forceinline__ int dcs_unfold(int rel, uint32_t mask, float x, float h, int origin) {
    if (mask == 0xffffffffu) return origin + rel;
    bool bad = false;
    const int period = (int)(mask + 1u), d = (cell_coord(x, h, bad) - origin) - rel;
    return origin + rel + floor_div(d + period / 2, period) * period;
}
__device__ __forceinline__ bool dcs_in_cells(uint32_t k, const TileGrid& g, const DcsParams& s, const float4& p) {
    const uint32_t tile = k / TCELLS, loc = k % TCELLS;
    const int tz = (int)(tile % (uint32_t)g.ntz), ty = (int)((tile / (uint32_t)g.ntz) % (uint32_t)g.nty),
              tx = (int)(tile / ((uint32_t)g.ntz * (uint32_t)g.nty));
    const int cx = dcs_unfold(tx * TX + (int)(loc / (TY * TZ)), g.mx, p.x, s.h, g.ox), cy = dcs_unfold(ty * TY + (int)((loc / TZ) % TY), g.my, p.y, s.h, g.oy),
              cz = dcs_unfold(tz * TZ + (int)(loc % TZ), g.mz, p.z, s.h, g.oz);
    return !(cx < s.clo[0] || cx > s.chi[0] || cy < s.clo[1] || cy > s.chi[1] || cz < s.clo[2] || cz > s.chi[2]);
}

// :219-243 from the projection on: dpt = particle_pos - proj.point; a particle inside the shape is pushed out along dpt by
// depth + margin and loses its velocity along it; one outside and farther than h + prediction emits nothing (false).
__device__ __forceinline__ bool dcs_finish(uint32_t i, float4 p, float4 v, float px, float py, float pz, float wx, float wy, float wz,
                                           bool inside, const DcsParams& s, float4* __restrict__ posm, float4* __restrict__ vel) {
    const float dx = px - wx, dy = py - wy, dz = pz - wz;
    const float sq = (dx * dx + dy * dy) + dz * dz;
    if (sq > s.eps * s.eps) {  // Unit::try_new_and_get(dpt, f32::EPSILON)
        const float depth = sqrtf(sq);
        const float nx = __fdiv_rn(dx, depth), ny = __fdiv_rn(dy, depth), nz = __fdiv_rn(dz, depth);
        if (inside) {
            const float m = depth + s.margin;
            p.x -= nx * m; p.y -= ny * m; p.z -= nz * m;
            posm[i] = p;
            const float vel_err = (nx * v.x + ny * v.y) + nz * v.z;
            if (vel_err > 0.0f) {
                v.x -= nx * vel_err; v.y -= ny * vel_err; v.z -= nz * vel_err;
                vel[i] = v;
            }
        } else if (depth > s.reach) {
            return false;
        }
    }
    return true;
}

__global__ __launch_bounds__(BLOCK) void k_dcs_project(uint32_t n, float4* __restrict__ posm, float4* __restrict__ vel,
                                                       const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                       const uint32_t* __restrict__ gtag, TileGrid g, DcsParams s,
                                                       float4* __restrict__ cand, uint8_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    flag[i] = 0;
    float4 p = posm[i];
    if (!dcs_in_cells(keys[i], g, s, p)) return;
    float4 v = vel[i];
    const float px = p.x + v.x * s.dt, py = p.y + v.y * s.dt, pz = p.z + v.z * s.dt;  // :206-207
    if (px < s.lo[0] || px > s.hi[0] || py < s.lo[1] || py > s.hi[1] || pz < s.lo[2] || pz > s.hi[2]) return;  // NaN: passes, as `<` / `>` do
    // m^-1 * pt
    float lx, ly, lz;
    quat_rot(-s.q[0], -s.q[1], -s.q[2], s.q[3], px - s.t[0], py - s.t[1], pz - s.t[2], lx, ly, lz);
    float jx, jy, jz;
    bool inside;
    if (s.kind == SALVA_HIP_SHAPE_BALL) {
        const float r = s.p[0], d2 = (lx * lx + ly * ly) + lz * lz;
        inside = d2 <= r * r;
        const float f = __fdiv_rn(r, sqrtf(d2));
        jx = lx * f; jy = ly * f; jz = lz * f;
    } else if (s.kind == SALVA_HIP_SHAPE_CAPSULE) {
        const float hh = s.p[0], r = s.p[1];
        // a = (0, -hh, 0), ab = (0, hh - (-hh), 0), ap = p - a; dots as ((x0 y0 + x1 y1) + x2 y2)
        const float aby = hh - (-hh);
        const float apx = lx, apy = ly - (-hh), apz = lz;
        const float ab_ap = (0.0f * apx + aby * apy) + 0.0f * apz;
        const float sqnab = (0.0f * 0.0f + aby * aby) + 0.0f * 0.0f;
        float sx = 0.0f, sy, sz = 0.0f;
        if (ab_ap <= 0.0f) sy = -hh;
        else if (ab_ap >= sqnab) sy = hh;
        else { const float u = __fdiv_rn(ab_ap, sqnab); sx = 0.0f + 0.0f * u; sy = -hh + aby * u; sz = 0.0f + 0.0f * u; }
        const float ex = lx - sx, ey = ly - sy, ez = lz - sz;
        const float sq = (ex * ex + ey * ey) + ez * ez;
        if (sq > s.eps * s.eps) {
            const float dist = sqrtf(sq);
            inside = dist <= r;
            jx = sx + __fdiv_rn(ex, dist) * r; jy = sy + __fdiv_rn(ey, dist) * r; jz = sz + __fdiv_rn(ez, dist) * r;
        } else {
            inside = true;
            jx = sx + 1.0f * r; jy = sy + 0.0f * r; jz = sz + 0.0f * r;
        }
    } else if (s.kind == SALVA_HIP_SHAPE_CYLINDER) {
        const float hh = s.p[0], r = s.p[1];
        const float planar = sqrtf(lx * lx + lz * lz);
        float dx2 = __fdiv_rn(lx, planar), dz2 = __fdiv_rn(lz, planar);
        if (planar <= s.eps) { dx2 = 1.0f; dz2 = 0.0f; }
        const float qx = dx2 * r, qz = dz2 * r;
        if (ly >= -hh && ly <= hh && planar <= r) {
            inside = true;
            const float top = hh - ly, bottom = ly - (-hh), side = r - planar;
            if (top < bottom && top < side) { jx = lx; jy = hh; jz = lz; }
            else if (bottom < top && bottom < side) { jx = lx; jy = -hh; jz = lz; }
            else { jx = qx; jy = ly; jz = qz; }
        } else {
            inside = false;
            if (ly > hh) { jy = hh; if (planar <= r) { jx = lx; jz = lz; } else { jx = qx; jz = qz; } }
            else if (ly < -hh) { jy = -hh; if (planar <= r) { jx = lx; jz = lz; } else { jx = qx; jz = qz; } }
            else { jx = qx; jy = ly; jz = qz; }
        }
    } else {
        const float l3[3] = {lx, ly, lz};
        float mins_pt[3], pt_maxs[3], shift[3];
        inside = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mins_pt[a] = -s.p[a] - l3[a];
            pt_maxs[a] = l3[a] - s.p[a];
            shift[a] = fmaxf(mins_pt[a], 0.0f) - fmaxf(pt_maxs[a], 0.0f);
            if (shift[a] != 0.0f) inside = false;
        }
        if (inside) {
            float best = -3.402823466e+38f;
            bool is_mins = false;
            int best_id = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (mins_pt[a] < pt_maxs[a]) {
                    if (pt_maxs[a] > best) { best_id = a; is_mins = false; best = pt_maxs[a]; }
                } else if (mins_pt[a] > best) { best_id = a; is_mins = true; best = mins_pt[a]; }
            }
            const float sh = is_mins ? best : -best;
            shift[0] = best_id == 0 ? sh : 0.0f; shift[1] = best_id == 1 ? sh : 0.0f; shift[2] = best_id == 2 ? sh : 0.0f;
        }
        jx = l3[0] + shift[0]; jy = l3[1] + shift[1]; jz = l3[2] + shift[2];
    }
    float wx, wy, wz;
    quat_rot(s.q[0], s.q[1], s.q[2], s.q[3], jx, jy, jz, wx, wy, wz);
    wx += s.t[0]; wy += s.t[1]; wz += s.t[2];
    if (dcs_finish(i, p, v, px, py, pz, wx, wy, wz, inside, s, posm, vel)) {
        // decomposed run (gtag != nullptr): a ghost is pushed like its owner — same inputs, same arithmetic — but only the owner
        // emits; the row then carries the SORTED index (k_dcs_pack turns it into global id + fluid)
        if (gtag && (gtag[i] & GTAG_GHOST)) return;
        cand[i] = make_float4(wx, wy, wz, __uint_as_float(gtag ? i : perm[i]));
        flag[i] = 1;
    }
}

// The host-shape arm (salva_hip_set_boundary_dynamic_sampling_host): the same pass cut in two around the host's
// `project_point_and_get_feature`.  k_dcs_gather: the particles whose cell and predicted position pass the box tests
// (:204-211), as (predicted position, sorted index); k_dcs_apply: the rest of the loop body for those, from the host's
// (projection, is_inside).
__global__ __launch_bounds__(BLOCK) void k_dcs_gather(uint32_t n, const float4* __restrict__ posm, const float4* __restrict__ vel,
                                                      const uint32_t* __restrict__ keys, TileGrid g, DcsParams s,
                                                      float4* __restrict__ cand, uint8_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    flag[i] = 0;
    const float4 p = posm[i];
    if (!dcs_in_cells(keys[i], g, s, p)) return;
    const float4 v = vel[i];
    const float px = p.x + v.x * s.dt, py = p.y + v.y * s.dt, pz = p.z + v.z * s.dt;
    if (px < s.lo[0] || px > s.hi[0] || py < s.lo[1] || py > s.hi[1] || pz < s.lo[2] || pz > s.hi[2]) return;
    cand[i] = make_float4(px, py, pz, __uint_as_float(i));
    flag[i] = 1;
}
__global__ __launch_bounds__(BLOCK) void k_dcs_apply(uint32_t cnt, const float4* __restrict__ pred, const float4* __restrict__ proj,
                                                     float4* __restrict__ posm, float4* __restrict__ vel, const uint32_t* __restrict__ perm,
                                                     const uint32_t* __restrict__ gtag, DcsParams s, float4* __restrict__ cand,
                                                     uint8_t* __restrict__ flag) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= cnt) return;
    const float4 pr = pred[k], w = proj[k];
    const uint32_t i = __float_as_uint(pr.w);
    const float4 p = posm[i], v = vel[i];
    bool keep = dcs_finish(i, p, v, pr.x, pr.y, pr.z, w.x, w.y, w.z, w.w != 0.0f, s, posm, vel);
    if (gtag && (gtag[i] & GTAG_GHOST)) keep = false;  // (decomposed run: pushed here too, emitted by its owner; see k_dcs_project)
    flag[k] = keep ? 1 : 0;
    if (keep) cand[k] = make_float4(w.x, w.y, w.z, __uint_as_float(gtag ? i : perm[i]));
}

// Decomposed run: this rank's compacted rows (point, sorted index of the source particle) -> its section of the table every
// rank assembles (World::dist_gather_emitted): (point, global id of the source) and the source's fluid.
__global__ __launch_bounds__(BLOCK) void k_dcs_pack(uint32_t cnt, const float4* __restrict__ rows, const uint32_t* __restrict__ gid,
                                                    const uint32_t* __restrict__ model, float4* __restrict__ out_rows,
                                                    uint32_t* __restrict__ out_models) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= cnt) return;
    const float4 r = rows[k];
    const uint32_t i = __float_as_uint(r.w);
    out_rows[k] = make_float4(r.x, r.y, r.z, __uint_as_float(gid[i]));
    out_models[k] = model[i];
}

__global__ __launch_bounds__(BLOCK) void k_dcs_emit(uint32_t cnt, const float4* __restrict__ cand, SalvaHipRigidPose pose, uint32_t slot,
                                                    float4* __restrict__ pos, float4* __restrict__ vel, uint32_t* __restrict__ src) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= cnt) return;
    const float4 c = cand[i];
    pos[i] = make_float4(c.x, c.y, c.z, 0.0f);  // boundary.volumes.push(0) :249
    float vx = 0.0f, vy = 0.0f, vz = 0.0f;
    if (pose.has_body) {  // body.velocity_at_point(&proj.point) :241-242 (the WORLD point here, unlike the static arm)
        const float rx = c.x - pose.world_com[0], ry = c.y - pose.world_com[1], rz = c.z - pose.world_com[2];
        vx = pose.linvel[0] + (pose.angvel[1] * rz - pose.angvel[2] * ry);
        vy = pose.linvel[1] + (pose.angvel[2] * rx - pose.angvel[0] * rz);
        vz = pose.linvel[2] + (pose.angvel[0] * ry - pose.angvel[1] * rx);
    }
    vel[i] = make_float4(vx, vy, vz, __uint_as_float(slot));
    src[i] = __float_as_uint(c.w);
}

// half extents of the posed shape's AABB about the pose's translation (parry compute_aabb, see the header of this file)
void shape_world_extent(const SalvaHipShape& shape, const float q[4], float ext[3]) {
    if (shape.kind == SALVA_HIP_SHAPE_BALL) {
        ext[0] = ext[1] = ext[2] = shape.params[0];
    } else if (shape.kind == SALVA_HIP_SHAPE_CAPSULE) {
        // the posed segment ends are t -+ q * (0, hh, 0): extent = |q * b| + radius
        const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
        const float bx = 0.0f, by = shape.params[0], bz = 0.0f;
        const float tx = (qy * bz - qz * by) * 2.0f, ty = (qz * bx - qx * bz) * 2.0f, tz = (qx * by - qy * bx) * 2.0f;
        const float cx = qy * tz - qz * ty, cy = qz * tx - qx * tz, cz = qx * ty - qy * tx;
        const float u[3] = {(tx * qw + cx) + bx, (ty * qw + cy) + by, (tz * qw + cz) + bz};
        for (int a = 0; a < 3; ++a) ext[a] = std::fabs(u[a]) + shape.params[1];
    } else {
        // UnitQuaternion::to_rotation_matrix (nalgebra geometry/quaternion.rs), then |R| * half_extents
        // (cylinder: the half extents of its local box are (radius, half_height, radius))
        const float i = q[0], j = q[1], k = q[2], w = q[3];
        const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
        const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f, ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
        const float m[3][3] = {{ww + ii - jj - kk, ij - wk, wj + ik}, {wk + ij, ww - ii + jj - kk, jk - wi}, {ik - wj, wi + jk, ww - ii - jj + kk}};
        const bool cyl = shape.kind == SALVA_HIP_SHAPE_CYLINDER;
        const float he[3] = {cyl ? shape.params[1] : shape.params[0], cyl ? shape.params[0] : shape.params[1], cyl ? shape.params[1] : shape.params[2]};
        for (int a = 0; a < 3; ++a)
            ext[a] = (std::fabs(m[a][0]) * he[0] + std::fabs(m[a][1]) * he[1]) + std::fabs(m[a][2]) * he[2];
    }
}

// Ball::compute_aabb / Cuboid::compute_aabb of the posed shape, loosened by h + prediction (:196-199), and the cell range
// HGrid::cells_intersecting_aabb walks (hgrid.rs:128-131).
DcsParams dcs_params(const SalvaHipShape& shape, const SalvaHipRigidPose& pose, float h, float particle_radius, float dt) {
    DcsParams s{};
    s.kind = shape.kind;
    for (int a = 0; a < 3; ++a) { s.p[a] = shape.params[a]; s.t[a] = pose.translation[a]; }
    for (int a = 0; a < 4; ++a) s.q[a] = pose.rotation[a];
    const float prediction = h * 0.5f;
    s.margin = particle_radius * 0.1f;
    s.reach = h + prediction;
    s.h = h;
    s.dt = dt;
    s.eps = 1.1920929e-7f;
    float ext[3];
    shape_world_extent(shape, pose.rotation, ext);
    for (int a = 0; a < 3; ++a) {
        s.lo[a] = (pose.translation[a] - ext[a]) - s.reach;
        s.hi[a] = (pose.translation[a] + ext[a]) + s.reach;
        const float fl = std::floor(s.lo[a] / h), fh = std::floor(s.hi[a] / h);
        s.clo[a] = (int)std::fmin(std::fmax(fl, -1073741824.0f), 1073741824.0f);
        s.chi[a] = (int)std::fmin(std::fmax(fh, -1073741824.0f), 1073741824.0f);
    }
    return s;
}

// the host-shape arm: the box is the host's `collider.shape().compute_aabb(&collider_pos)`
DcsParams dcs_params_host(const float mins[3], const float maxs[3], float h, float particle_radius, float dt) {
    DcsParams s{};
    s.kind = SALVA_HIP_SHAPE_HOST;
    s.q[3] = 1.0f;
    const float prediction = h * 0.5f;
    s.margin = particle_radius * 0.1f;
    s.reach = h + prediction;
    s.h = h;
    s.dt = dt;
    s.eps = 1.1920929e-7f;
    for (int a = 0; a < 3; ++a) {
        s.lo[a] = mins[a] - s.reach;
        s.hi[a] = maxs[a] + s.reach;
        const float fl = std::floor(s.lo[a] / h), fh = std::floor(s.hi[a] / h);
        s.clo[a] = (int)std::fmin(std::fmax(fl, -1073741824.0f), 1073741824.0f);
        s.chi[a] = (int)std::fmin(std::fmax(fh, -1073741824.0f), 1073741824.0f);
    }
    return s;
}

void launch_dcs_project(uint32_t n, float4* posm, float4* vel, const uint32_t* keys, const uint32_t* perm, const uint32_t* gtag, TileGrid g,
                        const DcsParams& s, float4* cand, uint8_t* flag, hipStream_t st) {
    if (n == 0) return;
    k_dcs_project<<<div_up(n, BLOCK), BLOCK, 0, st>>>(n, posm, vel, keys, perm, gtag, g, s, cand, flag);
    SALVA_HIP_CHECK(hipGetLastError());
}
void launch_dcs_gather(uint32_t n, const float4* posm, const float4* vel, const uint32_t* keys, TileGrid g, const DcsParams& s, float4* cand,
                       uint8_t* flag, hipStream_t st) {
    if (n == 0) return;
    k_dcs_gather<<<div_up(n, BLOCK), BLOCK, 0, st>>>(n, posm, vel, keys, g, s, cand, flag);
    SALVA_HIP_CHECK(hipGetLastError());
}
void launch_dcs_apply(uint32_t cnt, const float4* pred, const float4* proj, float4* posm, float4* vel, const uint32_t* perm,
                      const uint32_t* gtag, const DcsParams& s, float4* cand, uint8_t* flag, hipStream_t st) {
    if (cnt == 0) return;
    k_dcs_apply<<<div_up(cnt, BLOCK), BLOCK, 0, st>>>(cnt, pred, proj, posm, vel, perm, gtag, s, cand, flag);
    SALVA_HIP_CHECK(hipGetLastError());
}
void launch_dcs_pack(uint32_t cnt, const float4* rows, const uint32_t* gid, const uint32_t* model, float4* out_rows, uint32_t* out_models,
                     hipStream_t st) {
    if (cnt == 0) return;
    k_dcs_pack<<<div_up(cnt, BLOCK), BLOCK, 0, st>>>(cnt, rows, gid, model, out_rows, out_models);
    SALVA_HIP_CHECK(hipGetLastError());
}
void launch_dcs_emit(uint32_t cnt, const float4* cand, const SalvaHipRigidPose& pose, uint32_t slot, float4* pos, float4* vel,
                     uint32_t* src, hipStream_t st) {
    if (cnt == 0) return;
    k_dcs_emit<<<div_up(cnt, BLOCK), BLOCK, 0, st>>>(cnt, cand, pose, slot, pos, vel, src);
    SALVA_HIP_CHECK(hipGetLastError());
}

}  // namespace salva
