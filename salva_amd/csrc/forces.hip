// forces.hip — the built-in NonPressureForce implementations as tile kernels (tile.h).
//
// Behaviour specified by /root/reference/src/solver/viscosity/xsph_viscosity.rs:31-95,
// src/solver/viscosity/artificial_viscosity.rs:41-124 and
// src/solver/surface_tension/akinci2013_surface_tension.rs:43-192.  Each force belongs to one Fluid
// (object/fluid.rs:14) and acts only between particles of that fluid (`c.i_model == c.j_model`) and between the
// fluid and the boundaries; `model` selects the fluid, particles of other fluids are skipped.
// At the time `predict_advection` runs, `fluid.velocities` equals w = v + dv of the divergence solve
// (DFSPH, dfsph_solver.rs:688-693) or v itself (IISPH, dv = 0), so neighbour velocities are read from w,
// whose .w component carries the neighbour's model id.
#include "kernels.h"
#include "tile.h"
#include "pairs.h"

namespace SALVA_KNS {
using namespace salva;

// ------------------------------------------------------------------------------------------------ XSPH
// a_i += inv_dt * [ sum_j (v_j - v_i) c_f W_ij m_j / rho_j  +  sum_b (v_b - v_i) c_b W_ib V_b rho0 / rho_i ]
// inv_dt is the *previous* substep's (timestep.advance happens after predict_advection, dfsph_solver.rs:693-702).
// Round 3: fixed P | W layout (pairs.h): P = posmr = (x_j, m_j / rho_j) written by k_density_alpha, W = (v_j + dv_j, model id);
// the weight comes from kernel_wg2 without a branch (W itself has no near-zero special case, so every slice takes this path).
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_xsph(StepCtx c, uint32_t model, float fc, float bc, float inv_dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    // own record and list head: in registers before the staging barrier
    struct Own { float4 pi, vi; float ri; uint32_t cnt; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) { return Own{c.posm[i], c.w[i], c.rho[i], c.nff[i], list_regs(c, gs)}; };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pw_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pw(c, static_cast<const float4*>(c.posmr), static_cast<const float4*>(c.w), dist, Bp, Bv, true);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);  // (wave-uniform: before any lane drops out)
        const float4 pi = o.pi;
        const float4 vi = o.vi;
        if (!active || __float_as_uint(vi.w) != model) return;
        const float rho0 = c.rho0_tab[model];
        float fx = 0.f, fy = 0.f, fz = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
#ifdef SALVA_OTHER_KERNELS  // (kernel_wg2 is the cubic spline: another KernelDensity evaluates W the general way)
        if (fc != 0.0f && c.sc.kd != 0) {
            for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const RecPW A = load_pw(s << 4, dist);
                const float dx = pi.x - A.p.x, dy = pi.y - A.p.y, dz = pi.z - A.p.z;
                const float wgt = kernel_weight(dx * dx + dy * dy + dz * dz, c.sc);
                const float sc = (__float_as_uint(A.w.w) == model) ? fc * wgt * A.p.w : 0.0f;
                fx += (A.w.x - vi.x) * sc; fy += (A.w.y - vi.y) * sc; fz += (A.w.z - vi.z) * sc;
            });
        } else
#endif
        if (fc != 0.0f) {
            // wave-uniform trip count over the padded list: a padding entry is the particle itself, v_j - v_i = 0 exactly
            f2 ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f};
            const f2 tiny = {1.0e-30f, 1.0e-30f};
            for_each_ff2<true, false, true>(c, gs, nqu, o.lh, [&](uint32_t off) { return load_pw(off, dist); },
                                            [&](const RecPW& A, const RecPW& B) { SALVA_PAIR_MATH
                const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                f2 r2 = dz * dz + tiny;
                r2 = dy * dy + r2;
                r2 = dx * dx + r2;
                const KernelWG2 k = kernel_wg2(r2, c.sc);
                f2 sc = k.w * f2{A.p.w, B.p.w};  // W_ij m_j / rho_j (the common factors once per particle, below)
                sc.x = (__float_as_uint(A.w.w) == model) ? sc.x : 0.0f;
                sc.y = (__float_as_uint(B.w.w) == model) ? sc.y : 0.0f;
                ax += f2{A.w.x - vi.x, B.w.x - vi.x} * sc;
                ay += f2{A.w.y - vi.y, B.w.y - vi.y} * sc;
                az += f2{A.w.z - vi.z, B.w.z - vi.z} * sc;
            });
            const float f = fc * c.sc.wscale;
            fx = (ax.x + ax.y) * f; fy = (ay.x + ay.y) * f; fz = (az.x + az.y) * f;
        }
        if (bc != 0.0f) {
            const float ri = o.ri;
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float4 vj = Bv[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float wgt = kernel_weight(dx * dx + dy * dy + dz * dz, c.sc);
                const float sc = fast_div(bc * wgt * pj.w * rho0, ri);
                const float ex = (vj.x - vi.x) * sc, ey = (vj.y - vi.y) * sc, ez = (vj.z - vi.z) * sc;
                bx += ex; by += ey; bz += ez;
                if (c.bforce && !is_ghost(c, i)) {
                    const float fs = -pi.w * inv_dt;  // delta * (-mi * inv_dt) :88-89
                    apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(vj.w), ex * fs, ey * fs, ez * fs);
                }
            });
        }
        float4 a = c.acc[i];
        a.x += fx * inv_dt + bx * inv_dt;
        a.y += fy * inv_dt + by * inv_dt;
        a.z += fz * inv_dt + bz * inv_dt;
        c.acc[i] = a;
    });
}
void launch_xsph(const StepCtx& c, const TileLds& L, uint32_t model, float fluid_coeff, float boundary_coeff,
                 float inv_dt_prev, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_xsph, c, L, model, fluid_coeff, boundary_coeff, inv_dt_prev, s);
    const uint32_t ds = pick_ds(pw_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_xsph, ds, c, L, pw_bytes(L, ds, false), s, c, model, fluid_coeff, boundary_coeff, inv_dt_prev);
}

// ------------------------------------------------------------------------------------------------ Monaghan artificial viscosity
// approaching pairs only (r.v < 0): mu = h r.v / (r^2 + 0.01 h^2);
// a_i += grad W_ij c_f (c_s alpha mu - beta mu^2) m_j / ((rho_i + rho_j)/2)
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_artificial_viscosity(StepCtx c, uint32_t model, float fc, float bc,
                                                                      float alpha, float beta, float cs) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float4* Lw = nullptr;
    const float* Lr = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.w), static_cast<const float*>(c.rho), Lp, Lw, Lr);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_boundary(c, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        const float4 vi = c.w[i];
        const float ri = c.rho[i];
        const float rho0 = c.rho0_tab[model];
        const float h = c.sc.h;
        const float eta2 = h * h * 0.01f;
        float fx = 0.f, fy = 0.f, fz = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
        if (fc != 0.0f) {
            for_each_ff_regs(c, gs, lo, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Lp[s];
                const float4 vj = lds_f4(Lw + s);
                const float rj = Lr[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                const float vr = dx * (vi.x - vj.x) + dy * (vi.y - vj.y) + dz * (vi.z - vj.z);
                if (__float_as_uint(vj.w) == model && vr < 0.0f) {
                    const float g = kernel_grad(r2, c.sc);
                    const float mu = fast_div(h * vr, r2 + eta2);
                    const float sc = g * (fc * (cs * alpha * mu - beta * mu * mu) * fast_div(pj.w, (ri + rj) * 0.5f));
                    fx += dx * sc; fy += dy * sc; fz += dz * sc;
                }
            });
        }
        if (bc != 0.0f) {
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float4 vj = Bv[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                const float vr = dx * (vi.x - vj.x) + dy * (vi.y - vj.y) + dz * (vi.z - vj.z);
                if (vr < 0.0f) {
                    const float g = kernel_grad(r2, c.sc);
                    const float mu = fast_div(h * vr, r2 + eta2);
                    const float sc = g * (bc * (cs * alpha * mu - beta * mu * mu) * fast_div(pj.w * rho0, ri));
                    bx += dx * sc; by += dy * sc; bz += dz * sc;
                    // the reference applies the *running sum* of the boundary acceleration here (:117)
                    if (c.bforce && !is_ghost(c, i))
                        apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(vj.w), bx * -pi.w, by * -pi.w, bz * -pi.w);
                }
            });
        }
        float4 a = c.acc[i];
        a.x += fx + bx; a.y += fy + by; a.z += fz + bz;
        c.acc[i] = a;
    });
}
void launch_artificial_viscosity(const StepCtx& c, const TileLds& L, uint32_t model, float fluid_coeff,
                                 float boundary_coeff, float alpha, float beta, float speed_of_sound, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_artificial_viscosity, c, L, model, fluid_coeff, boundary_coeff, alpha, beta, speed_of_sound, s);
    SALVA_LAUNCH_TILE(k_artificial_viscosity, c, L, L.bytes(36, 32, 5), s, c, model, fluid_coeff, boundary_coeff, alpha, beta,
                      speed_of_sound);
}

// ------------------------------------------------------------------------------------------------ Akinci 2013
// pass 1 (compute_normals :43-68): n_i = h sum_{j same fluid} (m_j / rho_j) grad W_ij
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_akinci_normals(StepCtx c, uint32_t model) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float* Lr = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.rho), Lp, Lr);
    const uint32_t* Lm = nullptr;
    if (c.nmodels > 1) t.stage(c, static_cast<const uint32_t*>(c.model), Lm);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        float nx = 0.f, ny = 0.f, nz = 0.f;
        for_each_ff_regs(c, gs, lo, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Lp[s];
            const float rj = Lr[s];
            const bool same = Lm ? (Lm[s] == model) : true;
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            const float sc = same ? g * fast_div(pj.w, rj) : 0.0f;
            nx += dx * sc; ny += dy * sc; nz += dz * sc;
        });
        // .w carries rho_i: the force pass then stages (position, mass) + (normal, density) = 32 bytes per halo slot and two
        // tiles share a CU (36-40 bytes made it one: 127 us per launch at 10^6 particles)
        c.normal[i] = make_float4(nx * c.sc.h, ny * c.sc.h, nz * c.sc.h, c.rho[i]);
    });
}
// The same pass for a world with ONE fluid on the packed pair loop of the solver kernels (pairs.h): n_i = h sum_j grad W_ij (m_j / rho_j)
// is pair_sum_gradient over P = posmr = (x_j, m_j / rho_j) — the record k_density_alpha writes — with k_ij = 1; two contacts per step,
// kernel_gfac2, no branch; padded lists (a self contact adds no gradient); slices that hold a pair closer than 1e-5 h walk their
// exact lists with kernel_grad.  One 16-byte array: four or five tiles per CU.
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_akinci_normals_one_fluid(StepCtx c) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi; float rho; uint32_t cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) { return Own{c.posm[i], c.rho[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)}; };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const float4* Lp = nullptr;
    t.stage(c, static_cast<const float4*>(c.posmr), Lp);  // first carve: LDS byte 0 (lds_ld16)
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi;
        float nx, ny, nz;
        if (near) {
            nx = ny = nz = 0.0f;
            for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = lds_ld16(s << 4);
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float sc = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * pj.w;
                nx += dx * sc; ny += dy * sc; nz += dz * sc;
            });
        } else {
            f2 ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f};
            const f2 tiny = {1.0e-30f, 1.0e-30f};
            for_each_ff2<true, false, 1>(c, gs, nqu, o.lh, [&](uint32_t off) { return lds_ld16(off); }, [&](const float4& A, const float4& B) { SALVA_PAIR_MATH
                const f2 dx = {pi.x - A.x, pi.x - B.x}, dy = {pi.y - A.y, pi.y - B.y}, dz = {pi.z - A.z, pi.z - B.z};
                f2 r2 = dz * dz + tiny;
                r2 = dy * dy + r2;
                r2 = dx * dx + r2;
                const f2 coeff = kernel_gfac2(r2, c.sc) * f2{A.w, B.w};
                ax += dx * coeff; ay += dy * coeff; az += dz * coeff;
            });
            nx = (ax.x + ax.y) * c.sc.gscale; ny = (ay.x + ay.y) * c.sc.gscale; nz = (az.x + az.y) * c.sc.gscale;
        }
        c.normal[i] = make_float4(nx * c.sc.h, ny * c.sc.h, nz * c.sc.h, o.rho);  // (.w carries rho_i: see k_akinci_normals)
    });
}
// the packed loops do not reproduce `|d|^2 <= eps^2 -> zero` per contact: they rely on the slices k_density_alpha flags for pairs
// closer than 1e-5 h, which covers eps only while 1e-5 h >= eps (h >= 0.012)
static inline bool akinci_fast_ok(const StepCtx& c) { return c.nmodels == 1 && c.sc.tiny_r2 >= c.sc.eps2 && (c.sc.kd | c.sc.kg) == 0; }
void launch_akinci_normals(const StepCtx& c, const TileLds& L, uint32_t model, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_akinci_normals, c, L, model, s);
    if (akinci_fast_ok(c)) {
        SALVA_LAUNCH_TILE(k_akinci_normals_one_fluid, c, L, L.bytes(16, 0, 1), s, c);
        return;
    }
    SALVA_LAUNCH_TILE(k_akinci_normals, c, L, L.bytes(24, 0, 3), s, c, model);
}

// cohesion_kernel :71-88 — C(r) = 32/(pi h^9) * { 2 (h-r)^3 r^3 - h^6/64 | r <= h/2 ; (h-r)^3 r^3 | r <= h ; 0 }
__device__ __forceinline__ float cohesion_kernel(float r, float h, float norm, float h6_64) {
    SALVA_PAIR_MATH
    const float a = (h - r) * (h - r) * (h - r) * (r * r * r);
    float v = (r <= h * 0.5f) ? 2.0f * a - h6_64 : a;
    v = (r <= h) ? v : 0.0f;
    return norm * v;
}
// adhesion_kernel :90-111 — A(r) = 0.007 / h^3.25 * (-4 r^2/h + 6 r - 2 h)^(1/4) for h/2 < r <= h
__device__ __forceinline__ float adhesion_kernel(float r, float h, float norm) {
    SALVA_PAIR_MATH
    if (r > h * 0.5f && r <= h) {
        const float tt = fmaxf(fast_div(-4.0f * r * r, h) + 6.0f * r - 2.0f * h, 0.0f);
        return norm * __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(tt));  // (1 ulp each: the sum it feeds is compared at 1e-5)
    }
    return 0.0f;
}

// pass 2 (solve :114-192): cohesion + curvature between same-fluid particles, adhesion with boundaries.
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_akinci_forces(StepCtx c, uint32_t model, float tc, float ac, float cnorm,
                                                               float h6_64, float anorm) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float4* Ln = nullptr;  // (normal, density) — only meaningful for particles of `model`, the only ones used
    t.stage(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.normal), Lp, Ln);
    const uint32_t* Lm = nullptr;
    if (c.nmodels > 1) t.stage(c, static_cast<const uint32_t*>(c.model), Lm);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_boundary(c, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        const float ri = c.rho[i];
        const float rho0 = c.rho0_tab[model];
        const float h = c.sc.h;
        float4 a = c.acc[i];
        if (tc != 0.0f) {
            const float4 ni = c.normal[i];
            for_each_ff_regs(c, gs, lo, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Lp[s];
                const float4 nj = Ln[s];
                const float rj = nj.w;
                const bool same = Lm ? (Lm[s] == model) : true;
                if (same) {
                    const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    float cs = 0.0f;  // cohesion_vec = dir * C(dist) ; cohesion_acc = cohesion_vec * (-tc * m_j)
                    if (r2 > c.sc.eps2) {
                        const float rinv = __builtin_amdgcn_rsqf(r2);
                        cs = cohesion_kernel(r2 * rinv, h, cnorm, h6_64) * rinv * (-tc * pj.w);
                    }
                    const float kij = fast_div(2.0f * rho0, ri + rj);
                    a.x += ((ni.x - nj.x) * -tc + dx * cs) * kij;
                    a.y += ((ni.y - nj.y) * -tc + dy * cs) * kij;
                    a.z += ((ni.z - nj.z) * -tc + dz * cs) * kij;
                }
            });
        }
        if (ac != 0.0f) {
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                float sc = 0.0f;
                if (r2 > c.sc.eps2) {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    sc = adhesion_kernel(r2 * rinv, h, anorm) * rinv * (ac * pj.w * rho0);
                }
                const float ex = dx * sc, ey = dy * sc, ez = dz * sc;
                a.x -= ex; a.y -= ey; a.z -= ez;
                if (c.bforce && !is_ghost(c, i))
                    apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(Bv[s].w), ex * pi.w, ey * pi.w, ez * pi.w);
            });
        }
        c.acc[i] = a;
    });
}
// One fluid: the cohesion + curvature sum on the fixed P | N layout (pairs.h load_pw: position + mass at LDS byte 0, normal + density
// at a compile-time distance), two contacts per step, no branch:
//   C(r) / cnorm = (r <= h / 2) ? 2 (h - r)^3 r^3 - h^6 / 64 : (h - r)^3 r^3      (list contacts have r <= h)
// as one select between two values both contacts compute anyway; the division of k_ij = 2 rho0 / (rho_i + rho_j) is v_rcp_f32;
// r^2 = 0 (the self contact and the padding) is kept finite by the 1e-30 that rides in the first FMA and contributes dx * (...) = 0.
// 45 VALU per pair of contacts against ~90 for the scalar walk with its two branches per contact.
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_akinci_forces_one_fluid(StepCtx c, float tc, float ac, float cnorm, float h6_64, float anorm) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi, ni, a; uint32_t cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) { return Own{c.posm[i], c.normal[i], c.acc[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)}; };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pw_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pw(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.normal), dist, Bp, Bv, true);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi, ni = o.ni;
        const float ri = ni.w;  // (k_akinci_normals_one_fluid stored rho_i there)
        const float rho0 = c.rho0_single;
        const float h = c.sc.h;
        float4 a = o.a;
        if (tc != 0.0f) {
            if (near) {
                for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                    const RecPW A = load_pw(s << 4, dist);
                    const float dx = pi.x - A.p.x, dy = pi.y - A.p.y, dz = pi.z - A.p.z;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    float cs = 0.0f;
                    if (r2 > c.sc.eps2) {
                        const float rinv = __builtin_amdgcn_rsqf(r2);
                        cs = cohesion_kernel(r2 * rinv, h, cnorm, h6_64) * rinv * (-tc * A.p.w);
                    }
                    const float kij = fast_div(2.0f * rho0, ri + A.w.w);
                    a.x += ((ni.x - A.w.x) * -tc + dx * cs) * kij;
                    a.y += ((ni.y - A.w.y) * -tc + dy * cs) * kij;
                    a.z += ((ni.z - A.w.z) * -tc + dz * cs) * kij;
                });
            } else {
                f2 ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f};
                const f2 tiny = {1.0e-30f, 1.0e-30f};
                const float hh = 0.5f * h, two_rho0 = 2.0f * rho0, mtc = -tc, ctc = -tc * cnorm;
                for_each_ff2<true, false, 1>(c, gs, nqu, o.lh, [&](uint32_t off) { return load_pw(off, dist); },
                                             [&](const RecPW& A, const RecPW& B) { SALVA_PAIR_MATH
                    const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                    f2 r2 = dz * dz + tiny;
                    r2 = dy * dy + r2;
                    r2 = dx * dx + r2;
                    f2 rinv;
                    rinv.x = __builtin_amdgcn_rsqf(r2.x); rinv.y = __builtin_amdgcn_rsqf(r2.y);
                    const f2 r = r2 * rinv;
                    const f2 hm = h - r;
                    const f2 q = (hm * hm * hm) * (r * r * r);
                    const f2 q2 = q * 2.0f - h6_64;
                    f2 v;
                    v.x = (r.x <= hh) ? q2.x : q.x; v.y = (r.y <= hh) ? q2.y : q.y;
                    const f2 cs = v * rinv * (f2{A.p.w, B.p.w} * ctc);
                    f2 kij;
                    kij.x = __builtin_amdgcn_rcpf(ri + A.w.w); kij.y = __builtin_amdgcn_rcpf(ri + B.w.w);
                    kij = kij * two_rho0;
                    const f2 nx = {ni.x - A.w.x, ni.x - B.w.x}, ny = {ni.y - A.w.y, ni.y - B.w.y}, nz = {ni.z - A.w.z, ni.z - B.w.z};
                    ax += (nx * mtc + dx * cs) * kij;
                    ay += (ny * mtc + dy * cs) * kij;
                    az += (nz * mtc + dz * cs) * kij;
                });
                a.x += ax.x + ax.y; a.y += ay.x + ay.y; a.z += az.x + az.y;
            }
        }
        if (ac != 0.0f) {
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                float sc = 0.0f;
                if (r2 > c.sc.eps2) {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    sc = adhesion_kernel(r2 * rinv, h, anorm) * rinv * (ac * pj.w * rho0);
                }
                const float ex = dx * sc, ey = dy * sc, ez = dz * sc;
                a.x -= ex; a.y -= ey; a.z -= ez;
                if (c.bforce && !is_ghost(c, i))
                    apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(Bv[s].w), ex * pi.w, ey * pi.w, ez * pi.w);
            });
        }
        c.acc[i] = a;
    });
}
void launch_akinci_forces(const StepCtx& c, const TileLds& L, uint32_t model, float tension, float adhesion, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_akinci_forces, c, L, model, tension, adhesion, s);
    const double h = c.sc.h;
    // normalisers evaluated in f64 on the host then rounded once
    const float cnorm = (float)(32.0 / (3.14159265358979323846 * pow(h, 9)));
    const float h6_64 = (float)(pow(h, 6) / 64.0);
    const float anorm = (float)(0.007 / pow(h, 3.25));
    if (akinci_fast_ok(c)) {
        const uint32_t ds = pick_ds(pw_slots(L), L.ds_level);
        SALVA_LAUNCH_FIXED(k_akinci_forces_one_fluid, ds, c, L, pw_bytes(L, ds, false), s, c, tension, adhesion, cnorm, h6_64, anorm);
        return;
    }
    SALVA_LAUNCH_TILE(k_akinci_forces, c, L, L.bytes(36, 32, 5), s, c, model, tension, adhesion, cnorm, h6_64, anorm);
}

// ------------------------------------------------------------------------------------------------ He et al. 2014
// surface_tension/he2014_surface_tension.rs.  Three dependent neighbour passes over the same fluid:
// pass 1 (compute_colors :40-76): c_i = sum_{j same fluid} W_ij m_j / rho_j + sum_b W_ib V_b
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_he2014_colors(StepCtx c, uint32_t model, float* __restrict__ colors) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float* Lr = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.rho), Lp, Lr);
    const uint32_t* Lm = nullptr;
    if (c.nmodels > 1) t.stage(c, static_cast<const uint32_t*>(c.model), Lm);
    const float4* Bp = nullptr;
    t.stage_boundary(c, Bp);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        float color = 0.0f;
        for_each_ff_regs(c, gs, lo, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Lp[s];
            const float rj = Lr[s];
            const bool same = Lm ? (Lm[s] == model) : true;
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float wgt = kernel_weight(dx * dx + dy * dy + dz * dz, c.sc);
            color += same ? fast_div(wgt * pj.w, rj) : 0.0f;
        });
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Bp[s];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            color += kernel_weight(dx * dx + dy * dy + dz * dz, c.sc) * pj.w;
        });
        colors[i] = color;
    });
}
// pass 2 (compute_gradc :78-106): g_i = | (sum_{j same fluid} grad W_ij c_j m_j / rho_j) / c_i |^2
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_he2014_gradc(StepCtx c, uint32_t model, const float* __restrict__ colors,
                                                              float* __restrict__ gradcs) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float* Lr = nullptr;
    const float* Lc = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.rho), colors, Lp, Lr, Lc);
    const uint32_t* Lm = nullptr;
    if (c.nmodels > 1) t.stage(c, static_cast<const uint32_t*>(c.model), Lm);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for_each_ff_regs(c, gs, lo, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Lp[s];
            const float rj = Lr[s], cj = Lc[s];
            const bool same = Lm ? (Lm[s] == model) : true;
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            const float sc = same ? fast_div(g * cj * pj.w, rj) : 0.0f;
            gx += dx * sc; gy += dy * sc; gz += dz * sc;
        });
        const float ci = colors[i];
        gx /= ci; gy /= ci; gz /= ci;
        gradcs[i] = gx * gx + gy * gy + gz * gz;
    });
}
// pass 3 (solve :132-178): a_i += t_f/(2 m_i) sum_j grad W_ij (m_i/rho_i)(m_j/rho_j)(g_i+g_j)/2
//                                 + sum_b grad W_ib (1/rho_i) V_b rho0 ... (g_i t_b / 4), reaction -f on the boundary
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_he2014_forces(StepCtx c, uint32_t model, float tc, float bc,
                                                               const float* __restrict__ gradcs) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float* Lr = nullptr;
    const float* Lg = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.rho), gradcs, Lp, Lr, Lg);
    const uint32_t* Lm = nullptr;
    if (c.nmodels > 1) t.stage(c, static_cast<const uint32_t*>(c.model), Lm);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_boundary(c, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        const float ri = c.rho[i], gi = gradcs[i];
        const float rho0 = c.rho0_tab[model];
        const float mi = pi.w;
        float4 a = c.acc[i];
        if (tc != 0.0f) {
            const float ts = tc / (2.0f * mi);
            float fx = 0.f, fy = 0.f, fz = 0.f;
            const float mi_over_ri = mi / ri;  // (per own particle: correctly rounded, as the reference's left-to-right product starts)
            for_each_ff_regs(c, gs, lo, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Lp[s];
                const float rj = Lr[s], gj = Lg[s];
                const bool same = Lm ? (Lm[s] == model) : true;
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                const float sc = same ? g * (fast_div(mi_over_ri * pj.w, rj) * (gi + gj) * 0.5f) * ts : 0.0f;
                fx += dx * sc; fy += dy * sc; fz += dz * sc;
            });
            a.x += fx; a.y += fy; a.z += fz;
        }
        if (bc != 0.0f) {
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                const float sc = g * (mi / ri * (pj.w * rho0) / rho0 * gi * bc * 0.25f);
                const float ex = dx * sc, ey = dy * sc, ez = dz * sc;
                a.x += ex / mi; a.y += ey / mi; a.z += ez / mi;
                if (c.bforce && !is_ghost(c, i)) apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(Bv[s].w), -ex, -ey, -ez);
            });
        }
        c.acc[i] = a;
    });
}
void launch_he2014_colors(const StepCtx& c, const TileLds& L, uint32_t model, float* colors, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_he2014_colors, c, L, model, colors, s);
    SALVA_LAUNCH_TILE(k_he2014_colors, c, L, L.bytes(24, 16, 4), s, c, model, colors);
}
void launch_he2014_gradc(const StepCtx& c, const TileLds& L, uint32_t model, const float* colors, float* gradcs, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_he2014_gradc, c, L, model, colors, gradcs, s);
    SALVA_LAUNCH_TILE(k_he2014_gradc, c, L, L.bytes(28, 0, 4), s, c, model, colors, gradcs);
}
void launch_he2014_forces(const StepCtx& c, const TileLds& L, uint32_t model, float tension, float boundary_tension,
                          const float* gradcs, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_he2014_forces, c, L, model, tension, boundary_tension, gradcs, s);
    SALVA_LAUNCH_TILE(k_he2014_forces, c, L, L.bytes(28, 32, 6), s, c, model, tension, boundary_tension, gradcs);
}

// ------------------------------------------------------------------------------------------------ WCSPH surface tension
// surface_tension/wcsph_surface_tension.rs:49-64: a_i += sum_{j same fluid} (x_i - x_j) (-t_f W_ij m_j / m_i).
// (The reference's boundary loop :66-83 indexes the boundaries with fluid-fluid contacts and panics; the host rejects
// a non-zero boundary coefficient before a kernel is launched.)
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_wcsph_tension(StepCtx c, uint32_t model, float tc) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), Lp);
    const uint32_t* Lm = nullptr;
    if (c.nmodels > 1) t.stage(c, static_cast<const uint32_t*>(c.model), Lm);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        float fx = 0.f, fy = 0.f, fz = 0.f;
        for_each_ff_regs(c, gs, lo, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Lp[s];
            const bool same = Lm ? (Lm[s] == model) : true;
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float wgt = kernel_weight(dx * dx + dy * dy + dz * dz, c.sc);
            const float sc = same ? fast_div(-tc * wgt * pj.w, pi.w) : 0.0f;
            fx += dx * sc; fy += dy * sc; fz += dz * sc;
        });
        float4 a = c.acc[i];
        a.x += fx; a.y += fy; a.z += fz;
        c.acc[i] = a;
    });
}
void launch_wcsph_tension(const StepCtx& c, const TileLds& L, uint32_t model, float tension, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_wcsph_tension, c, L, model, tension, s);
    SALVA_LAUNCH_TILE(k_wcsph_tension, c, L, L.bytes(20, 0, 2), s, c, model, tension);
}

}  // namespace SALVA_KNS
