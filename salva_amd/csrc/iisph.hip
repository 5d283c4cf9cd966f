// iisph.hip — IISPH (Ihmsen et al. 2013) relaxed-Jacobi pressure solver passes as tile kernels (tile.h).
//
// Behaviour specified by /root/reference/src/solver/pressure/iisph_solver.rs (line numbers per kernel).
// Step order (:643-711): predict_advection -> advance -> integrate -> d_ii -> p *= 0.5 -> rho* -> a_ii ->
// loop { sum_j d_ij p_j ; next pressures (+error) ; swap } -> velocity changes -> v += dv ; x += v dt ; dv = 0.
// Pressures persist across steps (warm start); they ride in dv.w so the per-step cell sort carries them.
#include <climits>

#include "bbox.h"
#include "kernels.h"
#include "tile.h"
#include "pairs.h"

namespace SALVA_KNS {
using namespace salva;

// `acceleration += gravity` (:550-554)
__global__ __launch_bounds__(BLOCK) void k_iisph_begin(StepCtx c, float gx, float gy, float gz, int acc_has_user) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (acc_has_user) a = c.acc[i];
    c.acc[i] = make_float4(a.x + gx, a.y + gy, a.z + gz, 0.0f);
}
void launch_iisph_begin(const StepCtx& c, float gx, float gy, float gz, bool acc_has_user, hipStream_t s) {
    if (c.n) k_iisph_begin<<<num_blocks(c.n), BLOCK, 0, s>>>(c, gx, gy, gz, acc_has_user ? 1 : 0);
}

// compute_dii (:144-186): d_ii = -dt^2 / rho_i^2 * sum_j m_j grad W_ij ; also p_i = 0.5 * p_i(previous step) (:673-677).
// Packed pair loop over P alone (kernel_gfac2; slices with a near-coincident pair take the exact walk).
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_iisph_dii(StepCtx c, float dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi; float rhoi; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) { return Own{c.posm[i], c.rho[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)}; };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const float4* Lp = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), Lp);  // first carve: LDS byte 0 (lds_ld16)
    const float4* Bp = nullptr;
    t.stage_boundary(c, Bp);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi;
        const float rho0 = rho0_of(c, o.mi);
        const float rhoi = o.rhoi;
        const float factor = -dt * dt / (rhoi * rhoi);
        float x = 0.f, y = 0.f, z = 0.f;
        if (near) {
            for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = lds_ld16(s << 4);
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float sc = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * pj.w;
                x += dx * sc; y += dy * sc; z += dz * sc;
            });
        } else {
            f2 ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f};
            const f2 tiny = {1.0e-30f, 1.0e-30f};
            for_each_ff2<true, false, true>(c, gs, nqu, o.lh, [&](uint32_t off) { return lds_ld16(off); }, [&](const float4& A, const float4& B) { SALVA_PAIR_MATH
                const f2 dx = {pi.x - A.x, pi.x - B.x}, dy = {pi.y - A.y, pi.y - B.y}, dz = {pi.z - A.z, pi.z - B.z};
                f2 r2 = dz * dz + tiny;
                r2 = dy * dy + r2;
                r2 = dx * dx + r2;
                const f2 gm = kernel_gfac2(r2, c.sc) * f2{A.w, B.w};
                ax += dx * gm; ay += dy * gm; az += dz * gm;
            });
            x = (ax.x + ax.y) * c.sc.gscale; y = (ay.x + ay.y) * c.sc.gscale; z = (az.x + az.y) * c.sc.gscale;
        }
        x *= factor; y *= factor; z *= factor;
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Bp[s];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float sc = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (pj.w * rho0 * factor);
            x += dx * sc; y += dy * sc; z += dz * sc;
        });
        c.dii[i] = make_float4(x, y, z, 0.0f);
        c.kappa[i] = c.dv[i].w * 0.5f;
        // what particle i contributes as a NEIGHBOUR in compute_dij_pjl: its position and m_i / rho_i^2 — constant over the Jacobi
        // loop, so k_iisph_dij_pj stages this record + the pressure instead of position, density and pressure
        c.iisph_pr[i] = make_float4(pi.x, pi.y, pi.z, pi.w / (rhoi * rhoi));
    });
}
void launch_iisph_dii(const StepCtx& c, const TileLds& L, float dt, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_iisph_dii, c, L, dt, s);
    SALVA_LAUNCH_TILE(k_iisph_dii, c, L, L.bytes(16, 16, 2), s, c, dt);
}

// compute_predicted_densities (:92-142): the same sum as DFSPH's (pairs.h pair_sum_velocity_divergence), on the same skeleton
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_iisph_pred_density(StepCtx c, float dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi, wi; float rho; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.w[i], c.rho[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pw_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pw(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.w), dist, Bp, Bv, true);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi, wi = o.wi;
        const float rho0 = rho0_of(c, o.mi);
        float delta = near ? pair_sum_velocity_divergence_exact(c, i, gs, pi, wi, dist)
                           : pair_sum_velocity_divergence(c, gs, nqu, o.lh, pi, wi, dist);
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Bp[s];
            const float4 vj = Bv[s];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            delta += pj.w * rho0 * (((wi.x - vj.x) * dx + (wi.y - vj.y) * dy + (wi.z - vj.z) * dz) * g);
        });
        const float rs = o.rho + delta * dt;
        if (!(rs != 0.0f)) atomicOr(c.flags, 1u);  // :140
        c.rho_star[i] = rs;
    });
}
// the plane-layout form (tile.h stage_p3; every particle has the mass c.mass_uniform): three tiles per CU, see dfsph.hip
#define SALVA_IISPH_P3_BOUNDS(DS) __launch_bounds__(TILE_MAX_THREADS, (DS) == P3_DS_THREE ? 6 : 5)
template <uint32_t DS>
__global__ SALVA_IISPH_P3_BOUNDS(DS) void k_iisph_pred_density_p3(StepCtx c, float dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    // (three-component vectors, and rho_i / the model are loaded after the loop: no register carries them across it — the kernel
    // held to 80 VGPRs for the third tile spilled eight before, none now)
    struct Own { float px, py, pz, ux, uy, uz; uint32_t cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        const float4 p = c.posm[i], u = c.w[i];
        return Own{p.x, p.y, p.z, u.x, u.y, u.z, c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p3_dist8<DS>(t);
    t.stage_p3(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.w), dist8);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    if (c.bvel_zero) t.stage_boundary(c, Bp);
    else t.stage_boundary(c, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = make_float4(o.px, o.py, o.pz, 0.0f), wi = make_float4(o.ux, o.uy, o.uz, 0.0f);
        const float rho0 = t.SB ? rho0_of(c, c.model[i]) : 0.0f;  // (only the boundary arm weighs by it)
        float delta = near ? pair_sum_velocity_divergence_exact_p3(c, i, gs, pi, wi, dist8, t.mass)
                           : pair_sum_velocity_divergence_p3(c, gs, nqu, o.lh, pi, wi, dist8, t.mass);
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Bp[s];
            const float4 vj = c.bvel_zero ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : Bv[s];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            delta += pj.w * rho0 * (((wi.x - vj.x) * dx + (wi.y - vj.y) * dy + (wi.z - vj.z) * dz) * g);
        });
        const float rs = c.rho[i] + delta * dt;
        if (!(rs != 0.0f)) atomicOr(c.flags, 1u);  // :140
        c.rho_star[i] = rs;
    });
}
void launch_iisph_pred_density(const StepCtx& c, const TileLds& L, float dt, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_iisph_pred_density, c, L, dt, s);
    if (c.mass_uniform > 0.0f) {
        const uint32_t ds = pick_ds_p3(L.max_halo_fluid, L.ds_level);
        SALVA_LAUNCH_P3(k_iisph_pred_density_p3, ds, c, L, p3_bytes(L, ds, c.nmodels, !c.bvel_zero), s, c, dt);
        return;
    }
    const uint32_t ds = pick_ds(pw_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_iisph_pred_density, ds, c, L, pw_bytes(L, ds, false), s, c, dt);
}

// compute_aii (:188-233): a_ii = sum_j m_j (d_ii - d_ji) . grad W_ij with d_ji = grad W_ij dt^2 m_i / rho_i^2.
// With grad W_ij = G d: m_j (d_ii - G d f) . G d = m_j G (d_ii . d) - f m_j G^2 |d|^2 — two packed sums over P alone.
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_iisph_aii(StepCtx c, float dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi, di; float rhoi; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.dii[i], c.rho[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const float4* Lp = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), Lp);  // first carve: LDS byte 0 (lds_ld16)
    const float4* Bp = nullptr;
    t.stage_boundary(c, Bp);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi, di = o.di;
        const float rho0 = rho0_of(c, o.mi);
        const float rhoi = o.rhoi;
        const float factor = dt * dt * pi.w / (rhoi * rhoi);
        float a = 0.0f;
        if (near) {
            for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = lds_ld16(s << 4);
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                const float gx = dx * g, gy = dy * g, gz = dz * g;
                a += pj.w * ((di.x - gx * factor) * gx + (di.y - gy * factor) * gy + (di.z - gz * factor) * gz);
            });
        } else {
            f2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
            const f2 tiny = {1.0e-30f, 1.0e-30f};
            for_each_ff2<true, false, true>(c, gs, nqu, o.lh, [&](uint32_t off) { return lds_ld16(off); }, [&](const float4& A, const float4& B) { SALVA_PAIR_MATH
                const f2 dx = {pi.x - A.x, pi.x - B.x}, dy = {pi.y - A.y, pi.y - B.y}, dz = {pi.z - A.z, pi.z - B.z};
                f2 r2 = dz * dz + tiny;
                r2 = dy * dy + r2;
                r2 = dx * dx + r2;
                const f2 g = kernel_gfac2(r2, c.sc);
                const f2 gm = g * f2{A.w, B.w};
                sa += (dx * di.x + dy * di.y + dz * di.z) * gm;
                sb += (g * gm) * r2;
            });
            a = (sa.x + sa.y) * c.sc.gscale - (sb.x + sb.y) * (c.sc.gscale * c.sc.gscale) * factor;
        }
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Bp[s];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            const float gx = dx * g, gy = dy * g, gz = dz * g;
            a += pj.w * rho0 * ((di.x - gx * factor) * gx + (di.y - gy * factor) * gy + (di.z - gz * factor) * gz);
        });
        c.aii[i] = a;
    });
}
void launch_iisph_aii(const StepCtx& c, const TileLds& L, float dt, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_iisph_aii, c, L, dt, s);
    SALVA_LAUNCH_TILE(k_iisph_aii, c, L, L.bytes(16, 16, 2), s, c, dt);
}

// compute_dij_pjl (:235-268): sum_j d_ij p_j = dt^2 sum_j grad W_ij (-m_j p_j / rho_j^2)   (fluid neighbours only).
// Fixed P | K layout (pairs.h pair_sum_gradient): P = (x_j, m_j / rho_j^2) written once per step by k_iisph_dii, K = p_j.
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_iisph_dij_pj(StepCtx c, float dt, const float* __restrict__ p) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi; uint32_t cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) { return Own{c.posm[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)}; };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pk_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pk(c, static_cast<const float4*>(c.iisph_pr), p, dist, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi;
        float x, y, z;
        if (near) pair_sum_gradient_exact(c, i, gs, pi, dist, [&](float kj) { return -kj; }, x, y, z);
        else pair_sum_gradient(c, gs, nqu, o.lh, pi, dist, [&](float ka, float kb) { return f2{-ka, -kb}; }, x, y, z);
        const float dt2 = dt * dt;
        const float4 dp = make_float4(x * dt2, y * dt2, z * dt2, 0.0f);
        c.dijpj[i] = dp;
        // what particle i contributes as a NEIGHBOUR in compute_next_pressures (:318-321): d_ii p_i + sum_k d_ik p_k — one
        // 16-byte record instead of d_ii, sum d_ij p_j and p staged separately (52 B per halo slot: one tile per CU)
        const float4 di = c.dii[i];
        const float pl = p[i];
        c.iisph_q[i] = make_float4(di.x * pl + dp.x, di.y * pl + dp.y, di.z * pl + dp.z, 0.0f);
    });
}
void launch_iisph_dij_pj(const StepCtx& c, const TileLds& L, float dt, const float* p, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_iisph_dij_pj, c, L, dt, p, s);
    const uint32_t ds = pick_ds(pk_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_iisph_dij_pj, ds, c, L, pk_bytes(L, ds), s, c, dt, p);
}

// compute_next_pressures (:270-353).  The pair loop of the Jacobi pass on the fixed-layout skeleton of the DFSPH solver kernels
// (pairs.h): P at LDS byte 0, q = d_jj p_j + sum_k d_jk p_k at a compile-time distance, two contacts per step in packed
// arithmetic with kernel_gfac2.  With grad W_ij = G d (G = gscale * gfac, d = x_i - x_j) the summand
//   m_j (dpi - q_j + grad W_ij fji p_i) . grad W_ij  =  m_j G (dpi - q_j) . d  +  m_j G^2 |d|^2 fji p_i,
// so the loop accumulates the two sums separately and the per-particle constants are applied once.  Slices that hold a pair
// closer than 1e-5 h (StepCtx::slice_near) walk their exact lists with kernel_grad, as in dfsph.hip.
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_iisph_next_pressure(StepCtx c, float dt, float omega,
                                                                     const float* __restrict__ p, float* __restrict__ p_next) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    struct Own { float4 pi, dpi; float a, prs, rhoi, rstar; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.dijpj[i], c.aii[i], p[i], c.rho[i], c.rho_star[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pw_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pw(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.iisph_q), dist, Bp, Bv, false);
    TileErr E;
    E.init(carve_errtab(t), c);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        float err = 0.0f;
        uint32_t mi = 0;
        if (active) {
            mi = o.mi;
            const float a = o.a;
            float pn = 0.0f;
            if (fabsf(a) > 1.0e-9f) {
                const float rho0 = rho0_of(c, mi);
                const float4 pi = o.pi, dpi = o.dpi;
                const float prs = o.prs;
                const float derr = rho0 - o.rstar;
                const float fji = dt * dt * pi.w / (o.rhoi * o.rhoi);
                float sum = 0.0f;
                if (near) {
                    for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                        const RecPW A = load_pw(s << 4, dist);
                        const float dx = pi.x - A.p.x, dy = pi.y - A.p.y, dz = pi.z - A.p.z;
                        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                        const float gx = dx * g, gy = dy * g, gz = dz * g;
                        // factor = dij_pjl[i] - dii[j] p_j - (dij_pjl[j] - dji p_i)   (:318-321), the two neighbour terms pre-added
                        const float fx = (dpi.x - A.w.x) + gx * fji * prs;
                        const float fy = (dpi.y - A.w.y) + gy * fji * prs;
                        const float fz = (dpi.z - A.w.z) + gz * fji * prs;
                        sum += A.p.w * (fx * gx + fy * gy + fz * gz);
                    });
                } else {
                    f2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
                    const f2 tiny = {1.0e-30f, 1.0e-30f};
                    for_each_ff2<true, false, true>(c, gs, nqu, o.lh, [&](uint32_t off) { return load_pw(off, dist); },
                                                    [&](const RecPW& A, const RecPW& B) { SALVA_PAIR_MATH
                        const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                        f2 r2 = dz * dz + tiny;
                        r2 = dy * dy + r2;
                        r2 = dx * dx + r2;
                        const f2 g = kernel_gfac2(r2, c.sc);
                        const f2 ex = {dpi.x - A.w.x, dpi.x - B.w.x}, ey = {dpi.y - A.w.y, dpi.y - B.w.y}, ez = {dpi.z - A.w.z, dpi.z - B.w.z};
                        const f2 gm = {g.x * A.p.w, g.y * B.p.w};
                        sa += (ex * dx + ey * dy + ez * dz) * gm;
                        sb += (g * gm) * r2;
                    });
                    sum = (sa.x + sa.y) * c.sc.gscale + (sb.x + sb.y) * (c.sc.gscale * c.sc.gscale) * (fji * prs);
                }
                for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                    const float4 pj = Bp[s];
                    const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                    sum += pj.w * rho0 * ((dpi.x * dx + dpi.y * dy + dpi.z * dz) * g);
                });
                pn = (1.0f - omega) * prs + omega * (derr - sum) / a;
                if (pn > 0.0f) err = (-sum - a * pn) / rho0;
                else pn = 0.0f;  // clamp negative pressures (:336-339)
            }
            p_next[i] = pn;
        }
        E.add(c, err, mi, active && !is_ghost(c, i));
    });
    E.finish(c, t.slot);
}
// the plane-layout form: planes (x, y) | (z, q.x) | (q.y, q.z), the uniform mass applied to the two finished sums
// (round 4: 87 VGPRs, two tiles per CU; held to 80 for a third tile it spilled nine registers — 61.8 us against 57.8, and 59.0 on
// the 32-byte layout, profiles/r04_experiments/r04l_iisph_next_pressure.log.  Round 5: 80 VGPRs without scratch, see the Own record below)
#ifndef SALVA_IISPH_NP_WAVES
#define SALVA_IISPH_NP_WAVES 6
#endif
#ifndef SALVA_IISPH_NP_NARROW
#define SALVA_IISPH_NP_NARROW false
#endif
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS, (DS) == P3_DS_THREE ? SALVA_IISPH_NP_WAVES : 5) void k_iisph_next_pressure_p3(StepCtx c, float dt, float omega, const float* __restrict__ p,
                                                                   float* __restrict__ p_next) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    // (the per-lane record holds what the LOOP needs and nothing else: the position and sum_j d_ij p_j.  Everything the result is
    // made of afterwards — a_ii, p_i, rho_i, rho*_i, m_i — is loaded after the loop (five cached words per particle), so that no
    // register carries it across the loop: the kernel then fits 80 VGPRs, i.e. a third resident tile)
    struct Own { float px, py, pz, ex, ey, ez; uint32_t cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        const float4 pi = c.posm[i], dpi = c.dijpj[i];
        return Own{pi.x, pi.y, pi.z, dpi.x, dpi.y, dpi.z, c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p3_dist8<DS>(t);
    t.stage_p3(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.iisph_q), dist8);
    const float4* Bp = nullptr;
    t.stage_boundary(c, Bp);
    TileErrC E;
    E.init(reinterpret_cast<float*>(t.pool + t.pool_used), c);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        float err = 0.0f;
        uint32_t mi = 0;
        if (active) {
            const float3 pi = make_float3(o.px, o.py, o.pz), dpi = make_float3(o.ex, o.ey, o.ez);
            // sum_j m_j (sum d p - q_j + grad W f p_i) . grad W with grad W = G d: m (sum_a + f p_i sum_b), the two sums below
            float suma = 0.0f, sumb = 0.0f;
            if (near) {
                for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                    const RecP3 A = load_p3(s << 3, dist8);
                    const float dx = pi.x - A.xy.x, dy = pi.y - A.xy.y, dz = pi.z - A.zu.x;
                    const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                    const float gx = dx * g, gy = dy * g, gz = dz * g;
                    suma += (dpi.x - A.zu.y) * gx + (dpi.y - A.vw.x) * gy + (dpi.z - A.vw.y) * gz;
                    sumb += gx * gx + gy * gy + gz * gz;
                });
            } else {
                f2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
                const f2 tiny = {1.0e-30f, 1.0e-30f};
                for_each_ff2<true, false, 2>(c, gs, nqu, o.lh, [&](uint32_t off) { return load_p3(off, dist8); },
                                             [&](const RecP3& A, const RecP3& B) { SALVA_PAIR_MATH
                    const f2 dx = {pi.x - A.xy.x, pi.x - B.xy.x}, dy = {pi.y - A.xy.y, pi.y - B.xy.y}, dz = {pi.z - A.zu.x, pi.z - B.zu.x};
                    f2 r2 = dz * dz + tiny;
                    r2 = dy * dy + r2;
                    r2 = dx * dx + r2;
                    const f2 g = kernel_gfac2(r2, c.sc);
                    const f2 ex = {dpi.x - A.zu.y, dpi.x - B.zu.y}, ey = {dpi.y - A.vw.x, dpi.y - B.vw.x}, ez = {dpi.z - A.vw.y, dpi.z - B.vw.y};
                    sa += (ex * dx + ey * dy + ez * dz) * g;
                    sb += (g * g) * r2;
                });
                suma = (sa.x + sa.y) * c.sc.gscale;
                sumb = (sb.x + sb.y) * (c.sc.gscale * c.sc.gscale);
            }
            float bsum = 0.0f;
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                bsum += pj.w * ((dpi.x * dx + dpi.y * dy + dpi.z * dz) * g);
            });
            // ---- the particle's own scalars, only now
            mi = c.model[i];
            const float a = c.aii[i];
            float pn = 0.0f;
            if (fabsf(a) > 1.0e-9f) {
                const float rho0 = rho0_of(c, mi);
                const float prs = p[i], rhoi = c.rho[i];
                const float derr = rho0 - c.rho_star[i];
                const float fp = dt * dt * c.posm[i].w / (rhoi * rhoi) * prs;  // f_ji p_i, f_ji = dt^2 m_i / rho_i^2
                const float sum = (suma + sumb * fp) * c.mass_uniform + rho0 * bsum;
                pn = (1.0f - omega) * prs + omega * (derr - sum) / a;
                if (pn > 0.0f) err = (-sum - a * pn) / rho0;
                else pn = 0.0f;  // clamp negative pressures (:336-339)
            }
            p_next[i] = pn;
        }
        E.add(c, err, mi, active && !is_ghost(c, i));
    });
    E.finish(c, t.slot);
}
void launch_iisph_next_pressure(const StepCtx& c, const TileLds& L, float dt, float omega, const float* p, float* p_next,
                                hipStream_t s) {
    SALVA_OK_DISPATCH(launch_iisph_next_pressure, c, L, dt, omega, p, p_next, s);
    if (c.mass_uniform > 0.0f) {
        const uint32_t ds = pick_ds_p3(L.max_halo_fluid, L.ds_level);
        SALVA_LAUNCH_P3(k_iisph_next_pressure_p3, ds, c, L, p3_bytes(L, ds, c.nmodels, false), s, c, dt, omega, p, p_next);
        return;
    }
    const uint32_t ds = pick_ds(pw_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_iisph_next_pressure, ds, c, L, pw_bytes(L, ds, true), s, c, dt, omega, p, p_next);
}

// compute_velocity_changes (:355-404): dv_i -= dt sum_j grad W_ij m_j (p_i / rho_i^2 + p_j / rho_j^2) (+ the boundary term with its
// reaction force).  Fixed P | K layout: K = p_j / rho_j^2, written for every particle (ghosts included: their pressures were
// refreshed) by k_iisph_pr2 just before — into StepCtx::alpha, which IISPH does not use.
__global__ __launch_bounds__(BLOCK) void k_iisph_pr2(StepCtx c, const float* __restrict__ p) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const float r = c.rho[i];
    c.alpha[i] = p[i] / (r * r);
}
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_iisph_velocity_changes(StepCtx c, float dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi, d; float pri; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.dv[i], c.alpha[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pk_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pk(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.alpha), dist, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi;
        const float rho0 = rho0_of(c, o.mi);
        const float pri = o.pri;
        float4 d = o.d;
        float sx, sy, sz;
        if (near) pair_sum_gradient_exact(c, i, gs, pi, dist, [&](float kj) { return pri + kj; }, sx, sy, sz);
        else pair_sum_gradient(c, gs, nqu, o.lh, pi, dist, [&](float ka, float kb) { return f2{pri + ka, pri + kb}; }, sx, sy, sz);
        d.x -= sx * dt; d.y -= sy * dt; d.z -= sz * dt;
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Bp[s];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float sc = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (pj.w * rho0 * pri);
            const float ax = dx * sc, ay = dy * sc, az = dz * sc;
            d.x -= ax * dt; d.y -= ay * dt; d.z -= az * dt;
            if (c.bforce && !is_ghost(c, i))
                apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(Bv[s].w), ax * pi.w, ay * pi.w, az * pi.w);
        });
        c.dv[i] = d;
    });
}
// the 16-byte plane form (tile.h stage_p2)
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS, (DS) == P2_DS_THREE ? 6 : 5) void k_iisph_velocity_changes_p2(StepCtx c, float dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi, d; float pri; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.dv[i], c.alpha[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p2_dist8<DS>(t);
    t.stage_p2(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.alpha), dist8);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi;
        const float rho0 = rho0_of(c, o.mi);
        const float pri = o.pri;
        float4 d = o.d;
        float sx, sy, sz;
        if (near) pair_sum_gradient_exact_p2(c, i, gs, pi, dist8, t.mass, [&](float kj) { return pri + kj; }, sx, sy, sz);
        else pair_sum_gradient_p2(c, gs, nqu, o.lh, pi, dist8, t.mass, [&](float ka, float kb) { return f2{pri + ka, pri + kb}; }, sx, sy, sz);
        d.x -= sx * dt; d.y -= sy * dt; d.z -= sz * dt;
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = p2_boundary_pos(t, s, dist8);
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float sc = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (pj.w * rho0 * pri);
            const float ax = dx * sc, ay = dy * sc, az = dz * sc;
            d.x -= ax * dt; d.y -= ay * dt; d.z -= az * dt;
            if (c.bforce && !is_ghost(c, i)) {
                const uint32_t jb = boundary_sorted_of_slot(c, t, s);
                apply_boundary_force(c, jb, __float_as_uint(c.bvel[jb].w), ax * pi.w, ay * pi.w, az * pi.w);
            }
        });
        c.dv[i] = d;
    });
}
void launch_iisph_velocity_changes(const StepCtx& c, const TileLds& L, float dt, const float* p, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_iisph_velocity_changes, c, L, dt, p, s);
    if (!c.n) return;
    k_iisph_pr2<<<num_blocks(c.n), BLOCK, 0, s>>>(c, p);
    if (c.mass_uniform > 0.0f) {
        const uint32_t ds = pick_ds_p2(L.raw_slots(), L.ds_level);
        SALVA_LAUNCH_P2(k_iisph_velocity_changes_p2, ds, c, L, p2_bytes(L, ds), s, c, dt);
        return;
    }
    const uint32_t ds = pick_ds(pk_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_iisph_velocity_changes, ds, c, L, pk_bytes(L, ds), s, c, dt);
}

// update_velocities_and_positions (:406-420) + zero velocity changes (:707-709); stores the pressure for the
// next step's warm start and reduces the new cell bounding box.
__global__ __launch_bounds__(BLOCK) void k_iisph_finish(StepCtx c, float dt, const float* __restrict__ p, int32_t* bbox_partials) {
    __shared__ int red[6 * (BLOCK / WAVE)];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    if (i < c.n) {
        float4 v = c.vel[i];
        const float4 d = c.dv[i];
        float4 x = c.posm[i];
        v.x += d.x; v.y += d.y; v.z += d.z;
        x.x += v.x * dt; x.y += v.y * dt; x.z += v.z * dt;
        c.vel[i] = v;
        c.posm[i] = x;
        c.dv[i] = make_float4(0.f, 0.f, 0.f, p[i]);
        bool bad = false;
        mn[0] = mx[0] = cell_coord(x.x, c.sc.h, bad);
        mn[1] = mx[1] = cell_coord(x.y, c.sc.h, bad);
        mn[2] = mx[2] = cell_coord(x.z, c.sc.h, bad);
        if (bad) atomicOr(c.flags, 1u);
    }
    block_bbox_store(mn, mx, red, bbox_partials + 6 * blockIdx.x);
}
void launch_iisph_finish(const StepCtx& c, float dt, const float* p, int32_t* bbox_partials, int32_t* bbox6, hipStream_t s) {
    if (!c.n) return;
    k_iisph_finish<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt, p, bbox_partials);
    if (bbox6) launch_bbox_final(bbox_partials, num_blocks(c.n), bbox6, s);  // (nullptr: the end-of-step publication folds them)
}

}  // namespace SALVA_KNS
