// dfsph.hip — DFSPH pressure solver passes on LDS-staged cell tiles (tile.h).
//
// Behaviour specified by /root/reference/src/solver/pressure/dfsph_solver.rs (line numbers cited per kernel)
// and src/solver/helper.rs:9-65 (W / grad W are recomputed in-kernel from positions instead of being stored per
// contact).  One workgroup per tile: the halo's 16-byte records are copied to LDS once, then one lane per particle
// walks its list of 16-bit halo slots.  Contacts are directed (contacts.rs:40-55), so every fluid quantity is a
// pure gather — no atomics; only the optional boundary reaction forces are accumulated atomically.
#include <climits>

#include "bbox.h"
#include "kernels.h"
#include "tile.h"

namespace SALVA_KNS {
using namespace salva;

unsigned num_blocks(uint32_t n) { return div_up(n, BLOCK); }

}  // namespace SALVA_KNS
#include "pairs.h"
namespace SALVA_KNS {
using namespace salva;

// ------------------------------------------------------------------------------------------------
// compute_densities (dfsph_solver.rs:628-665) fused with compute_alphas (:165-216): both depend on positions only.
//   rho_i   = sum_j m_j W_ij + sum_b V_b rho0_i W_ib
//   alpha_i = 1 / (sum |m_j grad W_ij|^2 + |sum m_j grad W_ij|^2), 0 if the denominator <= 1e-5
// Also: c.slice_near[slice] = some particle of the slice has a neighbour (other than itself) with |d| <= 1e-5 h — the pairs for
// which cubic_spline_kernel.rs:63-65 returns a zero gradient; the solver kernels pick their pair loop by it.
// ------------------------------------------------------------------------------------------------
// Round 3: the fluid-fluid part walks the padded list two contacts at a time with kernel_wg2 (weight and gradient factor from the
// same intermediates, no branch); a slice in which the walk meets a pair closer than 1e-5 h — where the reference's gradient is
// zero — is summed again with kernel_eval over the exact lists.  The padding entries of a list are the particle itself: their
// gradients vanish exactly and zero-distance weights are masked out of the sum (the self weight is added once, separately).
// Also writes posmr = (x, m / rho): what a neighbour contributes to the sums that weigh by volume m_j / rho_j (XSPH).
// IISPH = true (single-domain IISPH worlds): d_ii = -dt^2 / rho_i^2 (sum_j m_j grad W_ij + sum_b V_b rho0 grad W_ib) (iisph_solver.rs:144-186) is
// that gradient sum times a factor of the density this pass has just finished — so the pass writes d_ii, p_i = p_i(previous step) / 2
// (:673-677) and the (x, m / rho^2) record of k_iisph_dij_pj itself, and k_iisph_dii (a whole neighbour pass: 39 us at 10^6
// particles) is not launched; alpha, which IISPH never reads, and its sum of squares are not computed.  The factor multiplies the
// finished sum instead of every boundary term: rounding only.  a_ii (:188-233) is made of own quantities and sums over the same
// contacts as well — d_ii . G_i - dt^2 m_i / rho_i^2 sum_j m_j |grad W_ij|^2 — and comes out of this pass too (no k_iisph_aii:
// another 38 us).  Decomposed runs keep the separate passes (a ghost's density is replaced by its owner's in between).
// The plane-layout kernels serve worlds with one particle mass (StepCtx::mass_uniform) and two-mass worlds (StepCtx::two_mass).
static inline bool plane_layouts(const StepCtx& c) { return c.mass_uniform > 0.0f || c.two_mass != 0u; }

#ifndef SALVA_DA_IISPH_WAVES
#define SALVA_DA_IISPH_WAVES 6  // (the IISPH form asks for 84 VGPRs by itself: two tiles per CU; held to 80 = three, two registers in scratch)
#endif
template <bool IISPH>
__global__ __launch_bounds__(TILE_MAX_THREADS, IISPH ? SALVA_DA_IISPH_WAVES : 1) void k_density_alpha(StepCtx c, float dt) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi; uint32_t mi, cnt; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) { return Own{c.posm[i], c.model[i], c.nff[i], list_regs(c, gs)}; };
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
        const float4 pi = o.pi;
        float rho = 0.0f, gsx = 0.0f, gsy = 0.0f, gsz = 0.0f, sq = 0.0f;
        uint32_t nnear = 0;
        if (active) {  // (an idle lane's list row was never written)
            f2 rw = {0.0f, 0.0f}, ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f}, s2 = {0.0f, 0.0f};
            const f2 tiny = {1.0e-30f, 1.0e-30f};
            for_each_ff2<true, false, true>(c, gs, nqu, o.lh, [&](uint32_t off) { return lds_ld16(off); }, [&](const float4& A, const float4& B) { SALVA_PAIR_MATH
                const f2 dx = {pi.x - A.x, pi.x - B.x}, dy = {pi.y - A.y, pi.y - B.y}, dz = {pi.z - A.z, pi.z - B.z};
                f2 r2 = dz * dz + tiny;
                r2 = dy * dy + r2;
                r2 = dx * dx + r2;
                const bool za = r2.x <= c.sc.tiny_r2, zb = r2.y <= c.sc.tiny_r2;
                nnear += (za ? 1u : 0u) + (zb ? 1u : 0u);
                const KernelWG2 k = kernel_wg2(r2, c.sc);
                const f2 m = {A.w, B.w};
                // the particle itself — its one real self contact and the padding — adds nothing here: its weight, W(0) m_i, goes
                // in once below, so that the sum does not depend on how long the slice's longest list is (a decomposed run cuts
                // the slices differently and must still produce the same bits)
                f2 wm = k.w * m;
                wm.x = za ? 0.0f : wm.x; wm.y = zb ? 0.0f : wm.y;
                rw += wm;
                const f2 gm = k.g * m;
                ax += dx * gm; ay += dy * gm; az += dz * gm;
                if (!IISPH) s2 += (gm * gm) * r2;  // sum |m_j grad W_ij|^2 (alpha)
                else s2 += (k.g * gm) * r2;        // sum m_j |grad W_ij|^2 (a_ii)
            });
            const uint32_t npad = 2u * nqu - o.cnt;  // self contacts appended by k_nbr_tile
            rho = (rw.x + rw.y) * c.sc.wscale + pi.w * c.sc.wnorm;
            gsx = (ax.x + ax.y) * c.sc.gscale; gsy = (ay.x + ay.y) * c.sc.gscale; gsz = (az.x + az.y) * c.sc.gscale;
            sq = (s2.x + s2.y) * (c.sc.gscale * c.sc.gscale);
            nnear -= npad;  // (the self contact itself stays counted: "> 1" below means another particle)
        }
        // (the self contact is always one of the pairs with r2 <= tiny)
#ifdef SALVA_OTHER_KERNELS  // (kernel_wg2 is the cubic spline: another KernelDensity / KernelGradient takes the exact walk)
        const bool any_near = __builtin_amdgcn_ballot_w64(active && nnear > 1u) != 0ull || (c.sc.kd | c.sc.kg) != 0;
#else
        const bool any_near = __builtin_amdgcn_ballot_w64(active && nnear > 1u) != 0ull;
#endif
        if (any_near && active) {  // rare: the slice's sums again, the reference's way
            rho = gsx = gsy = gsz = sq = 0.0f;
            for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = lds_ld16(s << 4);
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const KernelEval e = kernel_eval(dx * dx + dy * dy + dz * dz, c.sc);
                rho += pj.w * e.w;
                const float gm = e.g * pj.w;
                const float gx = dx * gm, gy = dy * gm, gz = dz * gm;
                if (!IISPH) sq += gx * gx + gy * gy + gz * gz;
                else sq += (dx * e.g) * gx + (dy * e.g) * gy + (dz * e.g) * gz;
                gsx += gx; gsy += gy; gsz += gz;
            });
        }
        if (active) {
            const float rho0 = rho0_of(c, o.mi);
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const KernelEval e = kernel_eval(dx * dx + dy * dy + dz * dz, c.sc);
                const float m = pj.w * rho0;
                rho += m * e.w;
                const float gm = e.g * m;
                const float gx = dx * gm, gy = dy * gm, gz = dz * gm;
                if (!IISPH) sq += gx * gx + gy * gy + gz * gz;
                else sq += (dx * e.g) * gx + (dy * e.g) * gy + (dz * e.g) * gz;
                gsx += gx; gsy += gy; gsz += gz;
            });
            if (!(rho > 0.0f)) atomicOr(c.flags, 1u);  // assert!(!density.is_zero()) :662
            c.rho[i] = rho;
            if (IISPH) {
                // d_ii = -dt^2 / rho_i^2 G_i with G_i = sum m_j grad W_ij (fluid and boundary neighbours), and
                // a_ii = sum m_j (d_ii - d_ji) . grad W_ij with d_ji = grad W_ij dt^2 m_i / rho_i^2 (iisph_solver.rs:188-233)
                //      = d_ii . G_i - dt^2 m_i / rho_i^2 sum m_j |grad W_ij|^2: own quantities and the two sums of this pass
                const float f0 = dt * dt / (rho * rho);
                const float dxi = -(gsx * f0), dyi = -(gsy * f0), dzi = -(gsz * f0);
                c.dii[i] = make_float4(dxi, dyi, dzi, 0.0f);
                c.aii[i] = (dxi * gsx + dyi * gsy + dzi * gsz) - (f0 * pi.w) * sq;
                c.kappa[i] = c.dv[i].w * 0.5f;
                c.iisph_pr[i] = make_float4(pi.x, pi.y, pi.z, pi.w / (rho * rho));
            } else {
                const float denom = sq + (gsx * gsx + gsy * gsy + gsz * gsz);
                c.alpha[i] = (denom <= 1.0e-5f) ? 0.0f : 1.0f / denom;
            }
            c.posmr[i] = make_float4(pi.x, pi.y, pi.z, pi.w / rho);
        }
        if ((threadIdx.x & (WAVE - 1)) == 0) c.slice_near[gs] = any_near ? 1u : 0u;
    });
}
// iisph_dt > 0: the IISPH form (d_ii and its companions ride along, see above)
void launch_density_alpha(const StepCtx& c, const TileLds& L, float iisph_dt, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_density_alpha, c, L, iisph_dt, s);
    if (iisph_dt > 0.0f) SALVA_LAUNCH_TILE(k_density_alpha<true>, c, L, L.bytes(16, 16, 2), s, c, iisph_dt);
    else SALVA_LAUNCH_TILE(k_density_alpha<false>, c, L, L.bytes(16, 16, 2), s, c, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// compute_densities + compute_alphas + the FIRST compute_divergences of the step (dfsph_solver.rs:628-665, :165-216, :279-356) in
// one pass.  The first evaluate of the divergence solve sums m_j (w_i - w_j) . grad W_ij over the same lists with the gradient
// factor this pass computes anyway, and needs nothing the density pass has not got (alpha_i follows from the particle's own
// sums; w = v + dv is written by the reorder): staging w next to the positions — the plane layout of the evaluate kernels — and
// ~8 more VALU per pair of contacts replace a whole neighbour pass, ~40 us of every DFSPH step at 10^6 particles.  Taken when
// every particle has the same mass (the plane layout's condition; the mass multiplies the finished sums) in a DFSPH world with
// the default kernels; World::dfsph_solve then skips the launch of its iteration 0.  Writes what k_density_alpha writes (rho,
// alpha, posmr, slice_near) and what k_divergence writes (kappa = D rho alpha, the tile's error partials).
// ------------------------------------------------------------------------------------------------
// (96 VGPRs, two tiles per CU: held to 80 for a third tile the kernel spills 20 registers and is slower — free-fall step 0.664
// against 0.656 ms, 0.683 without the fusion; profiles/r04_experiments/r04j_fused_first_divergence.log)
#ifndef SALVA_DAD_WAVES
#define SALVA_DAD_WAVES 5
#endif
// (one body, two instantiations: the uniform-mass kernel carries none of the two-mass code — its second pair loop, the extra
// list word — so that worlds with one mass run exactly what they ran before; device_types.h StepCtx::two_mass)
template <uint32_t DS, int TWO>  // TWO: 0 one mass, 1 two masses, 2 three or four (StepCtx::nmass)
__device__ __forceinline__ void k_density_alpha_div_p3_body(StepCtx c) {
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    struct Own { float px, py, pz, ux, uy, uz; uint32_t cnt; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        const float4 p = c.posm[i], u = c.w[i];
        return Own{p.x, p.y, p.z, u.x, u.y, u.z, c.nff[i], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p3_dist8<DS>(t);
    t.stage_p3(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.w), dist8);
    const float4* Bp = nullptr;
    t.stage_boundary(c, Bp);
    TileErrC E;
    E.init(reinterpret_cast<float*>(t.pool + t.pool_used), c);
    Tile::staged_barrier();
    const float m = t.mass;
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        // (the own mass, the model and the boundary count are loaded after the loop: nothing the loop does not need is carried across it)
        const float3 pi = make_float3(o.px, o.py, o.pz), wi = make_float3(o.ux, o.uy, o.uz);
        float rho = 0.0f, gsx = 0.0f, gsy = 0.0f, gsz = 0.0f, sq = 0.0f, div = 0.0f;
        uint32_t nnear = 0;
        if (active) {  // (an idle lane's list row was never written)
            f2 rw = {0.0f, 0.0f}, ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f}, s2 = {0.0f, 0.0f}, dv2 = {0.0f, 0.0f};
            const f2 tiny = {1.0e-30f, 1.0e-30f};
            for_each_ff2<true, false, 2>(c, gs, nqu, o.lh, [&](uint32_t off) { return load_p3(off, dist8); }, [&](const RecP3& A, const RecP3& B) { SALVA_PAIR_MATH
                const f2 dx = {pi.x - A.xy.x, pi.x - B.xy.x}, dy = {pi.y - A.xy.y, pi.y - B.xy.y}, dz = {pi.z - A.zu.x, pi.z - B.zu.x};
                f2 r2 = dz * dz + tiny;
                r2 = dy * dy + r2;
                r2 = dx * dx + r2;
                const bool za = r2.x <= c.sc.tiny_r2, zb = r2.y <= c.sc.tiny_r2;
                nnear += (za ? 1u : 0u) + (zb ? 1u : 0u);
                const KernelWG2 k = kernel_wg2(r2, c.sc);
                // (the particle itself — its one real self contact and the padding — adds nothing: its weight goes in once below)
                f2 wm = k.w;
                wm.x = za ? 0.0f : wm.x; wm.y = zb ? 0.0f : wm.y;
                rw += wm;
                ax += dx * k.g; ay += dy * k.g; az += dz * k.g;
                s2 += (k.g * k.g) * r2;
                const f2 ux = {wi.x - A.zu.y, wi.x - B.zu.y}, uy = {wi.y - A.vw.x, wi.y - B.vw.x}, uz = {wi.z - A.vw.y, wi.z - B.vw.y};
                dv2 += (ux * dx + uy * dy + uz * dz) * k.g;
            });
            const uint32_t npad = 2u * nqu - o.cnt;  // self contacts appended by k_nbr_tile
            const float gm = c.sc.gscale * m;
            // (the particle's own weight takes its OWN mass: in a two-mass world it may be the heavier one in a tile whose first
            // segment is the lighter; with one mass the two are the same number)
            rho = (rw.x + rw.y) * c.sc.wscale * m + c.posm[i].w * c.sc.wnorm;
            gsx = (ax.x + ax.y) * gm; gsy = (ay.x + ay.y) * gm; gsz = (az.x + az.y) * gm;
            sq = (s2.x + s2.y) * (gm * gm);
            div = (dv2.x + dv2.y) * gm;
            nnear -= npad;  // (the self contact itself stays counted: "> 1" below means another particle)
        }
        const bool any_near = __builtin_amdgcn_ballot_w64(active && nnear > 1u) != 0ull;
        if (any_near && active) {  // rare: the slice's sums again, the reference's way
            rho = gsx = gsy = gsz = sq = div = 0.0f;
            for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const RecP3 A = load_p3(s << 3, dist8);
                const float dx = pi.x - A.xy.x, dy = pi.y - A.xy.y, dz = pi.z - A.zu.x;
                const KernelEval e = kernel_eval(dx * dx + dy * dy + dz * dz, c.sc);
                rho += m * e.w;
                const float gmj = e.g * m;
                const float gx = dx * gmj, gy = dy * gmj, gz = dz * gmj;
                sq += gx * gx + gy * gy + gz * gz;
                gsx += gx; gsy += gy; gsz += gz;
                div += ((wi.x - A.zu.y) * dx + (wi.y - A.vw.x) * dy + (wi.z - A.vw.y) * dz) * gmj;
            });
        }
        if (active && (TWO && t.massb != 0.0f)) {  // several masses in this tile's halo: the heavier neighbours' shares on top of every sum
            // entries [first, last) of the list carry `mnew` where the sums so far counted them with `mold`
            auto on_top = [&](uint32_t first, uint32_t last, float mold, float mnew) {
                float rwb = 0.0f, gxb = 0.0f, gyb = 0.0f, gzb = 0.0f, s2b = 0.0f, dvb = 0.0f;
                for_each_ff_range(c, gs, first, last, [&](uint32_t s) { SALVA_PAIR_MATH
                    const RecP3 A = load_p3(s << 3, dist8);
                    const float dx = pi.x - A.xy.x, dy = pi.y - A.xy.y, dz = pi.z - A.zu.x;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    const KernelEval e = kernel_eval(r2, c.sc);
                    // (the packed walk left zero-distance weights out and the particle's own weight went in apart; the exact walk of
                    // a slice with a near-coincident pair summed every entry)
                    if (any_near || r2 > c.sc.tiny_r2) rwb += e.w;
                    const float gx = dx * e.g, gy = dy * e.g, gz = dz * e.g;
                    gxb += gx; gyb += gy; gzb += gz;
                    s2b += gx * gx + gy * gy + gz * gz;
                    dvb += ((wi.x - A.zu.y) * dx + (wi.y - A.vw.x) * dy + (wi.z - A.vw.y) * dz) * e.g;
                });
                const float dm = mnew - mold;
                rho += dm * rwb; gsx += dm * gxb; gsy += dm * gyb; gsz += dm * gzb;
                sq += (mnew * mnew - mold * mold) * s2b;
                div += dm * dvb;
            };
            const uint32_t nb2 = c.nffb[i];
            if (nb2) on_top(o.cnt - nb2, o.cnt, m, t.massb);
            if (TWO == 2 && t.massc != 0.0f) {  // (three or four masses: the third and fourth segment were just counted with the second one's)
                const uint32_t cd = c.nffc[i], l2 = cd & 0xffffu, l3 = cd >> 16;
                if (l2) on_top(o.cnt - l3 - l2, o.cnt - l3, t.massb, t.massc);
                if (l3) on_top(o.cnt - l3, o.cnt, t.massb, t.massd);
            }
        }
        float err = 0.0f;
        uint32_t mi = 0;
        if (active) {
            mi = c.model[i];
            const float rho0 = rho0_of(c, mi);
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const KernelEval e = kernel_eval(dx * dx + dy * dy + dz * dz, c.sc);
                const float mb = pj.w * rho0;
                rho += mb * e.w;
                const float gmb = e.g * mb;
                const float gx = dx * gmb, gy = dy * gmb, gz = dz * gmb;
                sq += gx * gx + gy * gy + gz * gz;
                gsx += gx; gsy += gy; gsz += gz;
                div += (wi.x * dx + wi.y * dy + wi.z * dz) * gmb;  // boundary velocity ignored (:332-333)
            });
            if (!(rho > 0.0f)) atomicOr(c.flags, 1u);  // assert!(!density.is_zero()) :662
            const float denom = sq + (gsx * gsx + gsy * gsy + gsz * gsz);
            const float alpha = (denom <= 1.0e-5f) ? 0.0f : 1.0f / denom;
            c.rho[i] = rho;
            c.alpha[i] = alpha;
            c.posmr[i] = make_float4(pi.x, pi.y, pi.z, c.posm[i].w / rho);
            const uint32_t cntb = c.nb ? c.nfb[i] : 0u;
            div = (o.cnt + cntb >= c.min_neighbors_for_divergence) ? fmaxf(div, 0.0f) : 0.0f;
            c.kappa[i] = div * alpha;
            err = div / rho0;
        }
        if ((threadIdx.x & (WAVE - 1)) == 0) c.slice_near[gs] = any_near ? 1u : 0u;
        E.add(c, err, mi, active && !is_ghost(c, i));
    });
    E.finish(c, t.slot);
}
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS, (DS) == P3_DS_THREE ? SALVA_DAD_WAVES : 5) void k_density_alpha_div_p3(StepCtx c) { k_density_alpha_div_p3_body<DS, 0>(c); }
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS, (DS) == P3_DS_THREE ? SALVA_DAD_WAVES : 5) void k_density_alpha_div_p3_two(StepCtx c) { k_density_alpha_div_p3_body<DS, 1>(c); }
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS, (DS) == P3_DS_THREE ? SALVA_DAD_WAVES : 5) void k_density_alpha_div_p3_multi(StepCtx c) { k_density_alpha_div_p3_body<DS, 2>(c); }
// true: the pass above ran and the divergence solve's iteration 0 must not launch its evaluate
bool launch_density_alpha_div(const StepCtx& c, const TileLds& L, hipStream_t s) {
#ifndef SALVA_OTHER_KERNELS
    if (!plane_layouts(c) || (c.sc.kd | c.sc.kg) != 0) return false;
    const uint32_t ds = pick_ds_p3(L.max_halo_fluid, L.ds_level);
    if (c.two_mass && c.nmass > 2u) SALVA_LAUNCH_P3(k_density_alpha_div_p3_multi, ds, c, L, p3_bytes(L, ds, c.nmodels, false), s, c);
    else if (c.two_mass) SALVA_LAUNCH_P3(k_density_alpha_div_p3_two, ds, c, L, p3_bytes(L, ds, c.nmodels, false), s, c);
    else SALVA_LAUNCH_P3(k_density_alpha_div_p3, ds, c, L, p3_bytes(L, ds, c.nmodels, false), s, c);
    return true;
#else
    return false;
#endif
}

// ------------------------------------------------------------------------------------------------
// Speculative applies (StepCtx::spec_k >= 0; single-domain divergence solves that ran more than a few iterations last step).
// The convergence test needs the error sums of the WHOLE evaluate pass, which is why it was a one-workgroup kernel between
// evaluate and apply — 4.7 us plus two launch gaps, a hundred times per settled step.  Here the apply pass does not wait for it:
// every workgroup computes w_out = w_in + delta into the OTHER w buffer, and workgroup 0 also reduces the evaluate pass's
// partials (same order as k_finalize_error: the same bits, hence the same decision) and writes the NEXT iteration's control
// record: converged -> the output just produced is never looked at (w stays where it was); not converged -> iters + 1, and
// the parity of iters is where w lives.  Records alternate (spec_ring[k & 1] is read by iteration k, [(k + 1) & 1] written), so
// workgroups of this launch that start after workgroup 0 has finished still read the record this launch was enqueued for.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ const SolveCtl* solve_record(const StepCtx& c) { return c.spec_k >= 0 ? c.spec_ring + (c.spec_k & 1) : c.ctl; }
__device__ __forceinline__ const float4* solve_w_in(const StepCtx& c, const SolveCtl* r) {
    return (c.spec_k >= 0 && (r->iters & 1u)) ? c.w2 : c.w;
}
// all threads of workgroup 0 of an apply pass; `scratch`: 16 floats of LDS nobody uses yet
__device__ __forceinline__ void spec_decide(const StepCtx& c, float* scratch) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const SolveCtl* rk = c.spec_ring + (c.spec_k & 1);
    SolveCtl* rn = c.spec_ring + ((c.spec_k + 1) & 1);
    const u32x4 c0 = reinterpret_cast<const u32x4*>(rk)[0], c1 = reinterpret_cast<const u32x4*>(rk)[1];
    const uint32_t done = c0.x, iters = c0.y, seq = c0.w + 1u, min_iter = c1.y;
    const float tol = __uint_as_float(c1.x);
    u32x4 out = {1u, iters, c0.z, seq};
    if (!done) {
        // the sums of k_finalize_error's 1024 threads, on however many threads this workgroup has: virtual thread v = t + j T
        // (T a multiple of 64: a virtual wave stays one real wave), wave sums by DPP, the 16 wave sums added in wave order
        float best = 0.0f;
        const uint32_t nb = c.nlaunch, nm = c.nmodels;
        for (uint32_t m = 0; m < nm; ++m) {
            for (uint32_t v = threadIdx.x; v < 1024u; v += blockDim.x) {
                float s = 0.0f;
                for (uint32_t b = v; b < nb; b += 1024u) s += c.partials[(size_t)b * nm + m];
                s = wave_sum(s);
                if ((v & 63u) == 0u) scratch[v >> 6] = s;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                float tot = 0.0f;
                for (int k = 0; k < 16; ++k) tot += scratch[k];
                const uint32_t cnt = c.model_counts[m];
                if (cnt != 0) best = fmaxf(best, tot / (float)cnt);
            }
            __syncthreads();
        }
        const bool ok = best <= tol && iters >= min_iter;
        out = u32x4{ok ? 1u : 0u, ok ? iters : iters + 1u, __float_as_uint(best), seq};
    }
    if (threadIdx.x == 0) {
        reinterpret_cast<u32x4*>(rn)[0] = out;
        reinterpret_cast<u32x4*>(rn)[1] = c1;
        if (c.spec_pub) *reinterpret_cast<volatile u32x4*>(c.spec_pub) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// compute_divergences (:279-356): D rho_i = max(sum_j m_j (w_i - w_j).grad W_ij + sum_b V_b rho0 w_i.grad W_ib, 0),
// skipped (0) when the particle has fewer than 20 contacts; stores kappa_i = D rho_i * alpha_i (the only use of
// the divergence, :370,:382) and the per-particle error D rho_i / rho0.
// ------------------------------------------------------------------------------------------------
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_divergence(StepCtx c) {
    const SolveCtl* const rec = solve_record(c);
    if (rec && rec->done) return;  // the solve converged earlier in this batch
    const float4* const win = rec ? solve_w_in(c, rec) : c.w;
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.skipped()) return;  // the other launch of this pass handles the tile (decomposed runs)
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    struct Own { float4 pi, wi; float alpha; uint32_t mi, cnt, cntb, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], win[i], c.alpha[i], c.model[i], c.nff[i], c.nb ? c.nfb[i] : 0u, c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pw_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pw(c, static_cast<const float4*>(c.posm), win, dist, Bp, Bv, false);
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
            const float rho0 = rho0_of(c, mi);
            float div = 0.0f;
            if (o.cnt + o.cntb >= c.min_neighbors_for_divergence) {
                const float4 pi = o.pi, wi = o.wi;
                div += near ? pair_sum_velocity_divergence_exact(c, i, gs, pi, wi, dist)
                            : pair_sum_velocity_divergence(c, gs, nqu, o.lh, pi, wi, dist);
                for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                    const float4 pj = Bp[s];
                    const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                    div += (wi.x * dx + wi.y * dy + wi.z * dz) * g * (pj.w * rho0);  // boundary velocity ignored (:332-333)
                });
                div = fmaxf(div, 0.0f);
            }
            c.kappa[i] = div * o.alpha;
            err = div / rho0;
        }
        E.add(c, err, mi, active && !is_ghost(c, i));  // a ghost's error belongs to the rank that owns it
    });
    E.finish(c, t.slot);
}
// The same pass over the 24-byte plane layout (tile.h stage_p3), taken when every particle has the same mass
// (c.mass_uniform > 0): three tiles per CU instead of two while the halos stay below P3_DS_THREE slots.
// Register budget: P3_DS_THREE only pays with 24 waves per CU, i.e. at most 80 VGPRs.
#define SALVA_P3_BOUNDS(DS) __launch_bounds__(TILE_MAX_THREADS, (DS) == P3_DS_THREE ? 6 : 5)
#ifndef SALVA_EVAL_POST
#define SALVA_EVAL_POST false  // (A/B: the uniform evaluate kernels load rho / alpha / model after the pair loop too)
#endif
// (one body, two instantiations: the uniform-mass kernel carries none of the two-mass code — its second pair loop, the extra
// list word — so that worlds with one mass run exactly what they ran before; device_types.h StepCtx::two_mass)
template <uint32_t DS, int TWO>  // TWO: 0 one mass, 1 two masses, 2 three or four (StepCtx::nmass)
__device__ __forceinline__ void k_divergence_p3_body(StepCtx c) {
    const SolveCtl* const rec = solve_record(c);
    if (rec && rec->done) return;  // the solve converged earlier in this batch
    const float4* const win = rec ? solve_w_in(c, rec) : c.w;
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.skipped()) return;
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    // (the loop needs the position, w_i and the two contact counts of the < 20 rule.  alpha_i and the model: before the staging
    // barrier with everything else in the uniform kernel — their latency hides behind the halo copy; AFTER the loop in the two-mass
    // kernel, whose second pair loop needs the registers — SALVA_EVAL_POST)
    constexpr bool POST = TWO || SALVA_EVAL_POST;
    struct Own { float px, py, pz, ux, uy, uz, alpha; uint32_t mi, cnt, cntb, nb2, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        const float4 p = c.posm[i], u = win[i];
        return Own{p.x, p.y, p.z, u.x, u.y, u.z, POST ? 0.0f : c.alpha[i], POST ? 0u : c.model[i], c.nff[i], c.nb ? c.nfb[i] : 0u,
                   (TWO && t.massb != 0.0f) ? c.nffb[i] : 0u, c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p3_dist8<DS>(t);
    t.stage_p3(c, static_cast<const float4*>(c.posm), win, dist8);
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
            mi = POST ? c.model[i] : o.mi;
            const float rho0 = rho0_of(c, mi);
            float div = 0.0f;
            if (o.cnt + o.cntb >= c.min_neighbors_for_divergence) {
                const float4 pi = make_float4(o.px, o.py, o.pz, 0.0f), wi = make_float4(o.ux, o.uy, o.uz, 0.0f);
                if ((TWO && t.massb != 0.0f) && !near) {  // two-mass world, this tile holds both: the mass inside the loop, by list position
                    div += pair_sum_velocity_divergence_p3_two(c, gs, nqu, o.lh, pi, wi, dist8, o.cnt - o.nb2, t.mass, t.massb);
                } else {
                    div += near ? pair_sum_velocity_divergence_exact_p3(c, i, gs, pi, wi, dist8, t.mass)
                                : pair_sum_velocity_divergence_p3(c, gs, nqu, o.lh, pi, wi, dist8, t.mass);
                    if ((TWO && t.massb != 0.0f) && o.nb2)  // (the exact walk of a slice with a near-coincident pair: the heavier share on top)
                        div += (t.massb - t.mass) * pair_tail_velocity_divergence_p3(c, gs, o.cnt - o.nb2, o.cnt, pi, wi, dist8);
                }
                if (TWO == 2 && t.massc != 0.0f) {  // three or four masses: the third and fourth segment were just counted with the second one's
                    const uint32_t cd = c.nffc[i], l2 = cd & 0xffffu, l3 = cd >> 16;
                    if (l2) div += (t.massc - t.massb) * pair_tail_velocity_divergence_p3(c, gs, o.cnt - l3 - l2, o.cnt - l3, pi, wi, dist8);
                    if (l3) div += (t.massd - t.massb) * pair_tail_velocity_divergence_p3(c, gs, o.cnt - l3, o.cnt, pi, wi, dist8);
                }
                for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                    const float4 pj = Bp[s];
                    const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                    div += (wi.x * dx + wi.y * dy + wi.z * dz) * g * (pj.w * rho0);  // boundary velocity ignored (:332-333)
                });
                div = fmaxf(div, 0.0f);
            }
            c.kappa[i] = div * (POST ? c.alpha[i] : o.alpha);
            err = div / rho0;
        }
        E.add(c, err, mi, active && !is_ghost(c, i));
    });
    E.finish(c, t.slot);
}
template <uint32_t DS>
__global__ SALVA_P3_BOUNDS(DS) void k_divergence_p3(StepCtx c) { k_divergence_p3_body<DS, 0>(c); }
template <uint32_t DS>
__global__ SALVA_P3_BOUNDS(DS) void k_divergence_p3_two(StepCtx c) { k_divergence_p3_body<DS, 1>(c); }
template <uint32_t DS>
__global__ SALVA_P3_BOUNDS(DS) void k_divergence_p3_multi(StepCtx c) { k_divergence_p3_body<DS, 2>(c); }
void launch_divergence(const StepCtx& c, const TileLds& L, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_divergence, c, L, s);
    if (plane_layouts(c)) {
        const uint32_t ds = pick_ds_p3(L.max_halo_fluid, L.ds_level);
        if (c.two_mass && c.nmass > 2u) SALVA_LAUNCH_P3(k_divergence_p3_multi, ds, c, L, p3_bytes(L, ds, c.nmodels, false), s, c);
        else if (c.two_mass) SALVA_LAUNCH_P3(k_divergence_p3_two, ds, c, L, p3_bytes(L, ds, c.nmodels, false), s, c);
        else SALVA_LAUNCH_P3(k_divergence_p3, ds, c, L, p3_bytes(L, ds, c.nmodels, false), s, c);
        return;
    }
    const uint32_t ds = pick_ds(pw_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_divergence, ds, c, L, pw_bytes(L, ds, true), s, c);
}

// ------------------------------------------------------------------------------------------------
// compute_velocity_changes_for_divergence (:358-409): dv_i += sum_j grad W_ij (-(k_i + k_j) m_j)
//                                                           + sum_b grad W_ib (-k_i V_b rho0), boundary reaction force.
// Applied to w_i = v_i + dv_i directly (see the kernel).
// ------------------------------------------------------------------------------------------------
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_divergence_apply(StepCtx c, float inv_dt_prev) {
    const SolveCtl* const rec = solve_record(c);
    const bool was_done = rec && rec->done;
    const float4* const win = rec ? solve_w_in(c, rec) : c.w;
    float4* const wout = c.spec_k >= 0 ? ((win == c.w) ? c.w2 : c.w) : c.w;  // (speculative: the other buffer; else in place)
    lds_base_check();
    if (c.spec_k >= 0 && !c.spec_external && blockIdx.x == 0 && c.slot_base == 0u) {  // the convergence test of this iteration rides in workgroup 0 of the pass's (first) launch (spec_decide)
        spec_decide(c, reinterpret_cast<float*>(tile_smem));
        __syncthreads();
    }
    if (was_done) return;  // the solve converged earlier in this batch
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    // Only w = v + dv is carried through the divergence solve: dv itself is zeroed right after it (:689-691) and v
    // becomes w (:422-430), so updating w in place saves two 16-byte loads and one store per particle and pass.
    struct Own { float4 pi, wi; float ki; uint32_t cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], win[i], c.kappa[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pk_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pk(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.kappa), dist, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi;
        const uint32_t mi = __float_as_uint(o.wi.w);
        const float rho0 = rho0_of(c, mi);
        const float ki = o.ki;
        float4 d = o.wi;
        float sx, sy, sz;
        if (near) pair_sum_gradient_exact(c, i, gs, pi, dist, [&](float kj) { return ki + kj; }, sx, sy, sz);
        else pair_sum_gradient(c, gs, nqu, o.lh, pi, dist, [&](float ka, float kb) { return f2{ki + ka, ki + kb}; }, sx, sy, sz);
        d.x -= sx; d.y -= sy; d.z -= sz;
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = Bp[s];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            const float coeff = -ki * pj.w * rho0 * g;
            const float ex = dx * coeff, ey = dy * coeff, ez = dz * coeff;
            d.x += ex; d.y += ey; d.z += ez;
            if (c.bforce && !is_ghost(c, i)) {
                const float fs = -inv_dt_prev * pi.w;  // delta * (-inv_dt * particle_mass) :404-406
                apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(Bv[s].w), ex * fs, ey * fs, ez * fs);
            }
        });
        wout[i] = d;
    });
}
// the plane-layout form (tile.h stage_p2): every particle has the mass c.mass_uniform
#ifndef SALVA_P2_WAVES
#define SALVA_P2_WAVES 6
#endif
#define SALVA_P2_BOUNDS(DS) __launch_bounds__(TILE_MAX_THREADS, (DS) == P2_DS_THREE ? SALVA_P2_WAVES : 5)
// (one body, two instantiations: the uniform-mass kernel carries none of the two-mass code — its second pair loop, the extra
// list word — so that worlds with one mass run exactly what they ran before; device_types.h StepCtx::two_mass)
template <uint32_t DS, int TWO>  // TWO: 0 one mass, 1 two masses, 2 three or four (StepCtx::nmass)
__device__ __forceinline__ void k_divergence_apply_p2_body(StepCtx c, float inv_dt_prev) {
    const SolveCtl* const rec = solve_record(c);
    const bool was_done = rec && rec->done;
    const float4* const win = rec ? solve_w_in(c, rec) : c.w;
    float4* const wout = c.spec_k >= 0 ? ((win == c.w) ? c.w2 : c.w) : c.w;
    lds_base_check();
    if (c.spec_k >= 0 && !c.spec_external && blockIdx.x == 0 && c.slot_base == 0u) {
        spec_decide(c, reinterpret_cast<float*>(tile_smem));
        __syncthreads();
    }
    if (was_done) return;
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    // (the loop needs the position and kappa_i; w_i, the mass and the model are loaded after it: no register carries them across)
    struct Own { float px, py, pz, ki; uint32_t cnt, nb2, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        const float4 p = c.posm[i];
        return Own{p.x, p.y, p.z, c.kappa[i], c.nff[i], (TWO && t.massb != 0.0f) ? c.nffb[i] : 0u, c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p2_dist8<DS>(t);
    t.stage_p2(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.kappa), dist8);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = make_float4(o.px, o.py, o.pz, 0.0f);
        const float ki = o.ki;
        float sx, sy, sz;
        if ((TWO && t.massb != 0.0f) && !near) {  // two-mass world, this tile holds both: the mass inside the loop, by list position
            pair_sum_gradient_p2_two(c, gs, nqu, o.lh, pi, dist8, o.cnt - o.nb2, t.mass, t.massb, [&](float ka, float kb) { return f2{ki + ka, ki + kb}; }, sx, sy, sz);
        } else {
            if (near) pair_sum_gradient_exact_p2(c, i, gs, pi, dist8, t.mass, [&](float kj) { return ki + kj; }, sx, sy, sz);
            else pair_sum_gradient_p2(c, gs, nqu, o.lh, pi, dist8, t.mass, [&](float ka, float kb) { return f2{ki + ka, ki + kb}; }, sx, sy, sz);
            if ((TWO && t.massb != 0.0f) && o.nb2) {  // (the exact walk of a slice with a near-coincident pair: the heavier share on top)
                float bx, by, bz;
                pair_tail_gradient_p2(c, gs, o.cnt - o.nb2, o.cnt, pi, dist8, [&](float kj) { return ki + kj; }, bx, by, bz);
                const float dm = t.massb - t.mass;
                sx += dm * bx; sy += dm * by; sz += dm * bz;
            }
        }
        if (TWO == 2 && t.massc != 0.0f) {  // three or four masses: the third and fourth segment were just counted with the second one's
            const uint32_t cd = c.nffc[i], l2 = cd & 0xffffu, l3 = cd >> 16;
            float bx, by, bz;
            if (l2) {
                pair_tail_gradient_p2(c, gs, o.cnt - l3 - l2, o.cnt - l3, pi, dist8, [&](float kj) { return ki + kj; }, bx, by, bz);
                const float dm = t.massc - t.massb;
                sx += dm * bx; sy += dm * by; sz += dm * bz;
            }
            if (l3) {
                pair_tail_gradient_p2(c, gs, o.cnt - l3, o.cnt, pi, dist8, [&](float kj) { return ki + kj; }, bx, by, bz);
                const float dm = t.massd - t.massb;
                sx += dm * bx; sy += dm * by; sz += dm * bz;
            }
        }
        float4 d = win[i];
        const uint32_t mi = __float_as_uint(d.w);
        const float rho0 = rho0_of(c, mi);
        d.x -= sx; d.y -= sy; d.z -= sz;
        for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
            const float4 pj = p2_boundary_pos(t, s, dist8);
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            const float coeff = -ki * pj.w * rho0 * g;
            const float ex = dx * coeff, ey = dy * coeff, ez = dz * coeff;
            d.x += ex; d.y += ey; d.z += ez;
            if (c.bforce && !is_ghost(c, i)) {
                const float fs = -inv_dt_prev * c.posm[i].w;
                const uint32_t jb = boundary_sorted_of_slot(c, t, s);
                apply_boundary_force(c, jb, __float_as_uint(c.bvel[jb].w), ex * fs, ey * fs, ez * fs);
            }
        });
        wout[i] = d;
    });
}
template <uint32_t DS>
__global__ SALVA_P2_BOUNDS(DS) void k_divergence_apply_p2(StepCtx c, float inv_dt_prev) { k_divergence_apply_p2_body<DS, 0>(c, inv_dt_prev); }
template <uint32_t DS>
__global__ SALVA_P2_BOUNDS(DS) void k_divergence_apply_p2_two(StepCtx c, float inv_dt_prev) { k_divergence_apply_p2_body<DS, 1>(c, inv_dt_prev); }
template <uint32_t DS>
__global__ SALVA_P2_BOUNDS(DS) void k_divergence_apply_p2_multi(StepCtx c, float inv_dt_prev) { k_divergence_apply_p2_body<DS, 2>(c, inv_dt_prev); }
void launch_divergence_apply(const StepCtx& c, const TileLds& L, float inv_dt_prev, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_divergence_apply, c, L, inv_dt_prev, s);
    if (plane_layouts(c)) {
        const uint32_t ds = pick_ds_p2(L.raw_slots(), L.ds_level);
        if (c.two_mass && c.nmass > 2u) SALVA_LAUNCH_P2(k_divergence_apply_p2_multi, ds, c, L, p2_bytes(L, ds), s, c, inv_dt_prev);
        else if (c.two_mass) SALVA_LAUNCH_P2(k_divergence_apply_p2_two, ds, c, L, p2_bytes(L, ds), s, c, inv_dt_prev);
        else SALVA_LAUNCH_P2(k_divergence_apply_p2, ds, c, L, p2_bytes(L, ds), s, c, inv_dt_prev);
        return;
    }
    const uint32_t ds = pick_ds(pk_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_divergence_apply, ds, c, L, pk_bytes(L, ds), s, c, inv_dt_prev);
}

// ------------------------------------------------------------------------------------------------
// update_velocities (:422-430) + zero velocity changes (:689-691) + `acceleration += gravity` (:574-578).
// v_new = v + dv is exactly the w written by the last apply pass.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_finish_divergence(StepCtx c, float gx, float gy, float gz, int acc_has_user) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n || gate_closed(c)) return;
    const float4 wi = c.w[i];
    const float4 v = c.vel[i];
    const float4 d = c.dv[i];
    c.vel[i] = make_float4(wi.x, wi.y, wi.z, v.w);
    c.dv[i] = make_float4(0.0f, 0.0f, 0.0f, d.w);
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (acc_has_user) a = c.acc[i];
    c.acc[i] = make_float4(a.x + gx, a.y + gy, a.z + gz, 0.0f);
}
void launch_finish_divergence(const StepCtx& c, float gx, float gy, float gz, bool acc_has_user, hipStream_t s) {
    if (c.n) k_finish_divergence<<<num_blocks(c.n), BLOCK, 0, s>>>(c, gx, gy, gz, acc_has_user ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// integrate_and_clear_accelerations (:505-519): dv += a dt ; a = 0.  Refreshes w = v + dv.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_integrate(StepCtx c, float dt) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n || gate_closed(c)) return;
    const float4 a = c.acc[i];
    float4 d = c.dv[i];
    d.x += a.x * dt; d.y += a.y * dt; d.z += a.z * dt;
    c.dv[i] = d;
    c.acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 v = c.vel[i];
    c.w[i] = make_float4(v.x + d.x, v.y + d.y, v.z + d.z, __uint_as_float(c.model[i]));
}
void launch_integrate(const StepCtx& c, float dt, hipStream_t s) {
    if (c.n) k_integrate<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt);
}

// ------------------------------------------------------------------------------------------------
// compute_predicted_densities (:98-162): rho*_i = rho_i + dt (sum_j m_j (w_i - w_j).grad W_ij
//                                                 + sum_b V_b rho0 (w_i - v_b).grad W_ib)
// error_i = max(rho*_i / rho0 - 1, 0); stores kappa_i = (rho*_i - rho0) alpha_i (:234,:245).
// This is THE representative neighbour-sum kernel of the roofline: N (4K + 52) bytes per launch.
// ------------------------------------------------------------------------------------------------
#ifdef SALVA_HIP_DIAG
#define SALVA_DIAG_STAMP(name) const unsigned long long name = __builtin_readcyclecounter()
#else
#define SALVA_DIAG_STAMP(name)
#endif
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_pred_density(StepCtx c, float dt) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    lds_base_check();
    SALVA_DIAG_STAMP(T0);
    Tile t;
    t.setup(c);
    if (t.skipped()) return;  // the other launch of this pass handles the tile (decomposed runs)
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    SALVA_DIAG_STAMP(T1);
    struct Own { float4 pi, wi; float rho, alpha; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.w[i], c.rho[i], c.alpha[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    // two separate 16-byte-strided arrays: a random 64-lane ds_read_b128 then spreads over all 16 bank groups
    // (an interleaved 32-byte record would confine each read to 8 of them)
    const uint32_t dist = pw_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pw(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.w), dist, Bp, Bv, true);
    TileErr E;
    E.init(carve_errtab(t), c);
    SALVA_DIAG_STAMP(T2);
    Tile::staged_barrier();
    SALVA_DIAG_STAMP(T3);
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        float err = 0.0f;
        uint32_t mi = 0;
        if (active) {
            mi = o.mi;
            const float rho0 = rho0_of(c, mi);
            const float4 pi = o.pi, wi = o.wi;
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
            if (!(rs != 0.0f)) atomicOr(c.flags, 1u);  // assert!(!predicted_density.is_zero()) :145 (also catches NaN)
            err = (rs < rho0) ? 0.0f : rs / rho0 - 1.0f;
            c.kappa[i] = (rs - rho0) * o.alpha;
        }
        E.add(c, err, mi, active && !is_ghost(c, i));  // a ghost's error belongs to the rank that owns it
    });
    SALVA_DIAG_STAMP(T4);
    E.finish(c, t.slot);
#ifdef SALVA_HIP_DIAG
    if (c.dbg && threadIdx.x == 0) {
        unsigned long long* d = c.dbg + (size_t)t.slot * 8;
        d[0] = T0; d[1] = T1; d[2] = T2; d[3] = T3; d[4] = T4; d[5] = __builtin_readcyclecounter(); d[6] = t.S; d[7] = t.own_end - t.own_begin;
    }
#endif
}
// the plane-layout form (see k_divergence_p3)
// (one body, two instantiations: the uniform-mass kernel carries none of the two-mass code — its second pair loop, the extra
// list word — so that worlds with one mass run exactly what they ran before; device_types.h StepCtx::two_mass)
template <uint32_t DS, int TWO>  // TWO: 0 one mass, 1 two masses, 2 three or four (StepCtx::nmass)
__device__ __forceinline__ void k_pred_density_p3_body(StepCtx c, float dt) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.skipped()) return;
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    // (the loop needs the position and w_i; rho_i, alpha_i and the model come before the barrier in the uniform kernel and after the
    // loop in the two-mass kernel: see k_divergence_p3_body)
    constexpr bool POST = TWO || SALVA_EVAL_POST;
    struct Own { float px, py, pz, ux, uy, uz, rho, alpha; uint32_t mi, cnt, nb2, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        const float4 p = c.posm[i], u = c.w[i];
        return Own{p.x, p.y, p.z, u.x, u.y, u.z, POST ? 0.0f : c.rho[i], POST ? 0.0f : c.alpha[i], POST ? 0u : c.model[i], c.nff[i],
                   (TWO && t.massb != 0.0f) ? c.nffb[i] : 0u, c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p3_dist8<DS>(t);
    t.stage_p3(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.w), dist8);
    // (boundaries at rest — the usual tank — contribute w_i alone: their velocities are not staged, which is what keeps a
    // 1770-slot fluid halo plus 330 wall slots at three tiles per CU)
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    if (c.bvel_zero) t.stage_boundary(c, Bp);
    else t.stage_boundary(c, Bp, Bv);
    TileErrC E;
    E.init(reinterpret_cast<float*>(t.pool + t.pool_used), c);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        float err = 0.0f;
        uint32_t mi = 0;
        if (active) {
            const float4 pi = make_float4(o.px, o.py, o.pz, 0.0f), wi = make_float4(o.ux, o.uy, o.uz, 0.0f);
            float delta;
            if ((TWO && t.massb != 0.0f) && !near) {  // two-mass world, this tile holds both: the mass inside the loop, by list position
                delta = pair_sum_velocity_divergence_p3_two(c, gs, nqu, o.lh, pi, wi, dist8, o.cnt - o.nb2, t.mass, t.massb);
            } else {
                delta = near ? pair_sum_velocity_divergence_exact_p3(c, i, gs, pi, wi, dist8, t.mass)
                             : pair_sum_velocity_divergence_p3(c, gs, nqu, o.lh, pi, wi, dist8, t.mass);
                if ((TWO && t.massb != 0.0f) && o.nb2)  // (the exact walk of a slice with a near-coincident pair: the heavier share on top)
                    delta += (t.massb - t.mass) * pair_tail_velocity_divergence_p3(c, gs, o.cnt - o.nb2, o.cnt, pi, wi, dist8);
            }
            if (TWO == 2 && t.massc != 0.0f) {  // three or four masses: the third and fourth segment were just counted with the second one's
                const uint32_t cd = c.nffc[i], l2 = cd & 0xffffu, l3 = cd >> 16;
                if (l2) delta += (t.massc - t.massb) * pair_tail_velocity_divergence_p3(c, gs, o.cnt - l3 - l2, o.cnt - l3, pi, wi, dist8);
                if (l3) delta += (t.massd - t.massb) * pair_tail_velocity_divergence_p3(c, gs, o.cnt - l3, o.cnt, pi, wi, dist8);
            }
            mi = POST ? c.model[i] : o.mi;
            const float rho0 = rho0_of(c, mi);
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float4 vj = c.bvel_zero ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : Bv[s];  // (w - 0 = w exactly)
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                delta += pj.w * rho0 * (((wi.x - vj.x) * dx + (wi.y - vj.y) * dy + (wi.z - vj.z) * dz) * g);
            });
            const float rs = (POST ? c.rho[i] : o.rho) + delta * dt;
            if (!(rs != 0.0f)) atomicOr(c.flags, 1u);  // assert!(!predicted_density.is_zero()) :145 (also catches NaN)
            err = (rs < rho0) ? 0.0f : rs / rho0 - 1.0f;
            c.kappa[i] = (rs - rho0) * (POST ? c.alpha[i] : o.alpha);
        }
        E.add(c, err, mi, active && !is_ghost(c, i));
    });
    E.finish(c, t.slot);
}
template <uint32_t DS>
__global__ SALVA_P3_BOUNDS(DS) void k_pred_density_p3(StepCtx c, float dt) { k_pred_density_p3_body<DS, 0>(c, dt); }
template <uint32_t DS>
__global__ SALVA_P3_BOUNDS(DS) void k_pred_density_p3_two(StepCtx c, float dt) { k_pred_density_p3_body<DS, 1>(c, dt); }
template <uint32_t DS>
__global__ SALVA_P3_BOUNDS(DS) void k_pred_density_p3_multi(StepCtx c, float dt) { k_pred_density_p3_body<DS, 2>(c, dt); }
void launch_pred_density(const StepCtx& c, const TileLds& L, float dt, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_pred_density, c, L, dt, s);
    if (plane_layouts(c)) {
        const uint32_t ds = pick_ds_p3(L.max_halo_fluid, L.ds_level);
        if (c.two_mass && c.nmass > 2u) SALVA_LAUNCH_P3(k_pred_density_p3_multi, ds, c, L, p3_bytes(L, ds, c.nmodels, !c.bvel_zero), s, c, dt);
        else if (c.two_mass) SALVA_LAUNCH_P3(k_pred_density_p3_two, ds, c, L, p3_bytes(L, ds, c.nmodels, !c.bvel_zero), s, c, dt);
        else SALVA_LAUNCH_P3(k_pred_density_p3, ds, c, L, p3_bytes(L, ds, c.nmodels, !c.bvel_zero), s, c, dt);
        return;
    }
    const uint32_t ds = pick_ds(pw_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_pred_density, ds, c, L, pw_bytes(L, ds, true), s, c, dt);
}

// ------------------------------------------------------------------------------------------------
// compute_velocity_changes (:218-277): k_ij = max(k_i,0) + max(k_j,0); if k_ij > 0: dv_i -= grad W_ij k_ij m_j / dt.
// Boundary term only when k_i > 0, with the reaction force delta * (inv_dt * m_i).
// ------------------------------------------------------------------------------------------------
template <uint32_t DS>
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_pressure_apply(StepCtx c, float inv_dt) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    struct Own { float4 pi, d, v; float ki; uint32_t mi, cnt, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.dv[i], c.vel[i], c.kappa[i], c.model[i], c.nff[i], c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist = pk_dist<DS>(c, t);
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_pk(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.kappa), dist, Bp, Bv);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = o.pi;
        const uint32_t mi = o.mi;
        const float rho0 = rho0_of(c, mi);
        const float ki = o.ki;
        const float kip = fmaxf(ki, 0.0f);
        float4 d = o.d;
        float sx, sy, sz;
        // k_ij == 0 contributes exactly nothing, so no branch is needed
        if (near) pair_sum_gradient_exact(c, i, gs, pi, dist, [&](float kj) { return kip + fmaxf(kj, 0.0f); }, sx, sy, sz);
        else pair_sum_gradient(c, gs, nqu, o.lh, pi, dist, [&](float ka, float kb) { return f2{kip + fmaxf(ka, 0.0f), kip + fmaxf(kb, 0.0f)}; }, sx, sy, sz);
        d.x -= sx * inv_dt; d.y -= sy * inv_dt; d.z -= sz * inv_dt;
        if (ki > 0.0f) {
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = Bp[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                const float coeff = ki * pj.w * rho0 * inv_dt * g;
                const float ex = dx * coeff, ey = dy * coeff, ez = dz * coeff;
                d.x -= ex; d.y -= ey; d.z -= ez;
                if (c.bforce && !is_ghost(c, i)) {
                    const float fs = inv_dt * pi.w;
                    apply_boundary_force(c, boundary_sorted_of_slot(c, t, s), __float_as_uint(Bv[s].w), ex * fs, ey * fs, ez * fs);
                }
            });
        }
        c.dv[i] = d;
        const float4 v = o.v;
        c.w[i] = make_float4(v.x + d.x, v.y + d.y, v.z + d.z, __uint_as_float(mi));
    });
}
// (one body, two instantiations: the uniform-mass kernel carries none of the two-mass code — its second pair loop, the extra
// list word — so that worlds with one mass run exactly what they ran before; device_types.h StepCtx::two_mass)
template <uint32_t DS, int TWO>  // TWO: 0 one mass, 1 two masses, 2 three or four (StepCtx::nmass)
__device__ __forceinline__ void k_pressure_apply_p2_body(StepCtx c, float inv_dt) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    lds_base_check();
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    // (the loop needs the position and kappa_i; dv_i, v_i, the mass and the model are loaded after it)
    struct Own { float px, py, pz, ki; uint32_t cnt, nb2, near; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        const float4 p = c.posm[i];
        return Own{p.x, p.y, p.z, c.kappa[i], c.nff[i], (TWO && t.massb != 0.0f) ? c.nffb[i] : 0u, c.slice_near[gs], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const uint32_t dist8 = p2_dist8<DS>(t);
    t.stage_p2(c, static_cast<const float4*>(c.posm), static_cast<const float*>(c.kappa), dist8);
    Tile::staged_barrier();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        const bool near = slice_is_near(c, o.near);
        if (!active) return;
        const float4 pi = make_float4(o.px, o.py, o.pz, 0.0f);
        const float ki = o.ki;
        const float kip = fmaxf(ki, 0.0f);
        float sx, sy, sz;
        if ((TWO && t.massb != 0.0f) && !near) {  // two-mass world, this tile holds both: the mass inside the loop, by list position
            pair_sum_gradient_p2_two(c, gs, nqu, o.lh, pi, dist8, o.cnt - o.nb2, t.mass, t.massb,
                                     [&](float ka, float kb) { return f2{kip + fmaxf(ka, 0.0f), kip + fmaxf(kb, 0.0f)}; }, sx, sy, sz);
        } else {
            if (near) pair_sum_gradient_exact_p2(c, i, gs, pi, dist8, t.mass, [&](float kj) { return kip + fmaxf(kj, 0.0f); }, sx, sy, sz);
            else pair_sum_gradient_p2(c, gs, nqu, o.lh, pi, dist8, t.mass, [&](float ka, float kb) { return f2{kip + fmaxf(ka, 0.0f), kip + fmaxf(kb, 0.0f)}; }, sx, sy, sz);
            if ((TWO && t.massb != 0.0f) && o.nb2) {  // (the exact walk of a slice with a near-coincident pair: the heavier share on top)
                float bx, by, bz;
                pair_tail_gradient_p2(c, gs, o.cnt - o.nb2, o.cnt, pi, dist8, [&](float kj) { return kip + fmaxf(kj, 0.0f); }, bx, by, bz);
                const float dm = t.massb - t.mass;
                sx += dm * bx; sy += dm * by; sz += dm * bz;
            }
        }
        if (TWO == 2 && t.massc != 0.0f) {  // three or four masses: the third and fourth segment were just counted with the second one's
            const uint32_t cd = c.nffc[i], l2 = cd & 0xffffu, l3 = cd >> 16;
            float bx, by, bz;
            if (l2) {
                pair_tail_gradient_p2(c, gs, o.cnt - l3 - l2, o.cnt - l3, pi, dist8, [&](float kj) { return kip + fmaxf(kj, 0.0f); }, bx, by, bz);
                const float dm = t.massc - t.massb;
                sx += dm * bx; sy += dm * by; sz += dm * bz;
            }
            if (l3) {
                pair_tail_gradient_p2(c, gs, o.cnt - l3, o.cnt, pi, dist8, [&](float kj) { return kip + fmaxf(kj, 0.0f); }, bx, by, bz);
                const float dm = t.massd - t.massb;
                sx += dm * bx; sy += dm * by; sz += dm * bz;
            }
        }
        const uint32_t mi = c.model[i];
        const float rho0 = rho0_of(c, mi);
        float4 d = c.dv[i];
        d.x -= sx * inv_dt; d.y -= sy * inv_dt; d.z -= sz * inv_dt;
        if (ki > 0.0f) {
            for_each_fb(c, t, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
                const float4 pj = p2_boundary_pos(t, s, dist8);
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                const float coeff = ki * pj.w * rho0 * inv_dt * g;
                const float ex = dx * coeff, ey = dy * coeff, ez = dz * coeff;
                d.x -= ex; d.y -= ey; d.z -= ez;
                if (c.bforce && !is_ghost(c, i)) {
                    const float fs = inv_dt * c.posm[i].w;
                    const uint32_t jb = boundary_sorted_of_slot(c, t, s);
                    apply_boundary_force(c, jb, __float_as_uint(c.bvel[jb].w), ex * fs, ey * fs, ez * fs);
                }
            });
        }
        c.dv[i] = d;
        const float4 v = c.vel[i];
        c.w[i] = make_float4(v.x + d.x, v.y + d.y, v.z + d.z, __uint_as_float(mi));
    });
}
template <uint32_t DS>
__global__ SALVA_P2_BOUNDS(DS) void k_pressure_apply_p2(StepCtx c, float inv_dt) { k_pressure_apply_p2_body<DS, 0>(c, inv_dt); }
template <uint32_t DS>
__global__ SALVA_P2_BOUNDS(DS) void k_pressure_apply_p2_two(StepCtx c, float inv_dt) { k_pressure_apply_p2_body<DS, 1>(c, inv_dt); }
template <uint32_t DS>
__global__ SALVA_P2_BOUNDS(DS) void k_pressure_apply_p2_multi(StepCtx c, float inv_dt) { k_pressure_apply_p2_body<DS, 2>(c, inv_dt); }
void launch_pressure_apply(const StepCtx& c, const TileLds& L, float inv_dt, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_pressure_apply, c, L, inv_dt, s);
    if (plane_layouts(c)) {
        const uint32_t ds = pick_ds_p2(L.raw_slots(), L.ds_level);
        if (c.two_mass && c.nmass > 2u) SALVA_LAUNCH_P2(k_pressure_apply_p2_multi, ds, c, L, p2_bytes(L, ds), s, c, inv_dt);
        else if (c.two_mass) SALVA_LAUNCH_P2(k_pressure_apply_p2_two, ds, c, L, p2_bytes(L, ds), s, c, inv_dt);
        else SALVA_LAUNCH_P2(k_pressure_apply_p2, ds, c, L, p2_bytes(L, ds), s, c, inv_dt);
        return;
    }
    const uint32_t ds = pick_ds(pk_slots(L), L.ds_level);
    SALVA_LAUNCH_FIXED(k_pressure_apply, ds, c, L, pk_bytes(L, ds), s, c, inv_dt);
}

// ------------------------------------------------------------------------------------------------
// update_positions (:411-420): x += (v + dv) dt = w dt.  (v is NOT updated here — the velocity lag of the
// reference.)  Also reduces the cell bounding box of the new positions for the next step's grid.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_update_positions(StepCtx c, float dt, int32_t* bbox_partials) {
    __shared__ int red[6 * (BLOCK / WAVE)];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    if (gate_closed(c)) return;  // (wave-uniform; k_bbox_final behind it is gated the same way)
    if (i < c.n) {
        float4 p = c.posm[i];
        const float4 wi = c.w[i];
        p.x += wi.x * dt; p.y += wi.y * dt; p.z += wi.z * dt;
        c.posm[i] = p;
        bool bad = false;
        mn[0] = mx[0] = cell_coord(p.x, c.sc.h, bad);
        mn[1] = mx[1] = cell_coord(p.y, c.sc.h, bad);
        mn[2] = mx[2] = cell_coord(p.z, c.sc.h, bad);
        if (bad) atomicOr(c.flags, 1u);
    }
    block_bbox_store(mn, mx, red, bbox_partials + 6 * blockIdx.x);
}
void launch_update_positions(const StepCtx& c, float dt, int32_t* bbox_partials, int32_t* bbox6, hipStream_t s) {
    if (!c.n) return;
    k_update_positions<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt, bbox_partials);
    if (bbox6) launch_bbox_final(bbox_partials, num_blocks(c.n), bbox6, s, c.gate);  // (nullptr: the end-of-step publication folds them)
}

// ------------------------------------------------------------------------------------------------
// err = max over fluids of (sum of per-particle errors / nparticles)  (:153-158, :347-352).  One block,
// fixed summation order => run-to-run deterministic iteration counts.
// ------------------------------------------------------------------------------------------------
// Publishes the control block to host-mapped memory after every test (`pub`, may be null): the host then learns the
// outcome of a batch while the batch's last apply pass is still running, instead of after a copy + an idle round trip.
// One 16-byte store of {done, iters, err, seq}: the four words reach the host together (one naturally aligned write to
// fine-grained host memory), so the host, which spins on `seq` and then reads the others, never sees a newer `seq` with older
// data — no system-scope fence (which writes the XCD's L2 back: a third of this one-workgroup kernel's time) is needed,
// because nothing else the host reads was produced by this kernel.
__device__ __forceinline__ void publish_ctl(const SolveCtl* ctl, SolveCtl* pub, uint32_t seq) {
    if (!pub) return;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {ctl->done, ctl->iters, __float_as_uint(ctl->err), seq};
    *reinterpret_cast<volatile u32x4*>(pub) = v;  // (global_store_dwordx4)
}
// (1024 threads: ~2 200 partials at 10^6 particles are two or three independent loads per thread instead of nine dependent
// trips of a 256-thread loop; the control block is loaded whole up front — two 16-byte loads in flight with the partials —
// instead of field by field by one thread at the end.  5.3 -> ~3.5 us per test, of which a settled step runs a hundred.)
constexpr int FINALIZE_THREADS = 1024;
__global__ __launch_bounds__(FINALIZE_THREADS) void k_finalize_error(const float* __restrict__ partials, unsigned nblocks,
                                                                     uint32_t nmodels, const uint32_t* __restrict__ model_counts,
                                                                     SolveCtl* ctl, SolveCtl* pub, const uint32_t* gate, uint32_t* close,
                                                                     uint32_t close_stage, uint32_t skipped) {
    __shared__ float red[FINALIZE_THREADS / WAVE];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // chained step, the chain broke upstream (device_types.h StepCtx::gate): this test does not exist
    if (gate && gate_words_closed(gate[0], gate[1], close_stage)) return;
    const u32x4 c0 = reinterpret_cast<const u32x4*>(ctl)[0], c1 = reinterpret_cast<const u32x4*>(ctl)[1];
    // (skipped: the tests of earlier iterations that were not launched because they could not end the solve — i < min_iter, World::run_solve
    // — count as failed ones: one apply / iteration and one tick each)
    const uint32_t done = c0.x, iters = c0.y + skipped, seq = c0.w + 1u + skipped, min_iter = c1.y, mode = c1.z;
    const float tol = __uint_as_float(c1.x);
    if (done) {
        if (threadIdx.x == 0) { ctl->seq = seq; publish_ctl(ctl, pub, seq); }
        return;
    }
    float best = 0.0f;
    for (uint32_t m = 0; m < nmodels; ++m) {
        const uint32_t cnt = model_counts[m];
        float s = 0.0f;
        for (unsigned b = threadIdx.x; b < nblocks; b += FINALIZE_THREADS) s += partials[(size_t)b * nmodels + m];
        s = block_sum(s, red);
        if (threadIdx.x == 0 && cnt != 0) best = fmaxf(best, s / (float)cnt);
    }
    if (threadIdx.x == 0) {
        const bool ok = best <= tol && iters >= min_iter;
        // mode 0: DFSPH protocol (test, then count the apply that follows); mode 1: IISPH (count the iteration, then test)
        const u32x4 out = {ok ? 1u : 0u, (mode == 0 && ok) ? iters : iters + 1u, __float_as_uint(best), seq};
        reinterpret_cast<u32x4*>(ctl)[0] = out;
        if (pub) *reinterpret_cast<volatile u32x4*>(pub) = out;
        // the last test of a chained batch: not converged -> everything enqueued behind this solve returns at once, and the host,
        // which learns it from the end-of-step publication, continues the solve from here (close = &Readback::chain_ok, [1] = stage)
        if (close && !ok) { close[0] = 0u; close[1] = close_stage; }
    }
}
// multi-GPU: the same reduction split around an all-reduce over the ranks
__global__ __launch_bounds__(BLOCK) void k_sum_partials(const float* __restrict__ partials, unsigned nblocks, uint32_t nmodels,
                                                        const SolveCtl* ctl, float* sums) {
    __shared__ float red[BLOCK / WAVE];
    if (ctl->done) {  // keep the value finite; k_decide ignores it
        if (threadIdx.x < nmodels) sums[threadIdx.x] = 0.0f;
        return;
    }
    for (uint32_t m = 0; m < nmodels; ++m) {
        float s = 0.0f;
        for (unsigned b = threadIdx.x; b < nblocks; b += BLOCK) s += partials[(size_t)b * nmodels + m];
        s = block_sum(s, red);
        if (threadIdx.x == 0) sums[m] = s;
    }
}
__global__ void k_decide(const float* __restrict__ sums, uint32_t nmodels, const uint32_t* __restrict__ model_counts, SolveCtl* ctl,
                         SolveCtl* pub, uint32_t skipped) {
    if (threadIdx.x != 0) return;
    if (skipped) { ctl->iters += skipped; ctl->seq += skipped; }  // (tests that were not launched: k_finalize_error)
    if (!ctl->done) {
        float best = 0.0f;
        for (uint32_t m = 0; m < nmodels; ++m)
            if (model_counts[m] != 0) best = fmaxf(best, sums[m] / (float)model_counts[m]);
        ctl->err = best;
        if (ctl->mode == 0) {
            if (best <= ctl->tol && ctl->iters >= ctl->min_iter) ctl->done = 1u;
            else ctl->iters += 1u;
        } else {
            const uint32_t i = ctl->iters;
            ctl->iters = i + 1u;
            if (best <= ctl->tol && i >= ctl->min_iter) ctl->done = 1u;
        }
    }
    const uint32_t s = ctl->seq + 1u;  // one tick per convergence test, converged or not: the host waits for the count it enqueued
    ctl->seq = s;
    publish_ctl(ctl, pub, s);
}
// the same decision for a speculative decomposed solve (World::run_solve, spec_dist): iteration k reads ring[k & 1] and the record of
// iteration k + 1 goes to ring[(k + 1) & 1] — the apply pass of iteration k runs beside this kernel and still reads the old one
__global__ void k_decide_ring(const float* __restrict__ sums, uint32_t nmodels, const uint32_t* __restrict__ model_counts, SolveCtl* ring, int k,
                              SolveCtl* pub) {
    if (threadIdx.x != 0) return;
    const SolveCtl r = ring[k & 1];
    SolveCtl n = r;
    if (!r.done) {
        float best = 0.0f;
        for (uint32_t m = 0; m < nmodels; ++m)
            if (model_counts[m] != 0) best = fmaxf(best, sums[m] / (float)model_counts[m]);
        n.err = best;
        if (best <= r.tol && r.iters >= r.min_iter) n.done = 1u;
        else n.iters = r.iters + 1u;
    }
    n.seq = r.seq + 1u;
    ring[(k + 1) & 1] = n;
    publish_ctl(&n, pub, n.seq);
}
void launch_decide_ring(const float* sums, uint32_t nmodels, const uint32_t* model_counts, SolveCtl* ring, int k, SolveCtl* pub, hipStream_t s) {
    k_decide_ring<<<1, 64, 0, s>>>(sums, nmodels, model_counts, ring, k, pub);
}
void launch_sum_partials(const float* partials, unsigned nblocks, uint32_t nmodels, const SolveCtl* ctl, float* sums, hipStream_t s) {
    k_sum_partials<<<1, BLOCK, 0, s>>>(partials, nblocks, nmodels, ctl, sums);
}
void launch_decide(const float* sums, uint32_t nmodels, const uint32_t* model_counts, SolveCtl* ctl, SolveCtl* pub, hipStream_t s, uint32_t skipped) {
    k_decide<<<1, 64, 0, s>>>(sums, nmodels, model_counts, ctl, pub, skipped);
}
void launch_finalize_error(const float* partials, unsigned nblocks, uint32_t nmodels, const uint32_t* model_counts,
                           SolveCtl* ctl, SolveCtl* pub, hipStream_t s, const uint32_t* gate, uint32_t* close, uint32_t close_stage,
                           uint32_t skipped) {
    k_finalize_error<<<1, FINALIZE_THREADS, 0, s>>>(partials, nblocks, nmodels, model_counts, ctl, pub, gate, close, close_stage, skipped);
}

// ------------------------------------------------------------------------------------------------ boundary volumes
// dfsph_solver.rs:72-96: V_b = 1 / sum over boundary-boundary contacts of W (same boundary always, other
// boundaries when their interaction groups allow, contacts.rs:261-296).  No list is kept: the sum is evaluated
// straight from the boundary cell table, once per change of the boundary set.
__global__ __launch_bounds__(BLOCK) void k_boundary_volumes(StepCtx c, unsigned long long* ncontacts_bb) {
    __shared__ float red[BLOCK / WAVE];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t cnt = 0;
    if (i < c.nb) {
        const float4 pi = c.bposv[i];
        const uint32_t mi = __float_as_uint(c.bvel[i].w);
        bool bad = false;
        const int cx = cell_coord(pi.x, c.sc.h, bad), cy = cell_coord(pi.y, c.sc.h, bad), cz = cell_coord(pi.z, c.sc.h, bad);
        float denom = 0.0f;
        for (int d = 0; d < 27; ++d) {
            bool in;
            const uint32_t k = tile_key(c.gb, cx + d / 9 - 1, cy + (d / 3) % 3 - 1, cz + d % 3 - 1, in);
            if (!in) continue;
            const uint32_t b = c.gb.cell_start[k], e = c.gb.cell_start[k + 1];
            for (uint32_t j = b; j < e; ++j) {
                const float4 pj = c.bposv[j];
                const float d2 = dist2_exact(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
                if (d2 <= c.sc.h2) {
                    const uint32_t mj = __float_as_uint(c.bvel[j].w);
                    if (mi == mj || c.bb_ok[mi * c.nbmodels + mj]) {
                        denom += kernel_weight(d2, c.sc);
                        ++cnt;
                    }
                }
            }
        }
        if (!(denom > 0.0f)) atomicOr(c.flags, 1u);  // assert!(!denominator.is_zero()) dfsph_solver.rs:92
        // a decomposed run replicates the boundary particles near a slab face on both ranks: each contact is reported by the rank
        // whose slab holds its first particle, so that the ranks' counts add up to the undivided domain's
        if (!(cx > c.ghost_lo_cx && cx < c.ghost_hi_cx)) cnt = 0;
        reinterpret_cast<float*>(&c.bposv[i])[3] = 1.0f / denom;
    }
    const float tot = block_sum((float)cnt, red);
    if (threadIdx.x == 0 && tot > 0.0f) atomicAdd(ncontacts_bb, (unsigned long long)tot);
}
void launch_boundary_volumes(const StepCtx& c, unsigned long long* ncontacts_bb, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_boundary_volumes, c, ncontacts_bb, s);
    if (c.nb == 0) return;
    k_boundary_volumes<<<div_up(c.nb, BLOCK), BLOCK, 0, s>>>(c, ncontacts_bb);
}

}  // namespace SALVA_KNS
