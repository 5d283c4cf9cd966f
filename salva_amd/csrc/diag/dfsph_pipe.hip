// dfsph_pipe.hip — the DFSPH solver passes on the persistent tile pipeline (pipe.h).
//
// Same arithmetic, in the same order, as the one-tile-per-workgroup kernels of dfsph.hip (behaviour specified by
// /root/reference/src/solver/pressure/dfsph_solver.rs, lines cited there): results are bit-identical between the two
// skeletons (salva_hip_time_variant returns a checksum of the outputs; tools/variant_probe.py compares them).
// Kernel-development build only (`make VARIANT=diag`): rejected as a production skeleton in round 2.
#include "kernels.h"
#include "pipe.h"
#include "../pairs.h"

namespace salva {

// ------------------------------------------------------------------------------------------------
// compute_predicted_densities (dfsph_solver.rs:98-162), cf. k_pred_density in dfsph.hip
// ------------------------------------------------------------------------------------------------
template <bool FB> struct OwnPD;
template <> struct OwnPD<true> { float4 pi, wi; float rho, alpha; uint32_t cnt, cntb; ListRegs lh; FbRegs fb; };
template <> struct OwnPD<false> { float4 pi, wi; float rho, alpha; uint32_t cnt, cntb; ListRegs lh; };

// MAXW: waves per workgroup the register budget is cut for; DOUBLE: one workgroup per CU with two halo buffers, or (false)
// single-buffered workgroups, as many per CU as fit (register budget: four waves per SIMD)
template <int MAXW, bool DOUBLE>
__global__ __launch_bounds__(MAXW * WAVE) __attribute__((amdgpu_waves_per_eu(DOUBLE ? 1 : 4)))
void k_pred_density_pipe(StepCtx c, float dt, uint32_t scap, uint32_t sbcap) {
    if (c.ctl && c.ctl->done) return;  // the solve converged earlier in this batch
    using Own = OwnPD<DOUBLE>;
    auto load_own = [&](uint32_t i, uint32_t gs) {
        if constexpr (DOUBLE) return Own{c.posm[i], c.w[i], c.rho[i], c.alpha[i], c.nff[i], c.nb ? c.nfb[i] : 0u, list_regs(c, gs), fb_regs(c, gs)};
        else return Own{c.posm[i], c.w[i], c.rho[i], c.alpha[i], c.nff[i], c.nb ? c.nfb[i] : 0u, list_regs(c, gs)};
    };
    tile_pipeline<2, 0, 2, true, DOUBLE, Own>(
        c, scap, sbcap, c.posm, c.w, nullptr, load_own,
        [&](const Own& o, uint32_t i, uint32_t gs, bool active, const PipeView& t, TileErr& E) {
            const float4* Lp = t.Lp;
            const float4* Lw = t.Lw;
            const uint32_t nqu = slice_list_dwords(o.cnt, active);
            float err = 0.0f;
            uint32_t mi = 0;
            if (active) {
                mi = __float_as_uint(o.wi.w);
                const float rho0 = rho0_of(c, t, mi);
                const float4 pi = o.pi, wi = o.wi;
                float delta = 0.0f;
                // round 3: the pair loop of the product kernel (pairs.h); the single-buffer pipeline stages P at LDS byte 0 and W
                // scap slots behind it (the double-buffered one alternates: runtime base, kept on the round-2 loop)
                if constexpr (!DOUBLE) {
                    delta += pair_sum_velocity_divergence<false>(c, gs, nqu, o.lh, pi, wi, scap * 16u);
                } else {
                f2 acc2 = {0.0f, 0.0f};
                for_each_ff2<false>(c, gs, nqu, o.lh, [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; }, [&](const RecPW& A, const RecPW& B) {
                    asm volatile("" ::"v"(A.w.w), "v"(B.w.w));  // keep the reads single ds_read_b128s
                    const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                    const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                    const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
                    const f2 mj = {A.p.w, B.p.w};
                    acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
                });
                delta += acc2.x + acc2.y;
                }
                auto fb = [&](uint32_t s) {
                    const float4 pj = t.Bp[s];
                    const float4 vj = t.Bv[s];
                    const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                    delta += pj.w * rho0 * (((wi.x - vj.x) * dx + (wi.y - vj.y) * dy + (wi.z - vj.z) * dz) * g);
                };
                if constexpr (DOUBLE) for_each_fb_regs(c, t.SB, gs, o.cntb, o.fb, fb);
                else if (t.SB) for_each_slot(c.nbr_fb, c.cap_fb, gs, o.cntb, [](uint32_t s) { return s; }, fb);
                const float rs = o.rho + delta * dt;
                if (!(rs != 0.0f)) atomicOr(c.flags, 1u);  // assert!(!predicted_density.is_zero()) :145 (also catches NaN)
                err = (rs < rho0) ? 0.0f : rs / rho0 - 1.0f;
                c.kappa[i] = (rs - rho0) * o.alpha;
            }
            E.add(c, err, mi, active && !is_ghost(c, i));
        });
}
void launch_pred_density_pipe(const StepCtx& c, const PipeCfg& P, float dt, hipStream_t s) {
    if (!c.n || !c.nlaunch) return;
    const uint32_t lds = P.bytes(2, 0, 2, true);
    if (P.threads <= 8 * WAVE) {
        ensure_tile_lds(k_pred_density_pipe<8, true>, lds);
        k_pred_density_pipe<8, true><<<P.grid(lds), P.threads, lds, s>>>(c, dt, P.scap, P.sbcap);
    } else {
        ensure_tile_lds(k_pred_density_pipe<12, true>, lds);
        k_pred_density_pipe<12, true><<<P.grid(lds), P.threads, lds, s>>>(c, dt, P.scap, P.sbcap);
    }
}
// single-buffered persistent workgroups (variant 4)
void launch_pred_density_loop(const StepCtx& c, const PipeCfg& P, float dt, hipStream_t s) {
    if (!c.n || !c.nlaunch) return;
    const uint32_t lds = P.bytes(2, 0, 2, true, 1u);
    if (P.threads <= 8 * WAVE) {
        ensure_tile_lds(k_pred_density_pipe<8, false>, lds);
        k_pred_density_pipe<8, false><<<P.grid(lds), P.threads, lds, s>>>(c, dt, P.scap, P.sbcap);
    } else {
        ensure_tile_lds(k_pred_density_pipe<12, false>, lds);
        k_pred_density_pipe<12, false><<<P.grid(lds), P.threads, lds, s>>>(c, dt, P.scap, P.sbcap);
    }
    SALVA_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Experimental variants of the one-tile-per-workgroup k_pred_density (diagnostics: salva_hip_time_variant).
//   VAR 1: the two workgroups that share a CU start half a tile period apart (the second arrival on a CU sleeps),
//          to test whether co-resident tiles run in phase;
//   VAR 3: halo staged by LDS-DMA instead of global_load + ds_write.
// ------------------------------------------------------------------------------------------------
//   VAR 7: VAR 3 with the register budget cut for five waves per SIMD, so that two 9-wave workgroups share a CU (a tile of
//          a jittered 2r lattice holds 512 +- a few particles: nine slices, the ninth almost empty — with eight waves one
//          wave walks two slices while seven wait).
template <int VAR>
__global__ __launch_bounds__(TILE_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(VAR == 7 ? 5 : 3)))
void k_pred_density_x(StepCtx c, float dt, uint32_t* cu_arrivals, uint32_t sleep_units) {
    if (c.ctl && c.ctl->done) return;
    __shared__ float errtab[TILE_MAX_WAVES][MAX_MODELS];
    if (VAR == 1) {
        __shared__ uint32_t order;
        if (threadIdx.x == 0) {
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID [3:0]
            order = atomicAdd(&cu_arrivals[((xcc & 15u) << 8) | ((hw >> 8) & 255u)], 1u);
        }
        __syncthreads();
        if (order & 1u) {
            for (uint32_t k = 0; k < sleep_units; ++k) __builtin_amdgcn_s_sleep(1);  // 64 cycles each
        }
    }
    Tile t;
    if (VAR == 8) t.setup_at(c, cu_arrivals[blockIdx.x]);  // cu_arrivals = dispatch order (heaviest tiles first)
    else t.setup(c);
    if (t.empty()) { TileErr::zero(c, t.slot); return; }
    struct Own { float4 pi, wi; float rho, alpha; uint32_t mi, cnt; ListRegs lh; };
    auto load_own = [&](uint32_t i, uint32_t gs) {
        return Own{c.posm[i], c.w[i], c.rho[i], c.alpha[i], c.model[i], c.nff[i], list_regs(c, gs)};
    };
    uint32_t i0, gs0;
    t.first_own(i0, gs0);
    const Own own0 = load_own(i0, gs0);
    const float4* Lp = nullptr;
    const float4* Lw = nullptr;
    if (VAR >= 3 && c.halo_stride) {
        const uint32_t cap = (t.S + 63u) & ~63u;
        float4* a = t.carve<float4>(cap);
        float4* b = t.carve<float4>(cap);
        const uint32_t nt = blockDim.x, lane = threadIdx.x & 63u, nw = nt / WAVE;
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
        const uint32_t pre[4] = {t.pre0, t.pre1, t.pre2, t.pre3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t s0 = (wv + (uint32_t)k * nw) * WAVE;
            if (s0 < t.S) {
                const uint32_t g = (s0 + lane < t.S) ? pre[k] : t.own_begin;
                glds16(c.posm + g, a + s0);
                glds16(c.w + g, b + s0);
            }
        }
        for (uint32_t s0 = (wv + 4u * nw) * WAVE; s0 < t.S; s0 += nt) {
            const uint32_t g = (s0 + lane < t.S) ? c.halo_src[t.hoff + s0 + lane] : t.own_begin;
            glds16(c.posm + g, a + s0);
            glds16(c.w + g, b + s0);
        }
        Lp = a; Lw = b;
    } else {
        t.stage(c, static_cast<const float4*>(c.posm), static_cast<const float4*>(c.w), Lp, Lw);
    }
    const float4* Bp = nullptr;
    const float4* Bv = nullptr;
    t.stage_boundary(c, Bp, Bv);
    TileErr E;
    E.init(errtab, c);
    if (VAR >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    t.for_own_pre(own0, load_own, [&](const Own& o, uint32_t i, uint32_t gs, bool active) {
        const uint32_t nqu = slice_list_dwords(o.cnt, active);
        float err = 0.0f;
        uint32_t mi = 0;
        if (active) {
            mi = o.mi;
            const float rho0 = rho0_of(c, mi);
            const float4 pi = o.pi, wi = o.wi;
            float delta = 0.0f;
            f2 acc2 = {0.0f, 0.0f};
            if (VAR == 6) {
                f2 acc2b = {0.0f, 0.0f};
                auto c2 = [&](const RecPW& A, const RecPW& B) {
                    asm volatile("" ::"v"(A.w.w), "v"(B.w.w));
                    const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                    const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                    const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
                    const f2 mj = {A.p.w, B.p.w};
                    acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
                };
                for_each_ff4(c, gs, nqu, o.lh, [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; },
                             [&](const RecPW& A, const RecPW& B, const RecPW& C, const RecPW& D) {
                    asm volatile("" ::"v"(A.w.w), "v"(B.w.w), "v"(C.w.w), "v"(D.w.w));
                    const f2 dxa = {pi.x - A.p.x, pi.x - B.p.x}, dxb = {pi.x - C.p.x, pi.x - D.p.x};
                    const f2 dya = {pi.y - A.p.y, pi.y - B.p.y}, dyb = {pi.y - C.p.y, pi.y - D.p.y};
                    const f2 dza = {pi.z - A.p.z, pi.z - B.p.z}, dzb = {pi.z - C.p.z, pi.z - D.p.z};
                    f2 ga, gb;
                    kernel_grad2x2(dxa * dxa + dya * dya + dza * dza, dxb * dxb + dyb * dyb + dzb * dzb, c.sc, ga, gb);
                    const f2 uxa = {wi.x - A.w.x, wi.x - B.w.x}, uxb = {wi.x - C.w.x, wi.x - D.w.x};
                    const f2 uya = {wi.y - A.w.y, wi.y - B.w.y}, uyb = {wi.y - C.w.y, wi.y - D.w.y};
                    const f2 uza = {wi.z - A.w.z, wi.z - B.w.z}, uzb = {wi.z - C.w.z, wi.z - D.w.z};
                    const f2 ma = {A.p.w, B.p.w}, mb = {C.p.w, D.p.w};
                    acc2 += (uxa * dxa + uya * dya + uza * dza) * ga * ma;
                    acc2 += (uxb * dxb + uyb * dyb + uzb * dzb) * gb * mb;
                }, c2);
                (void)acc2b;
            } else
            for_each_ff2<true, VAR == 5>(c, gs, nqu, o.lh, [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; }, [&](const RecPW& A, const RecPW& B) {
                asm volatile("" ::"v"(A.w.w), "v"(B.w.w));
                const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
                const f2 mj = {A.p.w, B.p.w};
                acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
            });
            delta += acc2.x + acc2.y;
            for_each_fb(c, t, i, gs, [&](uint32_t s) {
                const float4 pj = Bp[s];
                const float4 vj = Bv[s];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                delta += pj.w * rho0 * (((wi.x - vj.x) * dx + (wi.y - vj.y) * dy + (wi.z - vj.z) * dz) * g);
            });
            const float rs = o.rho + delta * dt;
            if (!(rs != 0.0f)) atomicOr(c.flags, 1u);
            err = (rs < rho0) ? 0.0f : rs / rho0 - 1.0f;
            c.kappa[i] = (rs - rho0) * o.alpha;
        }
        E.add(c, err, mi, active && !is_ghost(c, i));
    });
    E.finish(c, t.slot);
}
void launch_pred_density_variant(const StepCtx& c, const TileLds& L, const PipeCfg& P, float dt, int variant, uint32_t param,
                                 uint32_t* cu_arrivals, hipStream_t s) {
    switch (variant) {
        case 1:
            SALVA_HIP_CHECK(hipMemsetAsync(cu_arrivals, 0, 4096 * sizeof(uint32_t), s));
            SALVA_LAUNCH_TILE(k_pred_density_x<1>, c, L, L.bytes(32, 32, 4), s, c, dt, cu_arrivals, param);
            break;
        case 2: launch_pred_density_pipe(c, P, dt, s); break;
        case 4: launch_pred_density_loop(c, P, dt, s); break;
        case 3: SALVA_LAUNCH_TILE(k_pred_density_x<3>, c, L, L.bytes(32, 32, 4) + 4096u, s, c, dt, cu_arrivals, param); break;
        case 5: SALVA_LAUNCH_TILE(k_pred_density_x<5>, c, L, L.bytes(32, 32, 4) + 4096u, s, c, dt, cu_arrivals, param); break;
        case 8: SALVA_LAUNCH_TILE(k_pred_density_x<8>, c, L, L.bytes(32, 32, 4) + 4096u, s, c, dt, cu_arrivals, param); break;
        case 7: SALVA_LAUNCH_TILE(k_pred_density_x<7>, c, L, L.bytes(32, 32, 4) + 4096u, s, c, dt, cu_arrivals, param); break;
        case 6: SALVA_LAUNCH_TILE(k_pred_density_x<6>, c, L, L.bytes(32, 32, 4) + 4096u, s, c, dt, cu_arrivals, param); break;
        default: launch_pred_density(c, L, dt, s); break;
    }
}

}  // namespace salva
