// visc.hip — solver::DFSPHViscosity ("viscous DFSPH", /root/reference/src/solver/viscosity/dfsph_viscosity.rs) as tile
// kernels.  SURVEY.md §8 row f1.  The force belongs to one Fluid and acts between particles of that fluid only
// (`c.i_model == c.j_model`, :150,:222,:271); it has no boundary term.
//
//   solve (:290-327):   betas;  target = strain_rate * (1 - coefficient);
//                       for i < max_iter { err = mean |strain_rate - target|_1 / 6; if err <= max_err && i >= min_iter break;
//                                          accelerations += ... }
//
// Per-particle state (betas 6x6, strain-rate target, u = beta * error / rho^2) is rebuilt by every solve, so it lives in
// per-step scratch in sorted order; betas and targets are SoA ([component][particle]) for coalesced streaming.
// The 6x3 "gradient matrix" M(g) (:60-83) is linear in the kernel gradient g, so the neighbour pass of compute_betas
// only accumulates sum(s g) (3 values) and sum(s^2 g_a g_b) (6 values), s = m_j / (2 rho_i); M M^T is assembled once.
#include "kernels.h"
#include "tile.h"

namespace SALVA_KNS {
using namespace salva;

// rows of M(g): (2gx,0,0) (0,2gy,0) (0,0,2gz) (gy,gx,0) (gz,0,gx) (0,gz,gy);  entry (a,b) of M(g) M(g)^T from the six
// products q = {xx, yy, zz, xy, xz, yz}
__device__ __forceinline__ void mmt_from_products(const float q[6], float out[6][6]) {
    const float xx = q[0], yy = q[1], zz = q[2], xy = q[3], xz = q[4], yz = q[5];
    out[0][0] = 4.0f * xx; out[0][1] = 0.0f;      out[0][2] = 0.0f;      out[0][3] = 2.0f * xy; out[0][4] = 2.0f * xz; out[0][5] = 0.0f;
    out[1][1] = 4.0f * yy; out[1][2] = 0.0f;      out[1][3] = 2.0f * xy; out[1][4] = 0.0f;      out[1][5] = 2.0f * yz;
    out[2][2] = 4.0f * zz; out[2][3] = 0.0f;      out[2][4] = 2.0f * xz; out[2][5] = 2.0f * yz;
    out[3][3] = yy + xx;   out[3][4] = yz;        out[3][5] = xz;
    out[4][4] = zz + xx;   out[4][5] = xy;
    out[5][5] = zz + yy;
#pragma unroll
    for (int a = 1; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < a; ++b) out[a][b] = out[b][a];
}

// nalgebra 0.33 Matrix6::lu() + determinant() + try_inverse() (linalg/lu.rs, linalg/solve.rs — an un-vendored dependency,
// restated from its published algorithm): partial pivoting on the first largest |.| of the column, multipliers =
// entry * (1 / pivot), updates y = (-p) l + y; column-wise forward substitution with a unit diagonal, then back
// substitution dividing by the diagonal.  No FMA contraction, so that the CPU restatement used by the tests rounds alike.  Fully unrolled: every array index is a compile-time constant and
// row exchanges are conditional moves, so the matrix stays in registers.
__device__ __forceinline__ bool lu_inverse6(float a[6][6], float inv[6][6], float& det) {
    int perm[6] = {0, 1, 2, 3, 4, 5};
    bool odd = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int piv = i;
        float best = fabsf(a[i][i]);
#pragma unroll
        for (int r = i + 1; r < 6; ++r) {
            const float v = fabsf(a[r][i]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0f) continue;  // no non-zero entry in this column
#pragma unroll
        for (int r = i + 1; r < 6; ++r) {
            const bool sw = (piv == r);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float x = a[i][j], y = a[r][j];
                a[i][j] = sw ? y : x;
                a[r][j] = sw ? x : y;
            }
            const int pi = perm[i], pr = perm[r];
            perm[i] = sw ? pr : pi;
            perm[r] = sw ? pi : pr;
            odd = sw ? !odd : odd;
        }
        const float inv_diag = __fdiv_rn(1.0f, a[i][i]);
#pragma unroll
        for (int r = i + 1; r < 6; ++r) a[r][i] = __fmul_rn(a[r][i], inv_diag);
#pragma unroll
        for (int k = i + 1; k < 6; ++k) {
            const float pk = a[i][k];
#pragma unroll
            for (int r = i + 1; r < 6; ++r) a[r][k] = __fadd_rn(__fmul_rn(-pk, a[r][i]), a[r][k]);
        }
    }
    det = 1.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) det = __fmul_rn(det, a[i][i]);
    if (odd) det = -det;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float b[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) b[r] = (perm[r] == c) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float coeff = b[i];
#pragma unroll
            for (int r = i + 1; r < 6; ++r) b[r] = __fadd_rn(__fmul_rn(-coeff, a[r][i]), b[r]);
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            const float d = a[i][i];
            if (d == 0.0f) ok = false;
            const float coeff = __fdiv_rn(b[i], d);
            b[i] = coeff;
#pragma unroll
            for (int r = 0; r < i; ++r) b[r] = __fadd_rn(__fmul_rn(-coeff, a[r][i]), b[r]);
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) inv[r][c] = b[r];
    }
    return ok;
}

// ------------------------------------------------------------------------------------------------ compute_betas (:130-194)
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_visc_betas(StepCtx c, uint32_t model, float* __restrict__ beta) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const uint32_t* Lm = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), static_cast<const uint32_t*>(c.model), Lp, Lm);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        const float rho = c.rho[i];
        const float half_inv_rho = 1.0f / (2.0f * rho);
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        float q[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        struct Rec { float4 p; uint32_t m; };
        for_each_ff_regs(c, gs, lo, [&](uint32_t s) { return Rec{Lp[s], Lm[s]}; }, [&](const Rec& rc) { SALVA_PAIR_MATH
            const float4 pj = rc.p;
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            const float sc = (rc.m == model) ? g * (pj.w * half_inv_rho) : 0.0f;  // particle_mass(j) / (2 rho_i), times |grad|/r
            const float gx = dx * sc, gy = dy * sc, gz = dz * sc;
            sx += gx; sy += gy; sz += gz;
            q[0] += gx * gx; q[1] += gy * gy; q[2] += gz * gz; q[3] += gx * gy; q[4] += gx * gz; q[5] += gy * gz;
        });
        // denominator = sum(grad_i grad_i^T) / rho_i + grad_sum grad_sum^T / rho_i
        float sq[6][6], gg[6][6], den[6][6];
        mmt_from_products(q, sq);
        const float q2[6] = {sx * sx, sy * sy, sz * sz, sx * sy, sx * sz, sy * sz};
        mmt_from_products(q2, gg);
        const float inv_rho = 1.0f / rho;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) den[a][b] = sq[a][b] * inv_rho + gg[a][b] * inv_rho;
        // preconditioner (:163-175): rows of the first SPATIAL_DIM = 3 columns are scaled by 1 / diagonal
        float inv_diag[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) inv_diag[a] = (fabsf(den[a][a]) < 1.0e-6f) ? 1.0f : 1.0f / den[a][a];
#pragma unroll
        for (int col = 0; col < 3; ++col)
#pragma unroll
            for (int r = 0; r < 6; ++r) den[r][col] *= inv_diag[r];
        float inv[6][6], det = 0.0f;
        const bool ok = lu_inverse6(den, inv, det);
        const bool zero = !(fabsf(det) >= 1.0e-6f) || !ok;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                float v = zero ? 0.0f : inv[a][b];
                if (b < 3) v *= inv_diag[b];  // (:189-192) columns 0..2 only
                beta[(size_t)(a * 6 + b) * c.n + i] = v;
            }
    });
}
void launch_visc_betas(const StepCtx& c, const TileLds& L, uint32_t model, float* beta, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_visc_betas, c, L, model, beta, s);
    SALVA_LAUNCH_TILE(k_visc_betas, c, L, L.bytes(20, 0, 3), s, c, model, beta);
}

// v_i + a_i dt for the strain-rate passes (:220-221); .w keeps the model id.  dt is timestep.dt() = the previous step's.
__global__ __launch_bounds__(BLOCK) void k_visc_va(StepCtx c, float dt, float4* __restrict__ va) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const float4 v = c.w[i], a = c.acc[i];
    va[i] = make_float4(v.x + a.x * dt, v.y + a.y * dt, v.z + a.z * dt, v.w);
}
void launch_visc_va(const StepCtx& c, float dt_prev, float4* va, hipStream_t s) {
    if (c.n) k_visc_va<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt_prev, va);
}

// ------------------------------------------------------------------------------------------------ compute_strain_rates (:196-247)
// mode 0: target = rate (1 - coefficient).  mode 1: error = rate - target, per-particle |error|_1 / 6 into the tile's
// partial sum, and u = beta error / rho^2 (:262,:272-273) for the acceleration pass, as two float4 (u0 u1 u2 ·)(u3 u4 u5 model).
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_visc_strain(StepCtx c, uint32_t model, int mode, float coef,
                                                                 const float4* __restrict__ va, const float* __restrict__ beta,
                                                                 float* __restrict__ target, float4* __restrict__ u0,
                                                                 float4* __restrict__ u1) {
    if (mode == 1 && c.ctl && c.ctl->done) return;
    __shared__ float errtab[TILE_MAX_WAVES][MAX_MODELS];
    Tile t;
    t.setup(c);
    if (t.empty()) { if (mode == 1) TileErr::zero(c, t.slot); return; }
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float4* Lv = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), va, Lp, Lv);
    TileErr E;
    E.init(errtab, c);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        float err1 = 0.0f;
        const bool mine = active && c.model[i] == model;
        if (mine) {
            const float4 pi = c.posm[i];
            const float4 vi = va[i];
            const float half_inv_rho = 1.0f / (2.0f * c.rho[i]);
            float r[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            struct Rec { float4 p, v; };
            for_each_ff_regs(c, gs, lo, [&](uint32_t s) { return Rec{Lp[s], Lv[s]}; }, [&](const Rec& rc) { SALVA_PAIR_MATH
                const float4 pj = rc.p, vj = rc.v;
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                const float sc = (__float_as_uint(vj.w) == model) ? g * (pj.w * half_inv_rho) : 0.0f;
                const float gx = dx * sc, gy = dy * sc, gz = dz * sc;
                const float vx = vj.x - vi.x, vy = vj.y - vi.y, vz = vj.z - vi.z;  // v_ji
                r[0] += 2.0f * vx * gx; r[1] += 2.0f * vy * gy; r[2] += 2.0f * vz * gz;
                r[3] += vx * gy + vy * gx; r[4] += vx * gz + vz * gx; r[5] += vy * gz + vz * gy;
            });
            if (mode == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) target[(size_t)k * c.n + i] = r[k] * (1.0f - coef);
            } else {
                float e[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) { e[k] = r[k] - target[(size_t)k * c.n + i]; err1 += fabsf(e[k]); }
                err1 = err1 / 6.0f;
                const float rho = c.rho[i];
                const float inv_r2 = 1.0f / (rho * rho);
                float u[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float v = 0.0f;
#pragma unroll
                    for (int b = 0; b < 6; ++b) v += beta[(size_t)(a * 6 + b) * c.n + i] * e[b];
                    u[a] = v * inv_r2;
                }
                u0[i] = make_float4(u[0], u[1], u[2], 0.0f);
                u1[i] = make_float4(u[3], u[4], u[5], __uint_as_float(model));
            }
        } else if (active && mode == 1) {
            u1[i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(c.model[i]));  // other fluids: never matched by `model`
        }
        if (mode == 1) E.add(c, err1, model, mine && !is_ghost(c, i));
    });
    if (mode == 1) E.finish(c, t.slot);
}
void launch_visc_strain(const StepCtx& c, const TileLds& L, uint32_t model, int mode, float coef, const float4* va,
                        const float* beta, float* target, float4* u0, float4* u1, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_visc_strain, c, L, model, mode, coef, va, beta, target, u0, u1, s);
    SALVA_LAUNCH_TILE(k_visc_strain, c, L, L.bytes(32, 0, 3), s, c, model, mode, coef, va, beta, target, u0, u1);
}

// ------------------------------------------------------------------------------------------------ compute_accelerations (:249-287)
// a_i += sum_j M(grad W_ij)^T ((u_i + u_j) m_j / 2) * (m_i inv_dt); refreshes va_i = v_i + a_i dt for the next strain pass.
__global__ __launch_bounds__(TILE_MAX_THREADS) void k_visc_accel(StepCtx c, uint32_t model, float inv_dt, float dt,
                                                                const float4* __restrict__ u0, const float4* __restrict__ u1,
                                                                float4* __restrict__ va) {
    if (c.ctl && c.ctl->done) return;
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    uint32_t i0_, gs0_;
    t.first_own(i0_, gs0_);
    const ListOwn lo0_ = list_own(c, i0_, gs0_);  // list head and count: in registers before the staging barrier
    const float4* Lp = nullptr;
    const float4* La = nullptr;
    const float4* Lb = nullptr;
    t.stage(c, static_cast<const float4*>(c.posm), u0, u1, Lp, La, Lb);
    Tile::staged_barrier();
    t.for_own_pre(lo0_, [&](uint32_t i_, uint32_t gs_) { return list_own(c, i_, gs_); }, [&](const ListOwn& lo, uint32_t i, uint32_t gs, bool active) {
        if (!active || c.model[i] != model) return;
        const float4 pi = c.posm[i];
        const float4 a0 = u0[i], a1 = u1[i];
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        struct Rec { float4 p, a, b; };
        const float mi_inv_dt = pi.w * inv_dt;
        for_each_ff_regs(c, gs, lo, [&](uint32_t s) { return Rec{Lp[s], La[s], Lb[s]}; }, [&](const Rec& rc) { SALVA_PAIR_MATH
            const float4 pj = rc.p;
            if (__float_as_uint(rc.b.w) != model) return;
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
            const float gx = dx * g, gy = dy * g, gz = dz * g;
            const float hm = pj.w * 0.5f;  // volumes[j] * density0 / 2
            const float c0 = (a0.x + rc.a.x) * hm, c1 = (a0.y + rc.a.y) * hm, c2 = (a0.z + rc.a.z) * hm;
            const float c3 = (a1.x + rc.b.x) * hm, c4 = (a1.y + rc.b.y) * hm, c5 = (a1.z + rc.b.z) * hm;
            ax += ((gx * 2.0f) * c0 + gy * c3 + gz * c4) * mi_inv_dt;
            ay += ((gy * 2.0f) * c1 + gx * c3 + gz * c5) * mi_inv_dt;
            az += ((gz * 2.0f) * c2 + gx * c4 + gy * c5) * mi_inv_dt;
        });
        float4 a = c.acc[i];
        a.x += ax; a.y += ay; a.z += az;
        c.acc[i] = a;
        const float4 v = c.w[i];
        va[i] = make_float4(v.x + a.x * dt, v.y + a.y * dt, v.z + a.z * dt, v.w);
    });
}
void launch_visc_accel(const StepCtx& c, const TileLds& L, uint32_t model, float inv_dt_prev, float dt_prev, const float4* u0,
                       const float4* u1, float4* va, hipStream_t s) {
    SALVA_OK_DISPATCH(launch_visc_accel, c, L, model, inv_dt_prev, dt_prev, u0, u1, va, s);
    SALVA_LAUNCH_TILE(k_visc_accel, c, L, L.bytes(48, 0, 4), s, c, model, inv_dt_prev, dt_prev, u0, u1, va);
}

}  // namespace SALVA_KNS
