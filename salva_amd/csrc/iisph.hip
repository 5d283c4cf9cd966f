// iisph.hip — IISPH (Ihmsen et al. 2013) relaxed-Jacobi pressure solver passes.
//
// Behaviour specified by /root/reference/src/solver/pressure/iisph_solver.rs (line numbers per kernel).
// Step order (:643-711): predict_advection -> advance -> integrate -> d_ii -> p *= 0.5 -> rho* -> a_ii ->
// loop { sum_j d_ij p_j ; next pressures (+error) ; swap } -> velocity changes -> v += dv ; x += v dt ; dv = 0.
// Pressures persist across steps (warm start); they ride in dv.w so the per-step cell sort carries them.
#include <climits>

#include "kernels.h"
#include "nbr_loops.h"

namespace salva {

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

// compute_dii (:144-186): d_ii = -dt^2 / rho_i^2 * sum_j m_j grad W_ij ; also p_i = 0.5 * p_i(previous step) (:673-677)
__global__ __launch_bounds__(BLOCK) void k_iisph_dii(StepCtx c, float dt) {
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x, c.xcd);
    const uint32_t i = blk * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const float4 pi = c.posm[i];
    const float rho0 = c.rho0_tab[c.model[i]];
    const float rhoi = c.rho[i];
    const float factor = -dt * dt / (rhoi * rhoi);
    float x = 0.f, y = 0.f, z = 0.f;
    for_each_ff(c, i, [&](uint32_t j) {
        const float4 pj = c.posm[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float s = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (pj.w * factor);
        x += dx * s; y += dy * s; z += dz * s;
    });
    for_each_fb(c, i, [&](uint32_t j) {
        const float4 pj = c.bposv[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float s = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (pj.w * rho0 * factor);
        x += dx * s; y += dy * s; z += dz * s;
    });
    c.dii[i] = make_float4(x, y, z, 0.0f);
    c.kappa[i] = c.dv[i].w * 0.5f;
}
void launch_iisph_dii(const StepCtx& c, float dt, hipStream_t s) {
    if (c.n) k_iisph_dii<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt);
}

// compute_predicted_densities (:92-142)
__global__ __launch_bounds__(BLOCK) void k_iisph_pred_density(StepCtx c, float dt) {
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x, c.xcd);
    const uint32_t i = blk * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const float4 pi = c.posm[i];
    const float4 wi = c.w[i];
    const float rho0 = c.rho0_tab[c.model[i]];
    float delta = 0.0f;
    for_each_ff(c, i, [&](uint32_t j) {
        const float4 pj = c.posm[j];
        const float4 wj = c.w[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        delta += pj.w * (((wi.x - wj.x) * dx + (wi.y - wj.y) * dy + (wi.z - wj.z) * dz) * g);
    });
    for_each_fb(c, i, [&](uint32_t j) {
        const float4 pj = c.bposv[j];
        const float4 vj = c.bvel[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        delta += pj.w * rho0 * (((wi.x - vj.x) * dx + (wi.y - vj.y) * dy + (wi.z - vj.z) * dz) * g);
    });
    const float rs = c.rho[i] + delta * dt;
    if (!(rs != 0.0f)) atomicOr(c.flags, 1u);  // :140
    c.rho_star[i] = rs;
}
void launch_iisph_pred_density(const StepCtx& c, float dt, hipStream_t s) {
    if (c.n) k_iisph_pred_density<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt);
}

// compute_aii (:188-233): a_ii = sum_j m_j (d_ii - d_ji) . grad W_ij with d_ji = grad W_ij dt^2 m_i / rho_i^2
__global__ __launch_bounds__(BLOCK) void k_iisph_aii(StepCtx c, float dt) {
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x, c.xcd);
    const uint32_t i = blk * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const float4 pi = c.posm[i];
    const float rho0 = c.rho0_tab[c.model[i]];
    const float rhoi = c.rho[i];
    const float factor = dt * dt * pi.w / (rhoi * rhoi);
    const float4 di = c.dii[i];
    float a = 0.0f;
    for_each_ff(c, i, [&](uint32_t j) {
        const float4 pj = c.posm[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        const float gx = dx * g, gy = dy * g, gz = dz * g;
        a += pj.w * ((di.x - gx * factor) * gx + (di.y - gy * factor) * gy + (di.z - gz * factor) * gz);
    });
    for_each_fb(c, i, [&](uint32_t j) {
        const float4 pj = c.bposv[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        const float gx = dx * g, gy = dy * g, gz = dz * g;
        a += pj.w * rho0 * ((di.x - gx * factor) * gx + (di.y - gy * factor) * gy + (di.z - gz * factor) * gz);
    });
    c.aii[i] = a;
}
void launch_iisph_aii(const StepCtx& c, float dt, hipStream_t s) {
    if (c.n) k_iisph_aii<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt);
}

// compute_dij_pjl (:235-268): sum_j d_ij p_j = dt^2 sum_j grad W_ij (-m_j p_j / rho_j^2)   (fluid neighbours only)
__global__ __launch_bounds__(BLOCK) void k_iisph_dij_pj(StepCtx c, float dt, const float* __restrict__ p) {
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x, c.xcd);
    const uint32_t i = blk * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const float4 pi = c.posm[i];
    float x = 0.f, y = 0.f, z = 0.f;
    for_each_ff(c, i, [&](uint32_t j) {
        const float4 pj = c.posm[j];
        const float rhoj = c.rho[j];
        const float pjl = p[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float s = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (-pj.w * pjl / (rhoj * rhoj));
        x += dx * s; y += dy * s; z += dz * s;
    });
    const float dt2 = dt * dt;
    c.dijpj[i] = make_float4(x * dt2, y * dt2, z * dt2, 0.0f);
}
void launch_iisph_dij_pj(const StepCtx& c, float dt, const float* p, hipStream_t s) {
    if (c.n) k_iisph_dij_pj<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt, p);
}

// compute_next_pressures (:270-353)
__global__ __launch_bounds__(BLOCK) void k_iisph_next_pressure(StepCtx c, float dt, float omega,
                                                               const float* __restrict__ p, float* __restrict__ p_next) {
    __shared__ float red[BLOCK / WAVE];
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x, c.xcd);
    const uint32_t i = blk * BLOCK + threadIdx.x;
    const bool active = i < c.n;
    float err = 0.0f;
    uint32_t mi = 0;
    if (active) {
        mi = c.model[i];
        const float a = c.aii[i];
        float pn = 0.0f;
        if (fabsf(a) > 1.0e-9f) {
            const float rho0 = c.rho0_tab[mi];
            const float4 pi = c.posm[i];
            const float prs = p[i];
            const float rhoi = c.rho[i];
            const float derr = rho0 - c.rho_star[i];
            const float4 dpi = c.dijpj[i];
            const float fji = dt * dt * pi.w / (rhoi * rhoi);
            float sum = 0.0f;
            for_each_ff(c, i, [&](uint32_t j) {
                const float4 pj = c.posm[j];
                const float4 dj = c.dii[j];
                const float4 dpj = c.dijpj[j];
                const float pjl = p[j];
                const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                const float gx = dx * g, gy = dy * g, gz = dz * g;
                // factor = dij_pjl[i] - dii[j] p_j - (dij_pjl[j] - dji p_i)   (:318-321)
                const float fx = dpi.x - dj.x * pjl - (dpj.x - gx * fji * prs);
                const float fy = dpi.y - dj.y * pjl - (dpj.y - gy * fji * prs);
                const float fz = dpi.z - dj.z * pjl - (dpj.z - gz * fji * prs);
                sum += pj.w * (fx * gx + fy * gy + fz * gz);
            });
            for_each_fb(c, i, [&](uint32_t j) {
                const float4 pj = c.bposv[j];
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
    reduce_error(c, blk, err, mi, active, red);
}
void launch_iisph_next_pressure(const StepCtx& c, float dt, float omega, const float* p, float* p_next, hipStream_t s) {
    if (c.n) k_iisph_next_pressure<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt, omega, p, p_next);
}

// compute_velocity_changes (:355-404)
__global__ __launch_bounds__(BLOCK) void k_iisph_velocity_changes(StepCtx c, float dt, const float* __restrict__ p) {
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x, c.xcd);
    const uint32_t i = blk * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const float4 pi = c.posm[i];
    const float rho0 = c.rho0_tab[c.model[i]];
    const float rhoi = c.rho[i];
    const float pri = p[i] / (rhoi * rhoi);
    float4 d = c.dv[i];
    for_each_ff(c, i, [&](uint32_t j) {
        const float4 pj = c.posm[j];
        const float rhoj = c.rho[j];
        const float pjl = p[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float s = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (dt * pj.w * (pri + pjl / (rhoj * rhoj)));
        d.x -= dx * s; d.y -= dy * s; d.z -= dz * s;
    });
    for_each_fb(c, i, [&](uint32_t j) {
        const float4 pj = c.bposv[j];
        const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        const float s = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc) * (pj.w * rho0 * pri);
        const float ax = dx * s, ay = dy * s, az = dz * s;
        d.x -= ax * dt; d.y -= ay * dt; d.z -= az * dt;
        apply_boundary_force(c, j, __float_as_uint(c.bvel[j].w), ax * pi.w, ay * pi.w, az * pi.w);
    });
    c.dv[i] = d;
}
void launch_iisph_velocity_changes(const StepCtx& c, float dt, const float* p, hipStream_t s) {
    if (c.n) k_iisph_velocity_changes<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt, p);
}

// update_velocities_and_positions (:406-420) + zero velocity changes (:707-709); stores the pressure for the
// next step's warm start and reduces the new cell bounding box.
__device__ __forceinline__ int cell_of2(float x, float h) {
    float f = floorf(__fdiv_rn(x, h));
    if (!(f == f)) f = 0.0f;
    f = fminf(fmaxf(f, -1073741824.0f), 1073741824.0f);
    return (int)f;
}
__global__ __launch_bounds__(BLOCK) void k_iisph_finish(StepCtx c, float dt, const float* __restrict__ p, int32_t* bbox6) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < c.n;
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    if (active) {
        float4 v = c.vel[i];
        const float4 d = c.dv[i];
        float4 x = c.posm[i];
        v.x += d.x; v.y += d.y; v.z += d.z;
        x.x += v.x * dt; x.y += v.y * dt; x.z += v.z * dt;
        c.vel[i] = v;
        c.posm[i] = x;
        c.dv[i] = make_float4(0.f, 0.f, 0.f, p[i]);
        if (!(x.x == x.x) || !(x.y == x.y) || !(x.z == x.z)) atomicOr(c.flags, 1u);
        mn[0] = mx[0] = cell_of2(x.x, c.sc.h); mn[1] = mx[1] = cell_of2(x.y, c.sc.h); mn[2] = mx[2] = cell_of2(x.z, c.sc.h);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min_i32(mn[a]); mx[a] = wave_max_i32(mx[a]); }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // plain (possibly stale) read first: the bound is monotone, so almost every wave skips the atomic
            if (mn[a] < bbox6[a]) atomicMin(&bbox6[a], mn[a]);
            if (mx[a] > bbox6[3 + a]) atomicMax(&bbox6[3 + a], mx[a]);
        }
    }
}
void launch_iisph_finish(const StepCtx& c, float dt, const float* p, int32_t* bbox6, hipStream_t s) {
    if (c.n) k_iisph_finish<<<num_blocks(c.n), BLOCK, 0, s>>>(c, dt, p, bbox6);
}

}  // namespace salva
