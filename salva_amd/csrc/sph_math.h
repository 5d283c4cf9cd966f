// sph_math.h — SPH smoothing-kernel arithmetic shared by every neighbour-sum kernel.
//
// Behaviour specified by /root/reference/src/kernel/cubic_spline_kernel.rs:12-33 (W), :55-79 (dW/dr) and
// src/kernel/kernel.rs:13-24 (gradient = (v/|v|) dW/dr, zero when |v|^2 <= eps^2).  Support radius is h
// (q = r/h in [0,1]), not 2h.  The device evaluates the gradient as g(r) * (xi - xj) with
// g = (dW/dr)/r, from one v_rsq_f32; the CPU oracle keeps the reference's operation order instead.
#pragma once
#include <hip/hip_runtime.h>

namespace salva {

struct SphConsts {
    float h;       // kernel radius = cell width (liquid_world.rs:44, contacts.rs:164-165)
    float inv_h;
    float h2;      // h*h rounded once in f32: the contact test is d2 <= h*h (contacts.rs:285,322,366)
    float wnorm;   // 8 / (pi h^3)           cubic_spline_kernel.rs:18
    float gnorm;   // wnorm / h              cubic_spline_kernel.rs:78
    float eps2;    // f32::EPSILON^2         kernel.rs:19
};

__host__ inline SphConsts make_sph_consts(float h) {
    SphConsts c;
    c.h = h;
    c.inv_h = 1.0f / h;
    c.h2 = h * h;
    c.wnorm = 8.0f / (3.14159265358979323846f * h * h * h);
    c.gnorm = c.wnorm / h;
    c.eps2 = 1.1920929e-7f * 1.1920929e-7f;
    return c;
}

// W(q) / wnorm
__device__ __forceinline__ float cubic_w_unit(float q) {
    const float q2 = q * q;
    const float a = 1.0f + (q2 * q - q2) * 6.0f;
    const float omq = 1.0f - q;
    const float b = omq * omq * omq * 2.0f;
    float r = (q <= 0.5f) ? a : b;
    return (q <= 1.0f) ? r : 0.0f;
}

// (dW/dr) / gnorm
__device__ __forceinline__ float cubic_dw_unit(float q) {
    const float a = (q * 3.0f - 2.0f) * q * 6.0f;
    const float omq = 1.0f - q;
    const float b = -omq * omq * 6.0f;
    float r = (q <= 0.5f) ? a : b;
    return (q > 1.0f || q <= 1.0e-5f) ? 0.0f : r;
}

// Squared distance with the reference's rounding: ((dx*dx + dy*dy) + dz*dz), no FMA contraction, so that
// the contact *set* (d2 <= h2) is bit-identical to the CPU reference's (nalgebra norm_squared, Rust never fuses).
__device__ __forceinline__ float dist2_exact(float dx, float dy, float dz) {
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

struct KernelEval {
    float w;  // W(|d|)
    float g;  // (dW/dr)/|d|  — gradient = g * d
};

// Full evaluation for one contact given d = xi - xj and r2 = |d|^2.
__device__ __forceinline__ KernelEval kernel_eval(float r2, const SphConsts& c) {
    KernelEval e;
    const bool nz = r2 > c.eps2;
    const float rinv = nz ? __builtin_amdgcn_rsqf(r2) : 0.0f;
    const float r = r2 * rinv;
    const float q = r * c.inv_h;
    e.w = c.wnorm * cubic_w_unit(q);
    e.g = c.gnorm * cubic_dw_unit(q) * rinv;
    return e;
}

// Gradient factor only: g = gnorm * u(q) / r with u = (3q-2) q 6 for q <= 1/2, -6 (1-q)^2 for q <= 1, 0 beyond and for
// q <= 1e-5 (cubic_spline_kernel.rs:63-77).  Trimmed for the hot loops: r2 is clamped away from 0 so that 1/r stays
// finite (|d| = 0 then lands in the q <= 1e-5 case, kernel.rs:18-24 gives 0 there too), 1-q is clamped at 0 instead
// of testing q > 1, and the factor 6 is folded into the constant.
__device__ __forceinline__ float kernel_grad(float r2, const SphConsts& c) {
    const float rinv = __builtin_amdgcn_rsqf(fmaxf(r2, 1.0e-30f));
    const float q = r2 * rinv * c.inv_h;
    const float a = (q * 3.0f - 2.0f) * q;
    const float omq = fmaxf(1.0f - q, 0.0f);
    const float b = -omq * omq;
    float u = (q <= 0.5f) ? a : b;
    u = (q <= 1.0e-5f) ? 0.0f : u;
    return (c.gnorm * 6.0f) * u * rinv;
}

// Weight only.
__device__ __forceinline__ float kernel_weight(float r2, const SphConsts& c) {
    const float r = __builtin_amdgcn_sqrtf(r2);
    return c.wnorm * cubic_w_unit(r * c.inv_h);
}

}  // namespace salva
