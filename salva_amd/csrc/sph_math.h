// sph_math.h — SPH smoothing-kernel arithmetic shared by every neighbour-sum kernel.
//
// Behaviour specified by /root/reference/src/kernel/cubic_spline_kernel.rs:12-33 (W), :55-79 (dW/dr) and
// src/kernel/kernel.rs:13-24 (gradient = (v/|v|) dW/dr, zero when |v|^2 <= eps^2).  Support radius is h
// (q = r/h in [0,1]), not 2h.  The device evaluates the gradient as g(r) * (xi - xj) with
// g = (dW/dr)/r, from one v_rsq_f32; the CPU oracle keeps the reference's operation order instead.
#pragma once
#include <hip/hip_runtime.h>

namespace salva {

// Contraction policy.  The library is compiled with -ffp-contract=off: a*b+c rounds twice everywhere, as in the reference
// (Rust never fuses), so positions, velocities, cell coordinates and the d^2 <= h^2 test round like the CPU's by
// construction.  The neighbour-sum loops opt back in locally: their results are sums over contacts in an order the reference
// does not fix either (hash iteration, contacts.rs:222), compared at 1e-5 relative, and an FMA there is one VALU instead of two.
// Put SALVA_PAIR_MATH first in the body of a per-contact lambda (the pragma is lexical: it covers that compound statement).
#define SALVA_PAIR_MATH _Pragma("clang fp contract(fast)")

struct SphConsts {
    float h;       // kernel radius = cell width (liquid_world.rs:44, contacts.rs:164-165)
    float inv_h;
    float h2;      // h*h rounded once in f32: the contact test is d2 <= h*h (contacts.rs:285,322,366)
    float wnorm;   // 8 / (pi h^3)           cubic_spline_kernel.rs:18
    float gnorm;   // wnorm / h              cubic_spline_kernel.rs:78
    float eps2;    // f32::EPSILON^2         kernel.rs:19
    float tiny_r2; // (1e-5 h)^2: below it the gradient is zero (cubic_spline_kernel.rs:63-65, q <= 1e-5)
    float g18, g12, sg6;  // 18 gnorm, 12 gnorm, sqrt(6 gnorm): folded constants of kernel_grad2
    float gscale;         // 6 gnorm / h^2: the factor kernel_gfac2 leaves to its caller
    float wscale;         // wnorm / h^3: the factor kernel_wg2 leaves to its caller
    // KernelDensity / KernelGradient of DFSPHSolver<..> / IISPHSolver<..> (dfsph_solver.rs:17-20, iisph_solver.rs:17-20):
    // 0 = CubicSplineKernel (the default type parameters, everything above), 1 = Poly6Kernel, 2 = SpikyKernel,
    // 3 = ViscosityKernel (SALVA_HIP_KERNEL_* in salva_hip.h).  Non-zero kinds take the generic paths below.
    int kd, kg;
    float n_poly6, n_spiky, n_visc;  // 315 / (64 pi h^9), 15 / (pi h^6), 15 / (2 pi h^3)  (kernel/{poly6,spiky,viscosity}_kernel.rs)
};

__host__ inline SphConsts make_sph_consts(float h) {
    SphConsts c;
    c.h = h;
    c.inv_h = 1.0f / h;
    c.h2 = h * h;
    c.wnorm = 8.0f / (3.14159265358979323846f * h * h * h);
    c.gnorm = c.wnorm / h;
    c.eps2 = 1.1920929e-7f * 1.1920929e-7f;
    c.tiny_r2 = (1.0e-5f * h) * (1.0e-5f * h);
    c.g18 = 18.0f * c.gnorm;
    c.g12 = 12.0f * c.gnorm;
    c.sg6 = sqrtf(6.0f * c.gnorm);
    c.gscale = (float)(6.0 * (double)c.gnorm / ((double)h * (double)h));
    c.wscale = (float)((double)c.wnorm / ((double)h * (double)h * (double)h));
    c.kd = c.kg = 0;
    const float pi = 3.14159265358979323846f, h3 = h * h * h;
    c.n_poly6 = (315.0f / 64.0f) / (pi * (h3 * h3 * h3));
    c.n_spiky = 15.0f / (pi * (h3 * h3));
    c.n_visc = 15.0f / (2.0f * pi * h3);
    return c;
}

// W(q) / wnorm
__device__ __forceinline__ float cubic_w_unit(float q) {
    SALVA_PAIR_MATH
    const float q2 = q * q;
    const float a = 1.0f + (q2 * q - q2) * 6.0f;
    const float omq = 1.0f - q;
    const float b = omq * omq * omq * 2.0f;
    float r = (q <= 0.5f) ? a : b;
    return (q <= 1.0f) ? r : 0.0f;
}

// (dW/dr) / gnorm
__device__ __forceinline__ float cubic_dw_unit(float q) {
    SALVA_PAIR_MATH
    const float a = (q * 3.0f - 2.0f) * q * 6.0f;
    const float omq = 1.0f - q;
    const float b = -omq * omq * 6.0f;
    float r = (q <= 0.5f) ? a : b;
    return (q > 1.0f || q <= 1.0e-5f) ? 0.0f : r;
}

// Squared distance with the reference's rounding: ((dx*dx + dy*dy) + dz*dz), no FMA contraction, so that
// the contact *set* (d2 <= h2) is bit-identical to the CPU reference's (nalgebra norm_squared, Rust never fuses).
// HIP's __fmul_rn / __fadd_rn are plain `*` / `+`, and under -ffp-contract=fast the back end fuses any fmul + fadd pair it
// can see, pragmas notwithstanding (caught by tests/test_fuzz_gpu.py: k_boundary_volumes counted 48 borderline pairs more
// than the CPU on two nearly coincident plates, exactly the single-rounding result).  The empty asm statements make the
// three products and the first sum opaque values, so there is no multiply left for an add to absorb; they emit nothing.
__device__ __forceinline__ float dist2_exact(float dx, float dy, float dz) {
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    asm volatile("" : "+v"(xx), "+v"(yy), "+v"(zz));
    float xy = xx + yy;
    asm volatile("" : "+v"(xy));
    return xy + zz;
}

// A value the optimiser must treat as opaque: a product passed through it cannot be contracted into an FMA with a later
// add (cf. dist2_exact).  For the few places whose results feed the exact contact test and must round like the CPU's.
__device__ __forceinline__ float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

// a / b with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the 12-instruction correctly rounded sequence every
// plain `/` (and __fdividef, on this toolchain) expands to.  For quotients inside neighbour sums — one per contact pair — whose
// results are compared at 1e-5 relative; never for anything that feeds the exact d^2 <= h^2 test or a cell coordinate.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// The reference's other kernels (kernel/poly6_kernel.rs:8-40, spiky_kernel.rs:8-38, viscosity_kernel.rs:8-50), selectable as
// the solvers' KernelDensity / KernelGradient type parameters.  Plain scalar code: no configuration of BASELINE.json uses them,
// so they get correctness, not tuning.  They exist only in the SECOND compilation of the kernel sources (-DSALVA_OTHER_KERNELS,
// namespace salva_ok, see SALVA_KNS in common.h): as wave-uniform branches inside the default kernels they cost 2-16 VGPRs
// per kernel and pushed a dozen of them over an occupancy step (k_divergence_apply 76 -> 91, k_density_alpha 59 -> 72, ...).
__device__ __forceinline__ float other_kernel_w(int kind, float r, const SphConsts& c) {
    if (!(r <= c.h)) return 0.0f;
    if (kind == 1) { const float d = c.h * c.h - r * r; return c.n_poly6 * (d * d * d); }
    if (kind == 2) { const float d = c.h - r; return c.n_spiky * (d * d * d); }
    if (!(r > 0.0f)) return 0.0f;
    const float rr_hh = r * r / (c.h * c.h);
    return c.n_visc * (rr_hh * (1.0f - r / (2.0f * c.h)) + c.h / (2.0f * r) - 1.0f);
}
__device__ __forceinline__ float other_kernel_dw(int kind, float r, const SphConsts& c) {
    if (!(r <= c.h)) return 0.0f;
    if (kind == 1) { const float d = c.h * c.h - r * r; return c.n_poly6 * (d * d) * r * -6.0f; }
    if (kind == 2) { const float d = c.h - r; return -c.n_spiky * (d * d) * 3.0f; }
    if (!(r > 0.0f)) return 0.0f;
    const float rr = r * r, hh = c.h * c.h, hhh = hh * c.h;
    return c.n_visc * (-3.0f * rr / (2.0f * hhh) + 2.0f * r / hh - c.h / (2.0f * rr));
}
// Kernel::apply_diff (kernel.rs:18-24): dir * scalar_apply_diff(norm) when |v|^2 > eps^2, else 0 — as the factor of v
__device__ __forceinline__ float other_kernel_g(int kind, float r2, const SphConsts& c) {
    if (!(r2 > c.eps2)) return 0.0f;
    const float r = __builtin_amdgcn_sqrtf(r2);
    return other_kernel_dw(kind, r, c) / r;
}

struct KernelEval {
    float w;  // W(|d|)
    float g;  // (dW/dr)/|d|  — gradient = g * d
};

// Full evaluation for one contact given d = xi - xj and r2 = |d|^2.
__device__ __forceinline__ KernelEval kernel_eval(float r2, const SphConsts& c) {
    SALVA_PAIR_MATH
    KernelEval e;
    const bool nz = r2 > c.eps2;
    const float rinv = nz ? __builtin_amdgcn_rsqf(r2) : 0.0f;
    const float r = r2 * rinv;
    const float q = r * c.inv_h;
    e.w = c.wnorm * cubic_w_unit(q);
    e.g = c.gnorm * cubic_dw_unit(q) * rinv;
#ifdef SALVA_OTHER_KERNELS
    if (c.kd) e.w = other_kernel_w(c.kd, __builtin_amdgcn_sqrtf(r2), c);
    if (c.kg) e.g = other_kernel_g(c.kg, r2, c);
#endif
    return e;
}

// Gradient factor only: g = gnorm * u(q) / r with u = (3q-2) q 6 for q <= 1/2, -6 (1-q)^2 for q <= 1, 0 beyond and for
// q <= 1e-5 (cubic_spline_kernel.rs:63-77).  Trimmed for the hot loops: r2 is clamped away from 0 so that 1/r stays
// finite (|d| = 0 then lands in the q <= 1e-5 case, kernel.rs:18-24 gives 0 there too), 1-q is clamped at 0 instead
// of testing q > 1, and the factor 6 is folded into the constant.
__device__ __forceinline__ float kernel_grad(float r2, const SphConsts& c) {
    SALVA_PAIR_MATH
#ifdef SALVA_OTHER_KERNELS
    if (c.kg) return other_kernel_g(c.kg, r2, c);
#endif
    const float rinv = __builtin_amdgcn_rsqf(fmaxf(r2, 1.0e-30f));
    const float q = r2 * rinv * c.inv_h;
    const float a = (q * 3.0f - 2.0f) * q;
    const float omq = fmaxf(1.0f - q, 0.0f);
    const float b = -omq * omq;
    float u = (q <= 0.5f) ? a : b;
    u = (q <= 1.0e-5f) ? 0.0f : u;
    return (c.gnorm * 6.0f) * u * rinv;
}

// Two contacts at once in packed f32 (v_pk_mul_f32 / v_pk_fma_f32 issue at the scalar rate and carry two lanes' worth):
// same function as kernel_grad.  The q <= 1e-5 case is folded into the reciprocal root (rinv := 0 makes q = 0, u = 0 and
// g = 0), which also covers r2 = 0 without a clamp.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 kernel_grad2(f2 r2, const SphConsts& c) {
    SALVA_PAIR_MATH
#ifdef SALVA_OTHER_KERNELS
    if (c.kg) { f2 g; g.x = other_kernel_g(c.kg, r2.x, c); g.y = other_kernel_g(c.kg, r2.y, c); return g; }
#endif
    const float t2 = c.tiny_r2;  // (1e-5 h)^2
    f2 rinv;
    rinv.x = (r2.x > t2) ? __builtin_amdgcn_rsqf(r2.x) : 0.0f;
    rinv.y = (r2.y > t2) ? __builtin_amdgcn_rsqf(r2.y) : 0.0f;
    const f2 q = r2 * rinv * c.inv_h;
    const f2 a = (q * c.g18 - c.g12) * q;  // 6 gnorm (3q - 2) q
    f2 om = c.sg6 - q * c.sg6;             // sqrt(6 gnorm) (1 - q)
    om.x = fmaxf(om.x, 0.0f);
    om.y = fmaxf(om.y, 0.0f);
    const f2 b = -om * om;
    f2 u;
    u.x = (q.x <= 0.5f) ? a.x : b.x;
    u.y = (q.y <= 0.5f) ? a.y : b.y;
    return u * rinv;
}

// The hot-loop form for LIST contacts (two at once): returns g' with (dW/dr)/r = c.gscale * g', from r2 = |d|^2 + 1e-30.
//   (dW/dq) / (6 gnorm) = (1 - 2q)+^2 - (1 - q)+^2   — the cubic spline as a difference of two truncated quadratics, identical
// to cubic_spline_kernel.rs:63-77's piecewise form ((3q - 2) q for q <= 1/2, -(1 - q)^2 beyond) as a polynomial; in r:
//   (dW/dr)/r = (6 gnorm / h^2) [ max(h - 2r, 0)^2 - (h - r)^2 ] / r,
// and the factor 6 gnorm / h^2 (SphConsts::gscale) is applied once per particle to the finished sum, so the loop needs a
// single constant, h (one SGPR operand per packed instruction is all VOP3P takes).
// What is NOT tested here, and why it need not be:
//   * r > h: a list entry satisfies d^2 <= h^2 exactly (dist2_exact), so r <= h (1 + 2 ulp) and (h - r)^2 <= 1e-13 h^2;
//   * r2 == 0 (the self contact and the self-padding of a slice's lists): the caller adds 1e-30 to r2 (for free, as the
//     addend of the first FMA), so rinv is finite, r = 1e-15, both quadratics are h^2 exactly and g' = 0;
//   * 0 < r <= 1e-5 h (the reference returns 0 there, cubic_spline_kernel.rs:63-65): NOT reproduced — the callers take this
//     path only for slices in which k_density_alpha found no such pair (StepCtx::slice_near) and use kernel_grad otherwise.
// 11 VALU per two contacts (2 rsq, 2 max, 7 packed) against 21 for kernel_grad2.
__device__ __forceinline__ f2 kernel_gfac2(f2 r2, const SphConsts& c) {
    SALVA_PAIR_MATH
    f2 rinv;
    rinv.x = __builtin_amdgcn_rsqf(r2.x);
    rinv.y = __builtin_amdgcn_rsqf(r2.y);
    const f2 r = r2 * rinv;
    const f2 a1 = c.h - r;  // h - r
    f2 a2 = a1 - r;         // h - 2r
    a2.x = fmaxf(a2.x, 0.0f);
    a2.y = fmaxf(a2.y, 0.0f);
    const f2 u = a2 * a2 - a1 * a1;
    return u * rinv;
}

// Weight and gradient factor of two contacts at once from the same intermediates as kernel_gfac2.  With a1 = h - r >= 0 and
// a2 = (h - 2r)+ the cubic spline is W = (wnorm / h^3) (2 a1^3 - a2^3)  [q <= 1/2: 1 - 6q^2 + 6q^3; q <= 1: 2 (1 - q)^3] and
// (dW/dr)/r = gscale (a2^2 - a1^2) / r.  Returns {2 a1^3 - a2^3, (a2^2 - a1^2) / r}: the callers apply wnorm / h^3 and gscale to
// the finished sums.  r2 = 0 (self contact, padding; the caller adds 1e-30): weight h^3 exactly; the gradient factor is SET to 0 at
// r2 <= tiny_r2, as the reference does (cubic_spline_kernel.rs:63-65) — a2^2 - a1^2 is a2 a2 - fl(a1 a1) once the compiler contracts
// it, a rounding residue of ~4e-10 that 1 / r = 1e15 turns into a factor of 4e5: harmless in every sum that multiplies it by a zero
// distance, but sum |m grad W|^2 takes it squared times 1e-30, and fifteen padded trips of an ISOLATED particle that shares its
// slice with a dense clump then add up to 1.2e-5 — above the 1e-5 below which alpha is 0 (dfsph_solver.rs:208; found by the
// folded-grid test, which puts stray particles into the block's slices).
struct KernelWG2 { f2 w, g; };
__device__ __forceinline__ KernelWG2 kernel_wg2(f2 r2, const SphConsts& c) {
    SALVA_PAIR_MATH
    f2 rinv;
    rinv.x = __builtin_amdgcn_rsqf(r2.x);
    rinv.y = __builtin_amdgcn_rsqf(r2.y);
    const f2 r = r2 * rinv;
    const f2 a1 = c.h - r;
    f2 a2 = a1 - r;
    a2.x = fmaxf(a2.x, 0.0f); a2.y = fmaxf(a2.y, 0.0f);
    const f2 a1s = a1 * a1, a2s = a2 * a2;
    KernelWG2 o;
    o.g = (a2s - a1s) * rinv;
    o.g.x = (r2.x <= c.tiny_r2) ? 0.0f : o.g.x;
    o.g.y = (r2.y <= c.tiny_r2) ? 0.0f : o.g.y;
    o.w = (a1s * a1) * 2.0f - a2s * a2;
    return o;
}

// Two packed pairs at once, stage by stage (the two chains are independent: written interleaved so that the scheduler keeps
// them interleaved and one chain's latencies are covered by the other's issue slots).
__device__ __forceinline__ void kernel_grad2x2(f2 r2a, f2 r2b, const SphConsts& c, f2& ga, f2& gb) {
    SALVA_PAIR_MATH
#ifdef SALVA_OTHER_KERNELS
    if (c.kg) {
        ga.x = other_kernel_g(c.kg, r2a.x, c); ga.y = other_kernel_g(c.kg, r2a.y, c);
        gb.x = other_kernel_g(c.kg, r2b.x, c); gb.y = other_kernel_g(c.kg, r2b.y, c);
        return;
    }
#endif
    const float t2 = c.tiny_r2;
    f2 ra, rb;
    ra.x = (r2a.x > t2) ? __builtin_amdgcn_rsqf(r2a.x) : 0.0f;
    rb.x = (r2b.x > t2) ? __builtin_amdgcn_rsqf(r2b.x) : 0.0f;
    ra.y = (r2a.y > t2) ? __builtin_amdgcn_rsqf(r2a.y) : 0.0f;
    rb.y = (r2b.y > t2) ? __builtin_amdgcn_rsqf(r2b.y) : 0.0f;
    const f2 qa = r2a * ra * c.inv_h, qb = r2b * rb * c.inv_h;
    const f2 aa = (qa * c.g18 - c.g12) * qa, ab = (qb * c.g18 - c.g12) * qb;
    f2 oa = c.sg6 - qa * c.sg6, ob = c.sg6 - qb * c.sg6;
    oa.x = fmaxf(oa.x, 0.0f); ob.x = fmaxf(ob.x, 0.0f);
    oa.y = fmaxf(oa.y, 0.0f); ob.y = fmaxf(ob.y, 0.0f);
    const f2 ba = -oa * oa, bb = -ob * ob;
    f2 ua, ub;
    ua.x = (qa.x <= 0.5f) ? aa.x : ba.x; ub.x = (qb.x <= 0.5f) ? ab.x : bb.x;
    ua.y = (qa.y <= 0.5f) ? aa.y : ba.y; ub.y = (qb.y <= 0.5f) ? ab.y : bb.y;
    ga = ua * ra; gb = ub * rb;
}

// Weight only.
__device__ __forceinline__ float kernel_weight(float r2, const SphConsts& c) {
    SALVA_PAIR_MATH
    const float r = __builtin_amdgcn_sqrtf(r2);
#ifdef SALVA_OTHER_KERNELS
    if (c.kd) return other_kernel_w(c.kd, r, c);
#endif
    return c.wnorm * cubic_w_unit(r * c.inv_h);
}

}  // namespace salva
