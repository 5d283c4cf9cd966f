// pairs.h — the pair loops of the DFSPH solver kernels (dfsph.hip; shared with the kernel-development skeletons in diag/).
#pragma once
#include <cstdlib>

#include "tile.h"

namespace salva {

// staged neighbour records (LDS -> registers)
struct RecPW { float4 p, w; };   // position+mass, v+dv
struct RecPK { float4 p; float k; };  // position+mass, kappa

__device__ __forceinline__ float rho0_of(const StepCtx& c, uint32_t model) {
    return (c.nmodels == 1) ? c.rho0_single : c.rho0_tab[model];
}

// ------------------------------------------------------------------------------------------------
// The pair loops of the four solver kernels below (the kernels the step lives in).
//
// LDS layout (tile.h, stage_pw / stage_pk): the first staged array starts at LDS byte 0 and the second at a distance that is
// a compile-time constant in the DS != 0 instantiations, so a contact costs ONE address instruction (`v_lshlrev_b32_sdwa`:
// 16-bit list entry -> byte offset) for both of its ds_reads; the boundary halo rides in the tails of the same arrays.
// Arithmetic: kernel_gfac2 (sph_math.h) — 16.5 VALU per contact against 24 for the round-2 loop (ISA accounting in
// profiles/r03_isa/).  Exactness: the fast form does not reproduce the reference's `q <= 1e-5 -> 0` rule; k_density_alpha
// flags the slices that hold such a pair (c.slice_near) and those slices walk their lists with kernel_grad instead.
// ------------------------------------------------------------------------------------------------
template <uint32_t DS>
__device__ __forceinline__ uint32_t pw_dist(const StepCtx& c, const Tile& t) { return DS ? DS * 16u : (t.stage_cap(c) + t.SB) * 16u; }
template <uint32_t DS>
__device__ __forceinline__ uint32_t pk_dist(const StepCtx& c, const Tile& t) { return DS ? DS * 16u : (t.stage_cap(c) + 2u * t.SB) * 16u; }

// `o` = byte offset of the slot in a 16-byte-strided array (tile.h, entry_off16_*)
__device__ __forceinline__ RecPW load_pw(uint32_t o, uint32_t dist) {
    RecPW r{lds_ld16(o), lds_ld16(o + dist)};
    asm volatile("" ::"v"(r.w.w));  // keep the read a single ds_read_b128 (a b96 costs 8 LDS cycles, a b128 4)
    return r;
}
__device__ __forceinline__ RecPK load_pk(uint32_t o, uint32_t dist) { return RecPK{lds_ld16(o), lds_ld4((o >> 2) + dist)}; }

// sum_j m_j (w_i - w_j) . grad W_ij over the padded list of a slice without near-coincident pairs
// (AHEAD: tile.h for_each_ff2 — false inside the persistent skeletons, where a load pending at the end of the list tail would
// make the compiler drain the next tile's prefetch)
template <bool AHEAD = true>
__device__ __forceinline__ float pair_sum_velocity_divergence(const StepCtx& c, uint32_t gs, uint32_t nqu, const ListRegs& lh,
                                                              const float4& pi, const float4& wi, uint32_t dist) {
    f2 acc2 = {0.0f, 0.0f};
    const f2 tiny = {1.0e-30f, 1.0e-30f};
    for_each_ff2<AHEAD, false, true>(c, gs, nqu, lh, [&](uint32_t o) { return load_pw(o, dist);
    }, [&](const RecPW& A, const RecPW& B) { SALVA_PAIR_MATH
        const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
        f2 r2 = dz * dz + tiny;
        r2 = dy * dy + r2;
        r2 = dx * dx + r2;
        const f2 g = kernel_gfac2(r2, c.sc);
        const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
        const f2 gm = {g.x * A.p.w, g.y * B.p.w};
        acc2 += (ux * dx + uy * dy + uz * dz) * gm;
    });
    return (acc2.x + acc2.y) * c.sc.gscale;
}
// the same sum with the reference's q <= 1e-5 rule (kernel_grad), over the exact list: slices flagged by k_density_alpha
__device__ __forceinline__ float pair_sum_velocity_divergence_exact(const StepCtx& c, uint32_t i, uint32_t gs, const float4& pi,
                                                                    const float4& wi, uint32_t dist) {
    float acc = 0.0f;
    for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
        const RecPW A = load_pw(s << 4, dist);
        const float dx = pi.x - A.p.x, dy = pi.y - A.p.y, dz = pi.z - A.p.z;
        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        acc += ((wi.x - A.w.x) * dx + (wi.y - A.w.y) * dy + (wi.z - A.w.z) * dz) * g * A.p.w;
    });
    return acc;
}
// ---- the same sums over the 24-byte plane layout (tile.h stage_p3; every particle has the mass c.mass_uniform) ----
// Three ds_read_b64 per contact from ONE address VGPR (slot * 8; the planes at compile-time distances in the DS != 0
// instantiations); no multiplication by m_j per contact: the constant multiplies the finished sum.
struct RecP3 { lds_v2f xy, zu, vw; };  // (x, y) | (z, w.x) | (w.y, w.z)
template <uint32_t DS>
__device__ __forceinline__ uint32_t p3_dist8(const Tile& t) { return DS ? DS * 8u : ((t.S * 8u + 15u) & ~15u); }
template <uint32_t DS>
__device__ __forceinline__ uint32_t p2_dist8(const Tile& t) { return DS ? DS * 8u : (((t.S + t.SB) * 8u + 15u) & ~15u); }
__device__ __forceinline__ RecP3 load_p3(uint32_t o, uint32_t dist8) { return RecP3{lds_ld8(o), lds_ld8(o + dist8), lds_ld8(o + 2u * dist8)}; }
template <bool AHEAD = true>
// (`mass`: Tile::mass — the launch's uniform mass, or in a two-mass world the mass of the first segment of THIS tile's lists)
__device__ __forceinline__ float pair_sum_velocity_divergence_p3(const StepCtx& c, uint32_t gs, uint32_t nqu, const ListRegs& lh,
                                                                 const float4& pi, const float4& wi, uint32_t dist8, float mass) {
    f2 acc2 = {0.0f, 0.0f};
    const f2 tiny = {1.0e-30f, 1.0e-30f};
    for_each_ff2<AHEAD, false, 2>(c, gs, nqu, lh, [&](uint32_t o) { return load_p3(o, dist8); }, [&](const RecP3& A, const RecP3& B) { SALVA_PAIR_MATH
        const f2 dx = {pi.x - A.xy.x, pi.x - B.xy.x}, dy = {pi.y - A.xy.y, pi.y - B.xy.y}, dz = {pi.z - A.zu.x, pi.z - B.zu.x};
        f2 r2 = dz * dz + tiny;
        r2 = dy * dy + r2;
        r2 = dx * dx + r2;
        const f2 g = kernel_gfac2(r2, c.sc);
        const f2 ux = {wi.x - A.zu.y, wi.x - B.zu.y}, uy = {wi.y - A.vw.x, wi.y - B.vw.x}, uz = {wi.z - A.vw.y, wi.z - B.vw.y};
        acc2 += (ux * dx + uy * dy + uz * dz) * g;
    });
    return (acc2.x + acc2.y) * c.sc.gscale * mass;
}
// Two-mass worlds, a tile whose halo holds both masses: the same sum with the mass INSIDE — entries [0, na) of the list carry ma,
// the entries behind them mb (k_nbr_tile wrote the list that way; padding entries contribute nothing whatever their mass).  Two
// compares, two selects and one packed multiply per pair of contacts more than the uniform loop; taken by the mixed tiles only.
__device__ __forceinline__ float pair_sum_velocity_divergence_p3_two(const StepCtx& c, uint32_t gs, uint32_t nqu, const ListRegs& lh, const float4& pi,
                                                                     const float4& wi, uint32_t dist8, uint32_t na, float ma, float mb) {
    f2 acc2 = {0.0f, 0.0f};
    const f2 tiny = {1.0e-30f, 1.0e-30f};
    for_each_ff2_indexed<2>(c, gs, nqu, lh, [&](uint32_t o) { return load_p3(o, dist8); }, [&](const RecP3& A, const RecP3& B, uint32_t q) { SALVA_PAIR_MATH
        const f2 dx = {pi.x - A.xy.x, pi.x - B.xy.x}, dy = {pi.y - A.xy.y, pi.y - B.xy.y}, dz = {pi.z - A.zu.x, pi.z - B.zu.x};
        f2 r2 = dz * dz + tiny;
        r2 = dy * dy + r2;
        r2 = dx * dx + r2;
        const f2 g = kernel_gfac2(r2, c.sc);
        const f2 ux = {wi.x - A.zu.y, wi.x - B.zu.y}, uy = {wi.y - A.vw.x, wi.y - B.vw.x}, uz = {wi.z - A.vw.y, wi.z - B.vw.y};
        const f2 m = {2u * q < na ? ma : mb, 2u * q + 1u < na ? ma : mb};
        acc2 += (ux * dx + uy * dy + uz * dz) * (g * m);
    });
    return (acc2.x + acc2.y) * c.sc.gscale;
}
__device__ __forceinline__ float pair_sum_velocity_divergence_exact_p3(const StepCtx& c, uint32_t i, uint32_t gs, const float4& pi,
                                                                       const float4& wi, uint32_t dist8, float mass) {
    float acc = 0.0f;
    for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
        const RecP3 A = load_p3(s << 3, dist8);
        const float dx = pi.x - A.xy.x, dy = pi.y - A.xy.y, dz = pi.z - A.zu.x;
        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        acc += ((wi.x - A.zu.y) * dx + (wi.y - A.vw.x) * dy + (wi.z - A.vw.y) * dz) * g;
    });
    return acc * mass;
}
// ---- two-mass worlds (device_types.h StepCtx::two_mass): the second segment of a list, entries [first, cnt), read from memory —
// only in the tiles whose halo holds both masses and only for the particles that have such neighbours.  The callers add
// (m_b - m_a) x these sums to m_a x the sum over the whole list.  f(slot) per entry.
template <typename F>
__device__ __forceinline__ void for_each_ff_range(const StepCtx& c, uint32_t gs, uint32_t first, uint32_t cnt, F&& f) {
    const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gs * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
    for (uint32_t e = first; e < cnt; ++e) {
        const uint32_t d = p[ellq(e >> 1)];
        f((e & 1u) ? (d >> 16) : (d & 0xffffu));
    }
}
// sum over the second segment of (w_i - w_j) . grad W_ij (no mass)
__device__ __forceinline__ float pair_tail_velocity_divergence_p3(const StepCtx& c, uint32_t gs, uint32_t first, uint32_t cnt, const float4& pi,
                                                                  const float4& wi, uint32_t dist8) {
    float acc = 0.0f;
    for_each_ff_range(c, gs, first, cnt, [&](uint32_t s) { SALVA_PAIR_MATH
        const RecP3 A = load_p3(s << 3, dist8);
        const float dx = pi.x - A.xy.x, dy = pi.y - A.xy.y, dz = pi.z - A.zu.x;
        const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        acc += ((wi.x - A.zu.y) * dx + (wi.y - A.vw.x) * dy + (wi.z - A.vw.y) * dz) * g;
    });
    return acc;
}
// sum_j grad W_ij k_ij over the 16-byte plane layout (tile.h stage_p2), times the uniform mass
struct RecP2 { lds_v2f xy, zk; };  // (x, y) | (z, kappa)
__device__ __forceinline__ RecP2 load_p2(uint32_t o, uint32_t dist8) { return RecP2{lds_ld8(o), lds_ld8(o + dist8)}; }
#ifndef SALVA_P2_NARROW
#define SALVA_P2_NARROW false
#endif
template <typename K2>
__device__ __forceinline__ void pair_sum_gradient_p2(const StepCtx& c, uint32_t gs, uint32_t nqu, const ListRegs& lh, const float4& pi,
                                                     uint32_t dist8, float mass, K2&& kij2, float& sx, float& sy, float& sz) {
    f2 ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f};
    const f2 tiny = {1.0e-30f, 1.0e-30f};
    for_each_ff2<true, false, 2, SALVA_P2_NARROW>(c, gs, nqu, lh, [&](uint32_t o) { return load_p2(o, dist8); }, [&](const RecP2& A, const RecP2& B) { SALVA_PAIR_MATH
        const f2 dx = {pi.x - A.xy.x, pi.x - B.xy.x}, dy = {pi.y - A.xy.y, pi.y - B.xy.y}, dz = {pi.z - A.zk.x, pi.z - B.zk.x};
        f2 r2 = dz * dz + tiny;
        r2 = dy * dy + r2;
        r2 = dx * dx + r2;
        const f2 g = kernel_gfac2(r2, c.sc);
        const f2 coeff = kij2(A.zk.y, B.zk.y) * g;
        ax += dx * coeff; ay += dy * coeff; az += dz * coeff;
    });
    const float f = c.sc.gscale * mass;
    sx = (ax.x + ax.y) * f; sy = (ay.x + ay.y) * f; sz = (az.x + az.y) * f;
}
// the two-mass form of pair_sum_gradient_p2 (see pair_sum_velocity_divergence_p3_two)
template <typename K2>
__device__ __forceinline__ void pair_sum_gradient_p2_two(const StepCtx& c, uint32_t gs, uint32_t nqu, const ListRegs& lh, const float4& pi,
                                                         uint32_t dist8, uint32_t na, float ma, float mb, K2&& kij2, float& sx, float& sy, float& sz) {
    f2 ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f};
    const f2 tiny = {1.0e-30f, 1.0e-30f};
    for_each_ff2_indexed<2>(c, gs, nqu, lh, [&](uint32_t o) { return load_p2(o, dist8); }, [&](const RecP2& A, const RecP2& B, uint32_t q) { SALVA_PAIR_MATH
        const f2 dx = {pi.x - A.xy.x, pi.x - B.xy.x}, dy = {pi.y - A.xy.y, pi.y - B.xy.y}, dz = {pi.z - A.zk.x, pi.z - B.zk.x};
        f2 r2 = dz * dz + tiny;
        r2 = dy * dy + r2;
        r2 = dx * dx + r2;
        const f2 g = kernel_gfac2(r2, c.sc);
        const f2 m = {2u * q < na ? ma : mb, 2u * q + 1u < na ? ma : mb};
        const f2 coeff = kij2(A.zk.y, B.zk.y) * (g * m);
        ax += dx * coeff; ay += dy * coeff; az += dz * coeff;
    });
    sx = (ax.x + ax.y) * c.sc.gscale; sy = (ay.x + ay.y) * c.sc.gscale; sz = (az.x + az.y) * c.sc.gscale;
}
template <typename K1>
__device__ __forceinline__ void pair_sum_gradient_exact_p2(const StepCtx& c, uint32_t i, uint32_t gs, const float4& pi, uint32_t dist8,
                                                           float mass, K1&& kij1, float& sx, float& sy, float& sz) {
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
        const RecP2 A = load_p2(s << 3, dist8);
        const float dx = pi.x - A.xy.x, dy = pi.y - A.xy.y, dz = pi.z - A.zk.x;
        const float coeff = kij1(A.zk.y) * kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        ax += dx * coeff; ay += dy * coeff; az += dz * coeff;
    });
    sx = ax * mass; sy = ay * mass; sz = az * mass;
}
// sum over the second segment of grad W_ij k_ij (no mass)
template <typename K1>
__device__ __forceinline__ void pair_tail_gradient_p2(const StepCtx& c, uint32_t gs, uint32_t first, uint32_t cnt, const float4& pi, uint32_t dist8,
                                                      K1&& kij1, float& sx, float& sy, float& sz) {
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for_each_ff_range(c, gs, first, cnt, [&](uint32_t s) { SALVA_PAIR_MATH
        const RecP2 A = load_p2(s << 3, dist8);
        const float dx = pi.x - A.xy.x, dy = pi.y - A.xy.y, dz = pi.z - A.zk.x;
        const float coeff = kij1(A.zk.y) * kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        ax += dx * coeff; ay += dy * coeff; az += dz * coeff;
    });
    sx = ax; sy = ay; sz = az;
}
// a boundary halo slot of that layout: (x, y, z, V_b)
__device__ __forceinline__ float4 p2_boundary_pos(const Tile& t, uint32_t s, uint32_t dist8) {
    const uint32_t o = (t.S + s) * 8u;
    const lds_v2f xy = lds_ld8(o), zv = lds_ld8(o + dist8);
    return make_float4(xy.x, xy.y, zv.x, zv.y);
}
static inline uint32_t p2_bytes(const TileLds& L, uint32_t ds) {
    const uint32_t n = L.raw_slots();
    const uint32_t dist8 = ds ? ds * 8u : ((n * 8u + 15u) & ~15u);
    return dist8 + n * 8u + 32u;
}

// sum_j grad W_ij m_j k_ij with k_ij = f(k_j) supplied by `kij2` (two contacts at once) / `kij1`
template <typename K2>
__device__ __forceinline__ void pair_sum_gradient(const StepCtx& c, uint32_t gs, uint32_t nqu, const ListRegs& lh, const float4& pi,
                                                  uint32_t dist, K2&& kij2, float& sx, float& sy, float& sz) {
    f2 ax = {0.0f, 0.0f}, ay = {0.0f, 0.0f}, az = {0.0f, 0.0f};
    const f2 tiny = {1.0e-30f, 1.0e-30f};
    for_each_ff2<true, false, true>(c, gs, nqu, lh, [&](uint32_t o) { return load_pk(o, dist); }, [&](const RecPK& A, const RecPK& B) { SALVA_PAIR_MATH
        const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
        f2 r2 = dz * dz + tiny;
        r2 = dy * dy + r2;
        r2 = dx * dx + r2;
        const f2 g = kernel_gfac2(r2, c.sc);
        const f2 km = kij2(A.k, B.k) * f2{A.p.w, B.p.w};
        const f2 coeff = km * g;
        ax += dx * coeff; ay += dy * coeff; az += dz * coeff;
    });
    sx = (ax.x + ax.y) * c.sc.gscale; sy = (ay.x + ay.y) * c.sc.gscale; sz = (az.x + az.y) * c.sc.gscale;
}
template <typename K1>
__device__ __forceinline__ void pair_sum_gradient_exact(const StepCtx& c, uint32_t i, uint32_t gs, const float4& pi, uint32_t dist,
                                                        K1&& kij1, float& sx, float& sy, float& sz) {
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for_each_ff(c, i, gs, [&](uint32_t s) { SALVA_PAIR_MATH
        const RecPK A = load_pk(s << 4, dist);
        const float dx = pi.x - A.p.x, dy = pi.y - A.p.y, dz = pi.z - A.p.z;
        const float coeff = kij1(A.k) * A.p.w * kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
        ax += dx * coeff; ay += dy * coeff; az += dz * coeff;
    });
    sx = ax; sy = ay; sz = az;
}
// (a KernelGradient other than the cubic spline takes the same exact walk: kernel_gfac2 is the spline's form only)
__device__ __forceinline__ bool slice_is_near(const StepCtx& c, uint32_t near_word) {
#ifdef SALVA_OTHER_KERNELS
    return (__builtin_amdgcn_readfirstlane((int)near_word) | c.sc.kg) != 0;
#else
    return __builtin_amdgcn_readfirstlane((int)near_word) != 0;
#endif
}
__device__ __forceinline__ float (*carve_errtab(Tile& t))[MAX_MODELS] {
    return reinterpret_cast<float (*)[MAX_MODELS]>(t.carve<float>(TILE_MAX_WAVES * MAX_MODELS));
}

// ---- host side: launch shapes of the fixed-layout kernels (dfsph.hip, iisph.hip)
// launch one of the three layout instantiations of a solver kernel (tile.h: FIXED_DS_SMALL / _LARGE / runtime distance)
#define SALVA_LAUNCH_FIXED(kernel, DSV, c, L, lds, s, ...)                                                         \
    do {                                                                                                           \
        if ((DSV) == FIXED_DS_SMALL) SALVA_LAUNCH_TILE_3(kernel<FIXED_DS_SMALL>, kernel<FIXED_DS_SMALL>, FIXED_DS_SMALL, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else if ((DSV) == FIXED_DS_LARGE) SALVA_LAUNCH_TILE_3(kernel<FIXED_DS_LARGE>, kernel<FIXED_DS_SMALL>, FIXED_DS_SMALL, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else SALVA_LAUNCH_TILE_3(kernel<0u>, kernel<FIXED_DS_SMALL>, FIXED_DS_SMALL, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
    } while (0)
// (`level`, TileLds::ds_level = SALVA_HIP_DS_LEVEL: take the level-th larger layout than the halo needs — the tests' way to run every
// instantiation on scenes whose halos would all pick the smallest; the arithmetic does not depend on the layout)
static inline uint32_t pick_ds(uint32_t slots_needed, uint32_t level) {
    const uint32_t ds[3] = {FIXED_DS_SMALL, FIXED_DS_LARGE, 0u};
    uint32_t k = slots_needed <= FIXED_DS_SMALL ? 0u : (slots_needed <= FIXED_DS_LARGE ? 1u : 2u);
    k += level;
    return ds[k < 2u ? k : 2u];
}
// P | W kernels: both arrays hold fluid halo + boundary halo
static inline uint32_t pw_slots(const TileLds& L) { return L.sum_slots(); }
static inline uint32_t pw_bytes(const TileLds& L, uint32_t ds, bool errtab) {
    return (ds ? 2u * ds : 2u * pw_slots(L)) * 16u + (errtab ? TILE_ERR_BYTES : 0u) + 32u;
}
// plane layout (P3): 16 DS + 8 S bytes of planes (S = the launch's largest fluid halo), the boundary halo in one or two 16-byte
// arrays, the compact error table
static inline uint32_t pick_ds_p3(uint32_t s, uint32_t level) {
    const uint32_t ds[4] = {P3_DS_THREE, P3_DS_TWO, P3_DS_ONE, 0u};
    uint32_t k = s <= P3_DS_THREE ? 0u : (s <= P3_DS_TWO ? 1u : (s <= P3_DS_ONE ? 2u : 3u));
    k += level;
    return ds[k < 3u ? k : 3u];
}
static inline uint32_t p3_bytes(const TileLds& L, uint32_t ds, uint32_t nmodels, bool with_bv) {
    const uint32_t dist8 = ds ? ds * 8u : ((L.max_halo_fluid * 8u + 15u) & ~15u);
    // behind the second plane: 8 S + 16 k SB bytes of the fullest tile (k = 1 or 2 boundary arrays).  The tile with the fullest
    // fluid halo lies inside the fluid and the tile with the most boundary slots at a wall: bound the sum by the largest
    // (S + SB) of one tile (TileLds::max_raw) as well as by the two maxima
    const uint32_t bb = with_bv ? 32u : 16u;
    const uint32_t by_maxima = ((L.max_halo_fluid * 8u + 15u) & ~15u) + bb * L.max_halo_boundary;
    const uint32_t by_raw = ((L.raw_slots() * 8u + 15u) & ~15u) + (bb - 8u) * L.max_halo_boundary;
    static const uint32_t pad = getenv("SALVA_HIP_P3_PAD") ? (uint32_t)atoi(getenv("SALVA_HIP_P3_PAD")) : 0u;  // (A/B: where a CU stops taking three tiles)
    return 2u * dist8 + (by_raw < by_maxima ? by_raw : by_maxima) + ((TILE_MAX_WAVES * nmodels * 4u + 15u) & ~15u) + 32u + pad;
}
static inline uint32_t pick_ds_p2(uint32_t n, uint32_t level) {
    const uint32_t ds[4] = {P2_DS_THREE, P3_DS_TWO, P3_DS_ONE, 0u};
    uint32_t k = n <= P2_DS_THREE ? 0u : (n <= P3_DS_TWO ? 1u : (n <= P3_DS_ONE ? 2u : 3u));
    k += level;
    return ds[k < 3u ? k : 3u];
}
#define SALVA_LAUNCH_P3(kernel, DSV, c, L, lds, s, ...)                                                                                          \
    do {                                                                                                                                         \
        if ((DSV) == P3_DS_THREE) SALVA_LAUNCH_TILE_3(kernel<P3_DS_THREE>, kernel<P3_DS_THREE>, P3_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else if ((DSV) == P3_DS_TWO) SALVA_LAUNCH_TILE_3(kernel<P3_DS_TWO>, kernel<P3_DS_THREE>, P3_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else if ((DSV) == P3_DS_ONE) SALVA_LAUNCH_TILE_3(kernel<P3_DS_ONE>, kernel<P3_DS_THREE>, P3_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else SALVA_LAUNCH_TILE_3(kernel<0u>, kernel<P3_DS_THREE>, P3_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__);                             \
    } while (0)
#define SALVA_LAUNCH_P2(kernel, DSV, c, L, lds, s, ...)                                                                                          \
    do {                                                                                                                                         \
        if ((DSV) == P2_DS_THREE) SALVA_LAUNCH_TILE_3(kernel<P2_DS_THREE>, kernel<P2_DS_THREE>, P2_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else if ((DSV) == P3_DS_TWO) SALVA_LAUNCH_TILE_3(kernel<P3_DS_TWO>, kernel<P2_DS_THREE>, P2_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else if ((DSV) == P3_DS_ONE) SALVA_LAUNCH_TILE_3(kernel<P3_DS_ONE>, kernel<P2_DS_THREE>, P2_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__); \
        else SALVA_LAUNCH_TILE_3(kernel<0u>, kernel<P2_DS_THREE>, P2_DS_THREE, kernel<0u>, c, L, lds, s, __VA_ARGS__);                             \
    } while (0)
// P | K kernels: P holds fluid halo + 2 x boundary halo, K the fluid halo (4 bytes each)
static inline uint32_t pk_slots(const TileLds& L) { return L.sum_slots() + L.max_halo_boundary; }
static inline uint32_t pk_bytes(const TileLds& L, uint32_t ds) {
    return (ds ? ds : pk_slots(L)) * 16u + ((L.max_halo_fluid + 63u) & ~63u) * 4u + 32u;
}


}  // namespace salva
