// grid.hip — uniform-grid neighbour search on the device.
//
// Replaces (behaviour, not code) /root/reference/src/geometry/hgrid.rs:41-63 (cell = floor(x / h)),
// src/geometry/contacts.rs:133-151 (grid insertion), :154-400 (14-cell half stencil, all-pairs test
// d^2 <= h^2, interaction-group filter, directed contacts incl. the self contact) and the dead
// src/z_order.rs / Fluid::z_sort (fluid.rs:153-163): particles are radix-sorted by dense cell key every step,
// which is both the "hash grid" and the cache-locality sort.  The hash map + per-particle RwLock<Vec<Contact>>
// of the reference becomes: sorted SoA arrays + a dense lower-bound cell table + a sliced-ELL index list.
#include <hipcub/hipcub.hpp>

#include <climits>

#include "kernels.h"

namespace salva {

// floor(x / h) exactly as hgrid.rs:41-43 (IEEE f32 division, then floor), clamped to +-2^30.
__device__ __forceinline__ int cell_coord(float x, float h, bool& bad) {
    float f = floorf(__fdiv_rn(x, h));
    if (!(f == f)) { bad = true; f = 0.0f; }
    f = fminf(fmaxf(f, -1073741824.0f), 1073741824.0f);
    return (int)f;
}

// ------------------------------------------------------------------------------------------------ bbox
__global__ void k_bbox_init(int32_t* bbox6) {
    if (threadIdx.x < 3) bbox6[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) bbox6[threadIdx.x] = INT_MIN;
}

__global__ __launch_bounds__(BLOCK) void k_bbox(const float4* __restrict__ pts, uint32_t n, float h,
                                                int32_t* bbox6, uint32_t* flags) {
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    bool bad = false;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 p = pts[i];
        const int c[3] = {cell_coord(p.x, h, bad), cell_coord(p.y, h, bad), cell_coord(p.z, h, bad)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], c[a]); mx[a] = max(mx[a], c[a]); }
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
    if (bad) atomicOr(flags, 1u);
}

void launch_bbox_init(int32_t* bbox6, hipStream_t s) { k_bbox_init<<<1, 64, 0, s>>>(bbox6); }

void launch_bbox(const float4* pts, uint32_t n, float h, int32_t* bbox6, uint32_t* flags, hipStream_t s) {
    if (n == 0) return;
    unsigned nb = div_up(n, BLOCK);
    if (nb > 2048) nb = 2048;
    k_bbox<<<nb, BLOCK, 0, s>>>(pts, n, h, bbox6, flags);
}

// ------------------------------------------------------------------------------------------------ keys
__global__ __launch_bounds__(BLOCK) void k_cell_keys(const float4* __restrict__ pts, uint32_t n, float h, GridView g,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                     uint32_t* flags) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    bool bad = false;
    int ix = cell_coord(p.x, h, bad) - g.ox;
    int iy = cell_coord(p.y, h, bad) - g.oy;
    int iz = cell_coord(p.z, h, bad) - g.oz;
    if ((unsigned)ix >= (unsigned)g.nx || (unsigned)iy >= (unsigned)g.ny || (unsigned)iz >= (unsigned)g.nz) {
        atomicOr(flags, 2u);  // outside the table: the host recomputes the bbox and retries
        ix = min(max(ix, 0), g.nx - 1); iy = min(max(iy, 0), g.ny - 1); iz = min(max(iz, 0), g.nz - 1);
    }
    if (bad) atomicOr(flags, 1u);
    keys[i] = (uint32_t)(((size_t)ix * g.ny + iy) * g.nz + iz);
    idx[i] = i;
}

void launch_cell_keys(const float4* pts, uint32_t n, float h, GridView g, uint32_t* keys, uint32_t* idx,
                      uint32_t* flags, hipStream_t s) {
    if (n == 0) return;
    k_cell_keys<<<div_up(n, BLOCK), BLOCK, 0, s>>>(pts, n, h, g, keys, idx, flags);
}

// ------------------------------------------------------------------------------------------------ sort / scan (rocPRIM via hipCUB)
size_t sort_pairs_temp_bytes(uint32_t n, int end_bit) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, end_bit);
    return bytes;
}
void sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in,
                uint32_t* idx_out, uint32_t n, int end_bit, hipStream_t s) {
    SALVA_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (int)n, 0,
                                                       end_bit, s));
}
size_t scan_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n);
    return bytes;
}
void scan_u64(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, uint32_t n, hipStream_t s) {
    SALVA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, (int)n, s));
}

// ------------------------------------------------------------------------------------------------ cell table
// cell_start[c] = first sorted index with key >= c, for c in [0, ncells]; thread i owns the gap before key[i].
__global__ __launch_bounds__(BLOCK) void k_cell_start(const uint32_t* __restrict__ keys, uint32_t n, uint32_t ncells,
                                                      uint32_t* __restrict__ cell_start) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i > n) return;
    const uint32_t lo = (i == 0) ? 0u : keys[i - 1] + 1u;
    const uint32_t hi = (i == n) ? ncells : keys[i];  // inclusive
    for (uint32_t c = lo; c <= hi; ++c) cell_start[c] = i;
}
void launch_cell_start(const uint32_t* keys_sorted, uint32_t n, uint32_t ncells, uint32_t* cell_start, hipStream_t s) {
    k_cell_start<<<div_up((size_t)n + 1, BLOCK), BLOCK, 0, s>>>(keys_sorted, n, ncells, cell_start);
}

// ------------------------------------------------------------------------------------------------ reorder
__global__ __launch_bounds__(BLOCK) void k_reorder_fluid(uint32_t n, const uint32_t* __restrict__ idx, FluidArrays in,
                                                         FluidArrays out, float4* __restrict__ w) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    const float4 p = in.posm[j], v = in.vel[j], d = in.dv[j];
    const uint32_t m = in.model[j];
    out.posm[i] = p;
    out.vel[i] = v;
    out.dv[i] = d;
    out.model[i] = m;
    out.perm[i] = in.perm[j];
    // w = v + dv: what compute_divergences gathers (dfsph_solver.rs:323-324); model id rides in .w
    w[i] = make_float4(v.x + d.x, v.y + d.y, v.z + d.z, __uint_as_float(m));
}
void launch_reorder_fluid(uint32_t n, const uint32_t* idx, FluidArrays in, FluidArrays out, float4* w, hipStream_t s) {
    if (n == 0) return;
    k_reorder_fluid<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, idx, in, out, w);
}

__global__ __launch_bounds__(BLOCK) void k_reorder_boundary(uint32_t n, const uint32_t* __restrict__ idx,
                                                            const float4* __restrict__ bpos_in,
                                                            const float4* __restrict__ bvel_in,
                                                            const uint32_t* __restrict__ bperm_in,
                                                            float4* __restrict__ bposv_out, float4* __restrict__ bvel_out,
                                                            uint32_t* __restrict__ bperm_out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    bposv_out[i] = bpos_in[j];
    bvel_out[i] = bvel_in[j];
    bperm_out[i] = bperm_in ? bperm_in[j] : j;
}
void launch_reorder_boundary(uint32_t n, const uint32_t* idx, const float4* bpos_in, const float4* bvel_in,
                             const uint32_t* bperm_in, float4* bposv_out, float4* bvel_out, uint32_t* bperm_out,
                             hipStream_t s) {
    if (n == 0) return;
    k_reorder_boundary<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, idx, bpos_in, bvel_in, bperm_in, bposv_out, bvel_out,
                                                          bperm_out);
}

// canonical staging (host order; st_pos.w = volume, st_dv.w = pressure) -> working set with identity permutation
__global__ __launch_bounds__(BLOCK) void k_stage_to_sorted(uint32_t n, const float4* __restrict__ st_pos,
                                                           const float4* __restrict__ st_vel,
                                                           const float4* __restrict__ st_dv,
                                                           const uint32_t* __restrict__ st_model,
                                                           const float* __restrict__ rho0_tab, FluidArrays out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = st_pos[i], v = st_vel[i];
    const uint32_t m = st_model[i];
    // particle_mass = volumes[i] * density0 (fluid.rs:183-185)
    out.posm[i] = make_float4(p.x, p.y, p.z, p.w * rho0_tab[m]);
    out.vel[i] = make_float4(v.x, v.y, v.z, p.w);
    out.dv[i] = st_dv[i];
    out.model[i] = m;
    out.perm[i] = i;
}
void launch_stage_to_sorted(uint32_t n, const float4* st_pos, const float4* st_vel, const float4* st_dv,
                            const uint32_t* st_model, const float* rho0_tab, FluidArrays out, hipStream_t s) {
    if (n == 0) return;
    k_stage_to_sorted<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, st_pos, st_vel, st_dv, st_model, rho0_tab, out);
}

__global__ __launch_bounds__(BLOCK) void k_sorted_to_stage(uint32_t n, FluidArrays in, float4* __restrict__ st_pos,
                                                           float4* __restrict__ st_vel, float4* __restrict__ st_dv) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = in.perm[i];
    const float4 p = in.posm[i], v = in.vel[i];
    st_pos[o] = make_float4(p.x, p.y, p.z, v.w);
    st_vel[o] = make_float4(v.x, v.y, v.z, 0.0f);
    st_dv[o] = in.dv[i];
}
void launch_sorted_to_stage(uint32_t n, FluidArrays in, float4* st_pos, float4* st_vel, float4* st_dv, hipStream_t s) {
    if (n == 0) return;
    k_sorted_to_stage<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, in, st_pos, st_vel, st_dv);
}

__global__ __launch_bounds__(BLOCK) void k_unsort_f32(uint32_t n, const uint32_t* __restrict__ perm,
                                                      const float* __restrict__ in, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[perm[i]] = in[i];
}
__global__ __launch_bounds__(BLOCK) void k_unsort_u32_as_f32(uint32_t n, const uint32_t* __restrict__ perm,
                                                             const uint32_t* __restrict__ in, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[perm[i]] = (float)in[i];
}
__global__ __launch_bounds__(BLOCK) void k_unsort_f4(uint32_t n, const uint32_t* __restrict__ perm,
                                                     const float4* __restrict__ in, float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[perm[i]] = in[i];
}
__global__ __launch_bounds__(BLOCK) void k_gather_f4(uint32_t n, const uint32_t* __restrict__ perm,
                                                     const float4* __restrict__ in, float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}
void launch_unsort_f32(uint32_t n, const uint32_t* perm, const float* in, float* out, hipStream_t s) {
    if (n) k_unsort_f32<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, perm, in, out);
}
void launch_unsort_u32_as_f32(uint32_t n, const uint32_t* perm, const uint32_t* in, float* out, hipStream_t s) {
    if (n) k_unsort_u32_as_f32<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, perm, in, out);
}
void launch_unsort_f4(uint32_t n, const uint32_t* perm, const float4* in, float4* out, hipStream_t s) {
    if (n) k_unsort_f4<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, perm, in, out);
}
void launch_gather_f4(uint32_t n, const uint32_t* perm, const float4* in, float4* out, hipStream_t s) {
    if (n) k_gather_f4<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, perm, in, out);
}

// ------------------------------------------------------------------------------------------------ neighbour lists
// Visit every candidate in the 3x3x3 cells around (cx,cy,cz) of grid g: 9 rows, each one contiguous index run.
template <typename F>
__device__ __forceinline__ void for_each_candidate(const GridView& g, int cx, int cy, int cz, F&& f) {
    const int iz = cz - g.oz;
    const int z0 = max(iz - 1, 0), z1 = min(iz + 1, g.nz - 1);
    if (z0 > z1) return;
#pragma unroll 1
    for (int dx = -1; dx <= 1; ++dx) {
        const int ix = cx + dx - g.ox;
        if ((unsigned)ix >= (unsigned)g.nx) continue;
#pragma unroll 1
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = cy + dy - g.oy;
            if ((unsigned)iy >= (unsigned)g.ny) continue;
            const size_t base = ((size_t)ix * g.ny + iy) * g.nz;
            const uint32_t b = g.cell_start[base + z0], e = g.cell_start[base + z1 + 1];
            for (uint32_t c = b; c < e; ++c) f(c);
        }
    }
}

// One lane per fluid particle.  FILL=false: count contacts (d2 <= h2, groups ok) -> counts[i], slice width.
// FILL=true: write the candidate indices into the sliced-ELL list in traversal order.
// BOUNDARY selects fluid-boundary contacts (contacts.rs:329-346,378-383) instead of fluid-fluid (:347-392).
template <bool FILL, bool BOUNDARY>
__global__ __launch_bounds__(BLOCK) void k_nbr(StepCtx c, uint32_t* __restrict__ counts, uint64_t* __restrict__ slice_w,
                                               const uint64_t* __restrict__ slice_off, uint32_t* __restrict__ nbr,
                                               unsigned long long* ncontacts) {
    __shared__ float red[BLOCK / WAVE];
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x, c.xcd);
    const uint32_t i = blk * BLOCK + threadIdx.x;
    const bool active = i < c.n;
    uint32_t cnt = 0;
    if (active) {
        const float4 pi = c.posm[i];
        const uint32_t mi = c.model[i];
        bool bad = false;
        const int cx = cell_coord(pi.x, c.sc.h, bad), cy = cell_coord(pi.y, c.sc.h, bad), cz = cell_coord(pi.z, c.sc.h, bad);
        const GridView g = BOUNDARY ? c.gb : c.gf;
        const float4* __restrict__ pts = BOUNDARY ? c.bposv : c.posm;
        const uint64_t base = FILL ? slice_off[i / WAVE] + (i & (WAVE - 1)) : 0;
        const bool multi = BOUNDARY ? true : (c.nmodels > 1);
        for_each_candidate(g, cx, cy, cz, [&](uint32_t j) {
            const float4 pj = pts[j];
            const float d2 = dist2_exact(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
            if (d2 <= c.sc.h2) {
                bool ok = true;
                if (multi) {
                    if (BOUNDARY) ok = c.fb_ok[mi * c.nbmodels + __float_as_uint(c.bvel[j].w)] != 0;
                    else ok = c.ff_ok[mi * c.nmodels + c.model[j]] != 0;
                }
                if (ok) {
                    if (FILL) nbr[base + (uint64_t)WAVE * cnt] = j;
                    ++cnt;
                }
            }
        });
    }
    if (!FILL) {
        if (active) counts[i] = cnt;
        const unsigned wmax = wave_max_u32(cnt);
        if ((threadIdx.x & (WAVE - 1)) == 0 && (blk * BLOCK + (threadIdx.x & ~(WAVE - 1))) < c.n)
            slice_w[(blk * BLOCK + threadIdx.x) / WAVE] = (uint64_t)wmax * WAVE;
        // exact integer total (counts < 2^24 per block, so the float tree sum is exact)
        const float tot = block_sum((float)cnt, red);
        if (threadIdx.x == 0 && tot > 0.0f) atomicAdd(ncontacts, (unsigned long long)tot);
    }
}

void launch_nbr_count(const StepCtx& c, bool boundary, uint32_t* counts, uint64_t* slice_w,
                      unsigned long long* ncontacts, hipStream_t s) {
    if (c.n == 0) return;
    const unsigned nb = div_up(c.n, BLOCK);
    if (boundary) k_nbr<false, true><<<nb, BLOCK, 0, s>>>(c, counts, slice_w, nullptr, nullptr, ncontacts);
    else k_nbr<false, false><<<nb, BLOCK, 0, s>>>(c, counts, slice_w, nullptr, nullptr, ncontacts);
}
void launch_nbr_fill(const StepCtx& c, bool boundary, const uint64_t* slice_off, uint32_t* nbr, hipStream_t s) {
    if (c.n == 0) return;
    const unsigned nb = div_up(c.n, BLOCK);
    if (boundary) k_nbr<true, true><<<nb, BLOCK, 0, s>>>(c, nullptr, nullptr, slice_off, nbr, nullptr);
    else k_nbr<true, false><<<nb, BLOCK, 0, s>>>(c, nullptr, nullptr, slice_off, nbr, nullptr);
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
        for_each_candidate(c.gb, cx, cy, cz, [&](uint32_t j) {
            const float4 pj = c.bposv[j];
            const float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
            const float d2 = dist2_exact(dx, dy, dz);
            if (d2 <= c.sc.h2) {
                const uint32_t mj = __float_as_uint(c.bvel[j].w);
                if (mi == mj || c.bb_ok[mi * c.nbmodels + mj]) {
                    denom += kernel_weight(d2, c.sc);
                    ++cnt;
                }
            }
        });
        if (!(denom > 0.0f)) atomicOr(c.flags, 1u);  // assert!(!denominator.is_zero()) dfsph_solver.rs:92
        reinterpret_cast<float*>(&c.bposv[i])[3] = 1.0f / denom;
    }
    const float tot = block_sum((float)cnt, red);
    if (threadIdx.x == 0 && tot > 0.0f) atomicAdd(ncontacts_bb, (unsigned long long)tot);
}
void launch_boundary_volumes(const StepCtx& c, unsigned long long* ncontacts_bb, hipStream_t s) {
    if (c.nb == 0) return;
    k_boundary_volumes<<<div_up(c.nb, BLOCK), BLOCK, 0, s>>>(c, ncontacts_bb);
}

}  // namespace salva
