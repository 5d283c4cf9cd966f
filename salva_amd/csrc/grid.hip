// grid.hip — uniform-grid neighbour search on the device.
//
// Replaces (behaviour, not code) /root/reference/src/geometry/hgrid.rs:41-63 (cell = floor(x / h)),
// src/geometry/contacts.rs:133-151 (grid insertion), :154-400 (14-cell half stencil, all-pairs test
// d^2 <= h^2, interaction-group filter, directed contacts incl. the self contact) and the dead
// src/z_order.rs / Fluid::z_sort (fluid.rs:153-163): particles are radix-sorted by tile-major cell key every step,
// which is both the "hash grid" and the locality sort.  The hash map + per-particle RwLock<Vec<Contact>> of the
// reference becomes: sorted SoA arrays + a dense lower-bound cell table + per-tile sliced-ELL lists of 16-bit LDS
// slots (tile.h).
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include <climits>

#include "bbox.h"
#include "kernels.h"
#include "tile.h"

namespace salva {

// ------------------------------------------------------------------------------------------------ bbox
__global__ __launch_bounds__(BLOCK) void k_bbox(const float4* __restrict__ pts, uint32_t n, float h, int32_t* partials,
                                                uint32_t* flags) {
    __shared__ int red[6 * (BLOCK / WAVE)];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    bool bad = false;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 p = pts[i];
        const int c[3] = {cell_coord(p.x, h, bad), cell_coord(p.y, h, bad), cell_coord(p.z, h, bad)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], c[a]); mx[a] = max(mx[a], c[a]); }
    }
    block_bbox_store(mn, mx, red, partials + 6 * blockIdx.x);
    if (bad) atomicOr(flags, 1u);
}
__global__ __launch_bounds__(BLOCK) void k_bbox_final(const int32_t* __restrict__ partials, unsigned nblocks, int32_t* bbox6, const uint32_t* gate) {
    __shared__ int red[6 * (BLOCK / WAVE)];
    if (gate && *gate == 0u) return;  // (device_types.h StepCtx::gate: the position update it follows did not run)
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (unsigned b = threadIdx.x; b < nblocks; b += BLOCK) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], partials[6 * b + a]); mx[a] = max(mx[a], partials[6 * b + 3 + a]); }
    }
    block_bbox_store(mn, mx, red, bbox6);
}
unsigned bbox_blocks(uint32_t n) {
    unsigned nb = div_up(n ? n : 1, BLOCK);
    return nb > 1024 ? 1024 : nb;
}
void launch_bbox(const float4* pts, uint32_t n, float h, int32_t* partials, int32_t* bbox6, uint32_t* flags, hipStream_t s) {
    if (n == 0) return;
    const unsigned nb = bbox_blocks(n);
    k_bbox<<<nb, BLOCK, 0, s>>>(pts, n, h, partials, flags);
    k_bbox_final<<<1, BLOCK, 0, s>>>(partials, nb, bbox6, nullptr);
}
void launch_bbox_final(const int32_t* partials, unsigned nblocks, int32_t* bbox6, hipStream_t s, const uint32_t* gate) {
    k_bbox_final<<<1, BLOCK, 0, s>>>(partials, nblocks, bbox6, gate);
}

// ------------------------------------------------------------------------------------------------ keys
// `mass_mm` (may be null): "does every fluid particle have the same mass?" (StepCtx::mass_uniform), at no cost in the usual case.
// Every thread compares its particle's mass, pts[i].w, with that of particle 0 (one address for all: a scalar load); a wave that
// sees a different one raises one of MASS_SLOTS flag words (the one blockIdx selects — a two-fluid scene has thousands of such
// waves, and as many atomics on ONE address cost 0.3 ms), and only if a plain load still finds it clear.  mass_mm[MASS_SLOTS] = the
// reference bits.  k_publish_readback folds the flags and clears them.  (The first version kept running minima / maxima with
// system-scope reads before every atomic: 76 us per launch instead of 5, profiles/r04_experiments/r04g_cfg3_kernel_stats.csv.)
__global__ __launch_bounds__(BLOCK) void k_cell_keys(const float4* __restrict__ pts, uint32_t n, float h, TileGrid g,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                     uint32_t* flags, uint32_t* mass_mm, uint32_t* __restrict__ counts,
                                                     uint32_t* __restrict__ rank, const uint32_t* gate) {
    // (gate: this launch was enqueued at the end of the PREVIOUS step for a grid that step could only predict — World::pre_enqueue_grid;
    // 0 = the prediction did not hold, the step will run its own)
    if (gate && *gate == 0u) return;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool on = i < n;
    const float4 p = pts[on ? i : 0u];
    if (mass_mm) {  // (wave-uniform)
        const uint32_t ref = __float_as_uint(pts[0].w);
        if (__builtin_amdgcn_ballot_w64(__float_as_uint(p.w) != ref) != 0ull && (threadIdx.x & (WAVE - 1)) == 0) {
            uint32_t* f = mass_mm + (blockIdx.x & (MASS_SLOTS - 1u));
            if (*f == 0u) atomicOr(f, 1u);
        }
        if (i == 0) mass_mm[MASS_SLOTS] = ref;
    }
    bool bad = false, inside = true;
    uint32_t k = 0xffffffffu;
    if (on) {
        k = tile_key(g, cell_coord(p.x, h, bad), cell_coord(p.y, h, bad), cell_coord(p.z, h, bad), inside);
        if (!inside) { atomicOr(flags, 2u); k = 0; }  // cannot happen while the bbox is maintained with the positions
        if (bad) atomicOr(flags, 1u);
        keys[i] = k;
    }
    if (!counts) {
        if (on) idx[i] = i;
        return;
    }
    // Counting sort (cell_sort below): rank = the particle's place among those of its cell, in any order — fixed afterwards.  The
    // input is last step's sorted order, so a wave holds runs of equal keys: one atomic per run (its first lane adds the run's
    // length and hands the old count to the others) instead of one per particle — 64 -> 12 us at 10^6 particles (eight lanes on the
    // same address serialise in the L2).  Every lane of the wave takes part (no early return above).
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t kprev = (uint32_t)__shfl_up((int)k, 1, WAVE);
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(lane == 0 || kprev != k);
    const unsigned long long upto = heads & ((2ull << lane) - 1ull);            // (lane 63: 2 << 63 wraps to 0, - 1 = all ones)
    const uint32_t hl = 63u - (uint32_t)__builtin_clzll(upto);                    // first lane of my run
    const unsigned long long above = hl == 63u ? 0ull : (heads >> (hl + 1u));
    const uint32_t len = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - hl;
    uint32_t base = 0u;
    if (lane == hl && on) base = atomicAdd(&counts[k], len);
    base = (uint32_t)__shfl((int)base, (int)hl, WAVE);
    if (on) rank[i] = base + (lane - hl);
}
void launch_cell_keys(const float4* pts, uint32_t n, float h, TileGrid g, uint32_t* keys, uint32_t* idx,
                      uint32_t* flags, uint32_t* mass_mm, uint32_t* counts, uint32_t* rank, hipStream_t s, const uint32_t* gate) {
    if (n == 0) return;
    k_cell_keys<<<div_up(n, BLOCK), BLOCK, 0, s>>>(pts, n, h, g, keys, idx, flags, mass_mm, counts, rank, gate);
}

// ------------------------------------------------------------------------------------------------ sort / scan (rocPRIM via hipCUB)
// rocPRIM's default switches from its merge sort to the one-sweep radix sort only above 2^20 items; at 10^6 particles
// the merge path costs 10 merge passes (~140 us) where three 8-bit radix passes over 22-bit keys cost ~40.  Lower the
// switch-over to 64 Ki items.
// ... and the keys are short (18 bits at 10^6 particles, 21 at 8 x 10^6), so the number of passes is what counts: digits of 9 or
// 10 bits (rocPRIM's tuned configurations stop at 8) sort 18 bits in two passes of 9 instead of three.
template <int BITS>
using OnesweepBits = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, BITS,
                                                         rocprim::block_radix_rank_algorithm::match>;
template <int BITS>
using SortConfigBits = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, OnesweepBits<BITS>, 65536>;
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;
static int sort_digit_bits(int end_bit) {
#ifdef SALVA_HIP_DIAG
    static const bool off = getenv("SALVA_HIP_SORT_DEFAULT_DIGITS") != nullptr;  // (A/B: rocPRIM's own 8-bit configuration)
#else
    constexpr bool off = false;
#endif
    // (measured: 18 bits in 2 x 9 instead of 3 passes saves 49 us per step at 10^6 particles; 21 bits in 2 x 11 is no faster than
    // rocPRIM's 3 x 8 at 8 x 10^6, the 2048-bucket passes cost what they save: keep its configuration from 21 bits on)
    if (off || end_bit <= 8 || end_bit > 20) return 8;
    const int passes = (end_bit + 9) / 10;
    const int bits = (end_bit + passes - 1) / passes;
    return bits < 8 ? 8 : bits;
}
template <typename Cfg>
static hipError_t sort_with(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in, uint32_t* idx_out,
                            uint32_t n, int end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs<Cfg>(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (size_t)n, 0u, (unsigned)end_bit, s);
}
static hipError_t sort_dispatch(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in,
                                uint32_t* idx_out, uint32_t n, int end_bit, hipStream_t s) {
    switch (sort_digit_bits(end_bit)) {
        case 9: return sort_with<SortConfigBits<9>>(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, end_bit, s);
        case 10: return sort_with<SortConfigBits<10>>(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, end_bit, s);
        default: return sort_with<SortConfig>(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, end_bit, s);
    }
}
size_t sort_pairs_temp_bytes(uint32_t n, int end_bit) {
    size_t bytes = 0;
    (void)sort_dispatch(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, end_bit, nullptr);
    return bytes;
}
void sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in,
                uint32_t* idx_out, uint32_t n, int end_bit, hipStream_t s) {
    SALVA_HIP_CHECK(sort_dispatch(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, end_bit, s));
}
size_t select_flagged_temp_bytes(uint32_t n) {
    size_t b = 0;
    (void)hipcub::DeviceSelect::Flagged(nullptr, b, (const float4*)nullptr, (const uint8_t*)nullptr, (float4*)nullptr, (uint32_t*)nullptr, (int)n);
    return b;
}
// stable compaction of the flagged items (filter_from_mask, solver/helper.rs:4-12)
void select_flagged_f4(void* temp, size_t temp_bytes, const float4* in, const uint8_t* flags, float4* out, uint32_t* num_selected,
                       uint32_t n, hipStream_t s) {
    SALVA_HIP_CHECK(hipcub::DeviceSelect::Flagged(temp, temp_bytes, in, flags, out, num_selected, (int)n, s));
}
size_t scan_temp_bytes(uint32_t n) {
    size_t a = 0, b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
    return a > b ? a : b;
}
void scan_u64(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, uint32_t n, hipStream_t s) {
    SALVA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, (int)n, s));
}
void scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint32_t n, hipStream_t s) {
    SALVA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, (int)n, s));
}

// ------------------------------------------------------------------------------------------------ counting sort by cell
// The fluid's sort of every step, without a radix sort: the keys are cell ids and the cell table is wanted anyway.
//   k_cell_keys     key_i, and rank_i = atomicAdd(count[key_i], 1): the particle's place among those of its cell, in whatever order
//                   the atomics were served
//   scan            count -> cell_start, in place (ncells + 1 entries; the last count is 0, so cell_start[ncells] = n)
//   k_cell_scatter  particle i -> position cell_start[key_i] + rank_i: sorted by cell, arbitrary within a cell
//   k_cell_order    within each cell, ascending source index: position = start + (number of the cell's entries below mine).
// A stable sort of (key, index) pairs from the identity is exactly "by key, then by index", so the result is bit for bit the
// radix sort's and does not depend on the order of the atomics.  k_cell_order costs O(particles of the cell) per particle: 8 reads
// in a 2r lattice; a cell that holds thousands already costs that much in every list walk.  ~35 us instead of ~110 at 10^6
// particles (two 9-bit one-sweep passes with their five fills + k_cell_start), and it grows with n, not with n log(cells).
__global__ __launch_bounds__(BLOCK) void k_cell_scatter(uint32_t n, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ rank,
                                                        const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ keys_out,
                                                        uint32_t* __restrict__ idx_tmp, const uint32_t* gate) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (gate && *gate == 0u)) return;
    const uint32_t k = keys[i], p = cell_start[k] + rank[i];
    keys_out[p] = k;
    idx_tmp[p] = i;
}
__global__ __launch_bounds__(BLOCK) void k_cell_order(uint32_t n, const uint32_t* __restrict__ keys_sorted,
                                                      const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ idx_tmp,
                                                      uint32_t* __restrict__ idx_out, const uint32_t* gate) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n || (gate && *gate == 0u)) return;
    const uint32_t k = keys_sorted[p], b = cell_start[k], e = cell_start[k + 1], v = idx_tmp[p];
    uint32_t below = 0;
    for (uint32_t q = b; q < e; ++q) below += idx_tmp[q] < v ? 1u : 0u;
    idx_out[b + below] = v;
}
size_t cell_sort_temp_bytes(uint32_t ncells) {
    size_t b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)(ncells + 1u));
    return b;
}
// cell_start holds the counts of k_cell_keys on entry (ncells + 1 entries, the last one 0) and the cell table on return
void cell_sort(void* temp, size_t temp_bytes, uint32_t n, uint32_t ncells, const uint32_t* keys, const uint32_t* rank, uint32_t* cell_start,
               uint32_t* keys_out, uint32_t* idx_tmp, uint32_t* idx_out, hipStream_t s, const uint32_t* gate) {
    // (the library scan cannot be gated: behind a closed gate it sums whatever the table holds, in bounds, and nobody indexes by it)
    SALVA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, cell_start, cell_start, (int)(ncells + 1u), s));
    if (n == 0) return;
    k_cell_scatter<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, keys, rank, cell_start, keys_out, idx_tmp, gate);
    k_cell_order<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, keys_out, cell_start, idx_tmp, idx_out, gate);
}

// ------------------------------------------------------------------------------------------------ cell table
// cell_start[c] = first sorted index with key >= c, for c in [0, ncells].  Thread i owns the gap of cells before key[i]
// (all of them get the value i).  A gap is short in a dense scene, but a bounding box blown up by a few stray particles
// has gaps of hundreds of millions of cells: there the thread fills only the two partial chunks at the ends of its gap,
// and k_cell_start_chunks streams the value into the whole key-free chunks in between.
constexpr uint32_t CELL_CHUNK = 4096;
__global__ __launch_bounds__(BLOCK) void k_cell_start(const uint32_t* __restrict__ keys, uint32_t n, uint32_t ncells,
                                                      uint32_t* __restrict__ cell_start) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i > n) return;
    const uint32_t lo = (i == 0) ? 0u : keys[i - 1] + 1u;
    const uint32_t hi = (i == n) ? ncells : keys[i];  // inclusive
    if (lo > hi) return;
    if (lo / CELL_CHUNK == hi / CELL_CHUNK) {
        for (uint32_t c = lo; c <= hi; ++c) cell_start[c] = i;
    } else {
        const uint32_t lo_end = (lo / CELL_CHUNK + 1u) * CELL_CHUNK - 1u, hi_begin = (hi / CELL_CHUNK) * CELL_CHUNK;
        for (uint32_t c = lo; c <= lo_end; ++c) cell_start[c] = i;
        for (uint32_t c = hi_begin; c <= hi; ++c) cell_start[c] = i;
    }
}
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ keys, uint32_t lo, uint32_t hi, uint32_t v) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// one workgroup per chunk: a chunk without keys lies inside one gap and takes that gap's value
__global__ __launch_bounds__(BLOCK) void k_cell_start_chunks(const uint32_t* __restrict__ keys, uint32_t n, uint32_t ncells,
                                                             uint32_t* __restrict__ cell_start) {
    __shared__ uint32_t bracket[2];
    const uint64_t c0 = (uint64_t)blockIdx.x * CELL_CHUNK;
    const uint64_t c1 = min(c0 + CELL_CHUNK, (uint64_t)ncells + 1);  // cells [c0, c1); the table has ncells + 1 entries
    if (threadIdx.x < 2) bracket[threadIdx.x] = lower_bound_u32(keys, 0, n, threadIdx.x == 0 ? (uint32_t)c0 : (uint32_t)min(c1, (uint64_t)0xffffffffu));
    __syncthreads();
    if (bracket[0] != bracket[1]) return;  // the chunk holds keys: its cells were filled by k_cell_start
    const uint32_t v = bracket[0];
    for (uint64_t c = c0 + threadIdx.x; c < c1; c += BLOCK) cell_start[c] = v;
}
void launch_cell_start(const uint32_t* keys_sorted, uint32_t n, uint32_t ncells, uint32_t* cell_start, hipStream_t s) {
    k_cell_start<<<div_up((size_t)n + 1, BLOCK), BLOCK, 0, s>>>(keys_sorted, n, ncells, cell_start);
    if ((size_t)ncells + 1 > CELL_CHUNK)  // with a single chunk no gap can span a whole one
        k_cell_start_chunks<<<div_up((size_t)ncells + 1, CELL_CHUNK), BLOCK, 0, s>>>(keys_sorted, n, ncells, cell_start);
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
    if (in.gtag) out.gtag[i] = in.gtag[j];
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
                                                            float4* __restrict__ bposv_out, float4* __restrict__ bvel_out,
                                                            uint32_t* __restrict__ bperm_out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    bposv_out[i] = bpos_in[j];
    bvel_out[i] = bvel_in[j];
    bperm_out[i] = j;
}
void launch_reorder_boundary(uint32_t n, const uint32_t* idx, const float4* bpos_in, const float4* bvel_in,
                             float4* bposv_out, float4* bvel_out, uint32_t* bperm_out, hipStream_t s) {
    if (n == 0) return;
    k_reorder_boundary<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, idx, bpos_in, bvel_in, bposv_out, bvel_out, bperm_out);
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

// ------------------------------------------------------------------------------------------------ contact export
// ParticlesContacts of one fluid as a CSR structure in HOST order (geometry/contacts.rs:57-131: what
// `contacts.particle_contacts(i)` iterates): entry = (j_model, j) with j the index inside that model's host arrays.
// A list entry is an LDS slot of the particle's tile; the tile's slot table maps it to a sorted index.  Not a hot path
// (host-side custom forces, queries, and the parity tests, which compare contact SETS with it).
__global__ __launch_bounds__(BLOCK) void k_export_contacts(StepCtx c, const uint32_t* __restrict__ keys, uint32_t slot, int boundary,
                                                           const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ model_off,
                                                           const uint32_t* __restrict__ bmodel_off, uint32_t* __restrict__ out_model,
                                                           uint32_t* __restrict__ out_j) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n || c.model[i] != slot) return;
    const uint32_t host_local = c.perm[i] - model_off[slot];
    uint64_t o = offsets[host_local];
    const uint32_t cnt = boundary ? c.nfb[i] : c.nff[i];
    const uint32_t tile = keys[i] / TCELLS;
    // (the slot that owns particle i: the tile's only one, or — a split tile, tile.h Tile::part — the part whose range holds it)
    uint32_t slot_t = c.tile_rank[tile];
    for (const uint32_t last = c.tile_rank[tile + 1]; slot_t + 1 < last && i >= c.slot_desc[slot_t].z; ++slot_t) {}
    const uint32_t own_begin = c.slot_desc[slot_t].y;
    const TileAcc a0 = c.tile_off[slot_t];
    const uint32_t gs = a0.nsl + (i - own_begin) / WAVE, lane = (i - own_begin) % WAVE;
    const uint32_t cap = boundary ? c.cap_fb : c.cap_ff;
    const uint32_t* __restrict__ p = (boundary ? c.nbr_fb : c.nbr_ff) + (size_t)gs * cap * WAVE + 4u * lane;
    const uint64_t hoff = boundary ? (c.halo_stride ? (uint64_t)slot_t * c.bhalo_stride : a0.sb)
                                   : (c.halo_stride ? (uint64_t)slot_t * c.halo_stride : a0.s);
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t d = p[ellq(k >> 1)];
        const uint32_t s = (k & 1u) ? (d >> 16) : (d & 0xffffu);
        if (boundary) {
            const uint32_t g = c.bhalo_src[hoff + s];
            const uint32_t bm = __float_as_uint(c.bvel[g].w);
            out_model[o] = bm;
            out_j[o] = c.bperm[g] - bmodel_off[bm];
        } else {
            const uint32_t g = c.halo_src[hoff + s];
            const uint32_t m = c.model[g];
            out_model[o] = m;
            out_j[o] = c.perm[g] - model_off[m];
        }
        ++o;
    }
}
// the same lists over the working set as it is: rows in sorted (local) order for every particle, fluid neighbours as local indices
__global__ __launch_bounds__(BLOCK) void k_export_contacts_local(StepCtx c, const uint32_t* __restrict__ keys, int boundary,
                                                                 const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ bmodel_off,
                                                                 uint32_t* __restrict__ out_model, uint32_t* __restrict__ out_j) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    uint64_t o = offsets[i];
    const uint32_t cnt = boundary ? c.nfb[i] : c.nff[i];
    const uint32_t tile = keys[i] / TCELLS;
    // (the slot that owns particle i: the tile's only one, or — a split tile, tile.h Tile::part — the part whose range holds it)
    uint32_t slot_t = c.tile_rank[tile];
    for (const uint32_t last = c.tile_rank[tile + 1]; slot_t + 1 < last && i >= c.slot_desc[slot_t].z; ++slot_t) {}
    const uint32_t own_begin = c.slot_desc[slot_t].y;
    const TileAcc a0 = c.tile_off[slot_t];
    const uint32_t gs = a0.nsl + (i - own_begin) / WAVE, lane = (i - own_begin) % WAVE;
    const uint32_t cap = boundary ? c.cap_fb : c.cap_ff;
    const uint32_t* __restrict__ p = (boundary ? c.nbr_fb : c.nbr_ff) + (size_t)gs * cap * WAVE + 4u * lane;
    const uint64_t hoff = boundary ? (c.halo_stride ? (uint64_t)slot_t * c.bhalo_stride : a0.sb)
                                   : (c.halo_stride ? (uint64_t)slot_t * c.halo_stride : a0.s);
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t d = p[ellq(k >> 1)];
        const uint32_t s = (k & 1u) ? (d >> 16) : (d & 0xffffu);
        if (boundary) {
            const uint32_t g = c.bhalo_src[hoff + s];
            const uint32_t bm = __float_as_uint(c.bvel[g].w);
            out_model[o] = bm;
            out_j[o] = c.bperm[g] - bmodel_off[bm];
        } else {
            const uint32_t g = c.halo_src[hoff + s];
            out_model[o] = c.model[g];
            out_j[o] = g;
        }
        ++o;
    }
}
void launch_export_contacts_local(const StepCtx& c, const uint32_t* keys, int boundary, const uint64_t* offsets, const uint32_t* bmodel_off,
                                  uint32_t* out_model, uint32_t* out_j, hipStream_t s) {
    if (c.n) k_export_contacts_local<<<div_up(c.n, BLOCK), BLOCK, 0, s>>>(c, keys, boundary, offsets, bmodel_off, out_model, out_j);
}
void launch_export_contacts(const StepCtx& c, const uint32_t* keys, uint32_t slot, int boundary, const uint64_t* offsets,
                            const uint32_t* model_off, const uint32_t* bmodel_off, uint32_t* out_model, uint32_t* out_j, hipStream_t s) {
    if (c.n) k_export_contacts<<<div_up(c.n, BLOCK), BLOCK, 0, s>>>(c, keys, slot, boundary, offsets, model_off, bmodel_off, out_model, out_j);
}
__global__ __launch_bounds__(BLOCK) void k_unsort_u32(uint32_t n, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ in,
                                                      uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[perm[i]] = in[i];
}
void launch_unsort_u32(uint32_t n, const uint32_t* perm, const uint32_t* in, uint32_t* out, hipStream_t s) {
    if (n) k_unsort_u32<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, perm, in, out);
}

// ------------------------------------------------------------------------------------------------ tile tables
// One workgroup per tile.  k_tile_count: halo sizes (fluid / boundary particles in the 6x6x6 cell box) and number
// of 64-particle slices of the tile -> tile_cnt[tile], plus their maxima (which size the LDS staging area and the
// workgroup of every tile kernel of this step).  After an exclusive scan, k_tile_halo_fill writes the flat slot
// tables: halo_src[tile_off[tile].s + slot] = sorted index of the particle staged in that slot.
constexpr int TABLE_THREADS = HCELLS <= 256 ? 256 : 320;  // >= HCELLS
static_assert(TABLE_THREADS >= HCELLS, "one thread per halo cell");

// The slots of one tile of the dense grid: none when it is empty, one (the whole tile) unless StepCtx::split_s says otherwise — then a
// tile whose fluid halo holds more than split_s particles is cut along x into halves, or, where a half's halo is still too full,
// into single planes of own cells (tile.h Tile::part); parts without a particle of their own get no slot.  codes[k] = part code of
// the k-th slot.  One thread per tile: 36 cell-row lookups per halo plane, only for the non-empty tiles of a world that splits.
__device__ __forceinline__ uint32_t tile_parts(const TileGrid& g, uint32_t t, uint32_t split_s, uint32_t (&codes)[TX]) {
    const uint32_t* __restrict__ cs = g.cell_start;
    const size_t base = (size_t)t * TCELLS;
    if (cs[base + TCELLS] == cs[base]) return 0u;
    codes[0] = TILE_PART_WHOLE;
    if (split_s == 0u || TX != 4) return 1u;
    // fluid particles per halo plane hx = 0 .. 5
    const uint32_t ttz = t % g.ntz, tty = (t / g.ntz) % g.nty, ttx = t / (g.ntz * g.nty);
    const int hcx = g.ox + (int)ttx * TX - 1, hcy = g.oy + (int)tty * TY - 1, hcz = g.oz + (int)ttz * TZ - 1;
    uint32_t P[HX];
    uint32_t total = 0u;
    for (int hx = 0; hx < HX; ++hx) {
        uint32_t cnt = 0u;
        for (int hy = 0; hy < HY; ++hy)
            for (int hz = 0; hz < HZ; ++hz) {
                bool in;
                const uint32_t k = tile_key(g, hcx + hx, hcy + hy, hcz + hz, in);
                if (in) cnt += cs[(size_t)k + 1] - cs[k];
            }
        P[hx] = cnt; total += cnt;
    }
    if (total <= split_s) return 1u;
    auto own = [&](uint32_t ux0, uint32_t len) { return cs[base + (ux0 + len) * (TY * TZ)] - cs[base + ux0 * (TY * TZ)]; };
    uint32_t n = 0u;
    const uint32_t lo = P[0] + P[1] + P[2] + P[3], hi = P[2] + P[3] + P[4] + P[5];
    if (lo <= split_s && hi <= split_s) {
        if (own(0, 2)) codes[n++] = tile_part_code(0, 2);
        if (own(2, 2)) codes[n++] = tile_part_code(2, 2);
    } else {
        for (uint32_t ux = 0; ux < (uint32_t)TX; ++ux)
            if (own(ux, 1)) codes[n++] = tile_part_code(ux, 1);
    }
    return n;
}
// one thread per tile of the dense grid: how many slots does it get?
__global__ __launch_bounds__(BLOCK) void k_tile_flags(TileGrid g, uint32_t ntiles, uint32_t* __restrict__ flags, const uint32_t* gate, uint32_t split_s) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (gate && *gate == 0u) return;
    uint32_t codes[TX];
    if (t < ntiles) flags[t] = tile_parts(g, t, split_s, codes);
    if (t == ntiles) flags[t] = 0u;
}
__global__ __launch_bounds__(BLOCK) void k_tile_ids(TileGrid g, const uint32_t* __restrict__ rank, uint32_t ntiles, uint32_t* __restrict__ tile_ids,
                                                    const uint32_t* gate, uint32_t split_s) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (gate && *gate == 0u) return;  // (must not index by ranks scanned from stale flags)
    if (t >= ntiles) return;
    const uint32_t r = rank[t], cnt = rank[t + 1] - r;
    if (cnt == 0u) return;
    if (cnt == 1u && split_s == 0u) { tile_ids[r] = t | (TILE_PART_WHOLE << TILE_PART_SHIFT); return; }
    uint32_t codes[TX];
    const uint32_t n = tile_parts(g, t, split_s, codes);  // (the decision k_tile_flags took: the same cells, the same sums)
    for (uint32_t k = 0; k < n && k < cnt; ++k) tile_ids[r + k] = t | (codes[k] << TILE_PART_SHIFT);
}
void launch_tile_slots(TileGrid g, uint32_t ntiles, uint32_t* flags, uint32_t* rank, uint32_t* tile_ids, void* temp,
                       size_t temp_bytes, hipStream_t s, const uint32_t* gate, uint32_t split_s) {
    k_tile_flags<<<div_up((size_t)ntiles + 1, BLOCK), BLOCK, 0, s>>>(g, ntiles, flags, gate, split_s);
    SALVA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, flags, rank, (int)(ntiles + 1), s));
    k_tile_ids<<<div_up((size_t)ntiles, BLOCK), BLOCK, 0, s>>>(g, rank, ntiles, tile_ids, gate, split_s);
}

__global__ __launch_bounds__(TABLE_THREADS) void k_tile_count(StepCtx c, TileAcc* __restrict__ tile_cnt, uint4* __restrict__ slot_desc) {
    Tile t;
    TileAcc a{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (launched over an upper bound of the slot count: a surplus workgroup contributes a zero entry; workgroup 0 also zeroes the
    // extra element the exclusive scan reads behind the last one)
    if (gate_closed(c)) return;  // (a pre-enqueued launch whose grid did not come true: World::pre_enqueue_grid)
    if (blockIdx.x == 0 && threadIdx.x == 1) tile_cnt[gridDim.x] = a;
    if (!t.setup_geom(c)) {
        if (threadIdx.x == 0) tile_cnt[blockIdx.x] = a;
        return;
    }
    if (threadIdx.x == 0) slot_desc[t.slot] = make_uint4(t.tile, t.own_begin, t.own_end, t.part);
    {
        TileCells tc;
        tc.build(c, t);
        a.s = tc.lstart[HCELLS];
        a.sb = tc.blstart[HCELLS];
        a.nsl = (t.own_end - t.own_begin + WAVE - 1) / WAVE;
        a.wsl = min(a.nsl, 1024u) * min(a.nsl, 1024u);
        a.nonempty = 1;
        a.max_s = (uint32_t)a.s; a.max_sb = (uint32_t)a.sb; a.max_nsl = a.nsl;
        a.max_sum = (((uint32_t)a.s + 63u) & ~63u) + (uint32_t)a.sb;
        a.max_raw = (uint32_t)a.s + (uint32_t)a.sb;
        a.heavy = (t.part != TILE_PART_WHOLE || a.s > TILE_SPLIT_S) ? 1u : 0u;
        a.ntiny = (a.nsl <= 1u && a.s <= TILE_TINY_S && a.sb <= TILE_TINY_SB) ? 1u : 0u;
        a.nlight = (!a.ntiny && tile_is_light((uint32_t)a.s, (uint32_t)a.sb)) ? 1u : 0u;
    }
    if (threadIdx.x == 0) tile_cnt[t.slot] = a;
}
__global__ __launch_bounds__(TABLE_THREADS) void k_tile_halo_fill(StepCtx c, uint32_t* __restrict__ halo_src,
                                                                 uint32_t* __restrict__ bhalo_src, uint4* __restrict__ slot_info,
                                                                 uint32_t* __restrict__ slot_order, uint32_t nfull, uint32_t nlight,
                                                                 uint32_t use) {
    Tile t;
    t.setup(c, false);
    if (threadIdx.x == 0) {
        slot_info[t.slot] = make_uint4(t.own_begin, t.own_end, t.slice_base, t.S | (t.SB << 16));
        if (slot_order) {  // launch classes this step (StepCtx::slot_order): full | light | sparse, each kind in slot order
            // (`use` bit 0: the light slots have a launch of their own, bit 1: the sparse ones; a class without one counts as full)
            const TileAcc a0 = c.tile_off[t.slot], a1 = c.tile_off[t.slot + 1];
            const uint32_t tb = (use & 2u) ? a0.ntiny : 0u, tiny = (use & 2u) ? a1.ntiny - a0.ntiny : 0u;
            // (sparse slots that have no launch of their own are light ones: tile_is_light holds for them)
            const uint32_t l0 = a0.nlight + ((use & 2u) ? 0u : a0.ntiny), l1 = a1.nlight + ((use & 2u) ? 0u : a1.ntiny);
            const uint32_t lb = (use & 1u) ? l0 : 0u, light = (use & 1u) ? l1 - l0 : 0u;
            slot_order[tiny ? nfull + nlight + tb : (light ? nfull + lb : t.slot - tb - lb)] = t.slot;
        }
    }
    TileCells tc;
    tc.build(c, t, true);
    const int sub = threadIdx.x % 16, grp = threadIdx.x / 16;
    for (int h = grp; h < HCELLS; h += TABLE_THREADS / 16) {
        const uint32_t l0 = tc.lstart[h], cnt = tc.lstart[h + 1] - l0, g0 = tc.gstart[h];
        for (uint32_t k = sub; k < cnt; k += 16) halo_src[t.hoff + l0 + k] = g0 + k;
        if (t.SB) {
            const uint32_t bl0 = tc.blstart[h], bcnt = tc.blstart[h + 1] - bl0, bg0 = tc.bgstart[h];
            for (uint32_t k = sub; k < bcnt; k += 16) bhalo_src[t.hboff + bl0 + k] = bg0 + k;
        }
    }
}
// `nslots_bound` >= number of non-empty tiles (the host does not know the exact count yet)
void launch_tile_count(const StepCtx& c, uint32_t nslots_bound, TileAcc* tile_cnt, uint4* slot_desc, hipStream_t s) {
    if (nslots_bound) k_tile_count<<<nslots_bound, TABLE_THREADS, TILE_TABLE_BYTES, s>>>(c, tile_cnt, slot_desc);
}
void launch_tile_halo_fill(const StepCtx& c, uint32_t* halo_src, uint32_t* bhalo_src, uint4* slot_info, hipStream_t s, uint32_t* slot_order,
                           uint32_t nlight, uint32_t ntiny) {
    const uint32_t use = (nlight ? 1u : 0u) | (ntiny ? 2u : 0u);
    if (c.nlaunch)
        k_tile_halo_fill<<<c.nlaunch, TABLE_THREADS, TILE_TABLE_BYTES, s>>>(c, halo_src, bhalo_src, slot_info, slot_order, c.nlaunch - nlight - ntiny,
                                                                            nlight, use);
}
size_t scan_tiles_temp_bytes(uint32_t n) {
    size_t b = 0;
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, b, (const TileAcc*)nullptr, (TileAcc*)nullptr, hipcub::Sum(), TileAcc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, (int)n);
    return b;
}
void scan_tiles(void* temp, size_t temp_bytes, const TileAcc* in, TileAcc* out, uint32_t n, hipStream_t s) {
    SALVA_HIP_CHECK(hipcub::DeviceScan::ExclusiveScan(temp, temp_bytes, in, out, hipcub::Sum(), TileAcc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, (int)n, s));
}

// ------------------------------------------------------------------------------------------------ neighbour lists
// One workgroup per tile, one pass: for every own particle, test the candidates of its 3x3x3 cells (9 rows of 3
// z-adjacent halo cells) against d^2 <= h^2 and the interaction groups, and append the halo slots of the accepted
// ones to the particle's ELL row, two 16-bit slots per dword: fluid-fluid contacts (contacts.rs:347-392) and
// fluid-boundary contacts (:329-346, :378-383).  Writes nff / nfb, and per tile {sum, max} of the list lengths
// (the sums are counters.cd.ncontacts; the maxima tell the host whether the fixed ELL capacity was enough).

// V = 0: one candidate per iteration, nested branches on every accepted candidate (the round-1 kernel; kept as the fallback for
//        worlds with more than 32 fluid or boundary models).
// V = 1: four candidates in flight per lane (independent LDS reads and distance chains) and a branch-light append: with 64
//        lanes accepting candidates on different iterations the accept path runs on almost every iteration, and in V = 0 it
//        is a ladder of ~5 scalar branches per candidate; here it is selects plus one predicated store.
// Both write the same lists in the same order.  (Parking the completed dwords in a per-wave LDS buffer and writing them out
// with 16-byte stores was tried too: the 6 KiB per wave halve the resident waves and the kernel time doubles — the loop is
// bound by per-wave latency, not by the stores)
// V = 1 is held to 64 VGPRs (8 waves per SIMD) and asks for the LDS it really carves (12 bytes per fluid slot): FOUR tiles then
// share a CU (32 waves, the CU's limit) where three did — like the solver kernels this one spends a third of a tile's life in
// dependent fixed phases (cell tables, halo staging, barriers) that only another resident tile can overlap.
#ifndef SALVA_NBR_MIN_WAVES
#define SALVA_NBR_MIN_WAVES 8
#endif
// MM: 0 = one mass or the general kernels (no mass code at all), 1 = two masses, 2 = three or four (StepCtx::two_mass / nmass)
template <int V, int MM>
__global__ __launch_bounds__(TILE_MAX_THREADS, V == 1 ? SALVA_NBR_MIN_WAVES : 4) void k_nbr_tile(StepCtx c, TileListStats* __restrict__ tile_stats) {
    __shared__ uint32_t red[8][TILE_MAX_WAVES];
    Tile t;
    t.setup(c, false);
    if (t.empty()) {
        if (threadIdx.x == 0) {
            tile_stats[t.slot] = TileListStats{0, 0, 0, 0, 0, 0};
            if (MM != 0) { c.tile_mass_bits[t.slot] = 0u; c.tile_massb_bits[t.slot] = 0u; if (MM == 2) c.tile_masscd_bits[t.slot] = make_uint2(0u, 0u); }
        }
        return;
    }
    TileCells tc;
    tc.build(c, t, true);
    const bool multi = c.nmodels > 1;
    // V = 1: the staged positions as three planes (x | y | z): four consecutive candidates of a row are then ONE 16-byte read per
    // axis that lands as two aligned register pairs, which is what the packed f32 instructions want (see the candidate loop)
    float4* Lp = V == 0 ? t.carve<float4>(t.S + 3u) : nullptr;
    // (rows are walked from the 4-aligned slot at or below their start, four at a time: S + 8, rounded to 4)
    // and the planes an odd multiple of 256 bytes apart (tile.h, P3_DS_THREE): the three reads of a trip share their base
    // address (-2 % on this kernel)
    uint32_t plane = (t.S + 8u + 63u) & ~63u;
    if (!(plane & 64u)) plane += 64u;
    float* Lx = V != 0 ? t.carve<float>(plane) : nullptr;
    float* Ly = V != 0 ? t.carve<float>(plane) : nullptr;
    float* Lz = V != 0 ? t.carve<float>(plane) : nullptr;
    uint32_t* Lm = multi ? t.carve<uint32_t>(t.S) : nullptr;
    float4* Bp = t.carve<float4>(t.SB);
    float4* Bv = t.carve<float4>(t.SB);
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    // (worlds with a few masses, StepCtx::two_mass: which of the host's mass classes does this halo hold?  One bit per class)
    uint32_t pm = 0u;
    const uint32_t cm0 = __float_as_uint(c.class_mass[0]), cm1 = __float_as_uint(c.class_mass[1]), cm2 = __float_as_uint(c.class_mass[2]),
                   cm3 = __float_as_uint(c.class_mass[3]);
    t.for_halo(c, [&](uint32_t s, uint32_t g) {
        const float4 p = c.posm[g];
        if (V == 0) Lp[s] = p;
        else { Lx[s] = p.x; Ly[s] = p.y; Lz[s] = p.z; }
        if (multi) Lm[s] = c.model[g];
        if (MM != 0) {
            const uint32_t mb = __float_as_uint(p.w);
            const uint32_t cls = mb == cm0 ? 0u : (mb == cm1 ? 1u : (mb == cm2 ? 2u : 3u));
            if (cls == 3u && mb != cm3) atomicOr(c.flags, 8u);  // the host promised these masses: another one is an internal error, reported with the step's flags
            pm |= 1u << cls;
        }
    });
    t.for_halo_boundary(c, [&](uint32_t s, uint32_t g) { Bp[s] = c.bposv[g]; Bv[s] = c.bvel[g]; });
    if (MM != 0) {
        uint32_t w = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) w |= __builtin_amdgcn_ballot_w64((pm >> k) & 1u) != 0ull ? (1u << k) : 0u;
        if (lane == 0) red[6][threadIdx.x / WAVE] = w;
    }
    __syncthreads();
    // mixed: the halo holds several masses — the lists get the lightest class first and the heavier ones behind it, class by class
    // (one walk of the candidates per class present, each over what the others leave out), so that the plane-layout kernels can sum
    // the later segments apart
    bool mixed = false;
    uint32_t npass = 1u, clspack = 0u;  // classes present, ascending, two bits each
    if (MM != 0) {  // (every thread folds the per-wave rows: red[6] is not written again)
        uint32_t present = 0u;
        for (uint32_t k = 0; k < blockDim.x / WAVE; ++k) present |= red[6][k];
        present = (uint32_t)__builtin_amdgcn_readfirstlane((int)present);  // (workgroup-uniform: scalar registers)
        npass = 0u;
        uint32_t mb0 = 0u, mb1 = 0u, mb2 = 0u, mb3 = 0u;  // (scalars: no indexed array, no scratch)
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (!((present >> k) & 1u)) continue;
            const uint32_t b = k == 0u ? cm0 : (k == 1u ? cm1 : (k == 2u ? cm2 : cm3));
            if (npass == 0u) mb0 = b; else if (npass == 1u) mb1 = b; else if (npass == 2u) mb2 = b; else mb3 = b;
            clspack |= k << (2u * npass);
            ++npass;
        }
        mixed = npass > 1u;
        if (npass == 0u) npass = 1u;
        if (threadIdx.x == 0) {
            c.tile_mass_bits[t.slot] = mb0; c.tile_massb_bits[t.slot] = mb1;
            if (MM == 2) c.tile_masscd_bits[t.slot] = make_uint2(mb2, mb3);
        }
    }
    uint32_t sum_ff = 0, sum_fb = 0, max_ff = 0, max_fb = 0, own_ff = 0, own_fb = 0;
    t.for_own([&](uint32_t i, uint32_t gs, bool active) {
        uint32_t cnt = 0, cntb = 0;
        uint32_t self_slot = 0;
        uint32_t* __restrict__ out = c.nbr_ff + (size_t)gs * c.cap_ff * WAVE + 4u * lane;
        if (active) {
        const float4 pi = c.posm[i];
        const uint32_t mi = c.model[i];
        bool bad = false;
        int lx, ly, lz;
        if (c.stale_keys) {  // own cells sit one cell inside the halo box
            const uint32_t loc = c.stale_keys[i] % TCELLS;
            lx = 1 + (int)(loc / (TY * TZ)); ly = 1 + (int)((loc / TZ) % TY); lz = 1 + (int)(loc % TZ);
        } else {
            // (relative to the grid's origin first, modulo the axis' period where the grid is folded: device_types.h TileGrid)
            const TileGrid& g = c.gf;
            lx = (int)(((uint32_t)cell_coord(pi.x, c.sc.h, bad) - (uint32_t)g.ox) & g.mx) - (t.hcx - g.ox);
            ly = (int)(((uint32_t)cell_coord(pi.y, c.sc.h, bad) - (uint32_t)g.oy) & g.my) - (t.hcy - g.oy);
            lz = (int)(((uint32_t)cell_coord(pi.z, c.sc.h, bad) - (uint32_t)g.oz) & g.mz) - (t.hcz - g.oz);
        }
        uint32_t* __restrict__ outb = c.nbr_fb + (size_t)gs * c.cap_fb * WAVE + 4u * lane;
        uint32_t pend = 0, pendb = 0;
        // interaction-group tests of this particle's model as bit masks (one bit per fluid / boundary model): no table load
        // per accepted candidate
        // (V = 1 is launched only when both model counts are <= 32)
        uint32_t ffmask = 0xffffffffu, fbmask = 0u;
        if (V != 0) {
            if (multi) {
                ffmask = 0u;
                for (uint32_t m = 0; m < c.nmodels; ++m) ffmask |= (c.ff_ok[mi * c.nmodels + m] ? 1u : 0u) << m;
            }
            if (t.SB)
                for (uint32_t m = 0; m < c.nbmodels; ++m) fbmask |= (c.fb_ok[mi * c.nbmodels + m] ? 1u : 0u) << m;
        }
        // The paired append below (two accepted candidates per trip, no test per candidate) is the single-fluid path; a world of
        // several fluids takes it too wherever no candidate can be refused: every lane of the wave interacts with every fluid
        // (InteractionGroups::default, interaction_groups.rs:64-69 — nearly every scene) and the tile's halo holds one mass class
        // (round 6: k_nbr_tile 117 -> 100 us on a block of two fluids; a wave with a restricted lane, and the mixed tiles of a
        // world with several masses, walk candidate by candidate as before)
        bool paired = !multi;
        if (V != 0 && multi && !(MM != 0 && mixed)) {
            const uint32_t allm = c.nmodels >= 32u ? 0xffffffffu : ((1u << c.nmodels) - 1u);
            paired = __builtin_amdgcn_ballot_w64((ffmask & allm) != allm) == 0ull;
        }
        int pass = 0;            // mixed tiles walk the candidates once per class present: pass 0 takes the lightest, ...
        uint32_t cnta = 0, cpack = 0;  // list length behind the first segment; behind the second | behind the third << 16
        uint32_t cur_cls = clspack & 3u;
        auto ff_allowed = [&](uint32_t s) -> bool {
            if (V != 0 && MM != 0 && mixed && (uint32_t)((c.cmask >> (2u * Lm[s])) & 3ull) != cur_cls) return false;
            return V != 0 ? ((ffmask >> Lm[s]) & 1u) != 0u : c.ff_ok[mi * c.nmodels + Lm[s]] != 0;
        };
        auto append = [&](uint32_t s) {
            if (cnt & 1u) { if ((cnt >> 1) < c.cap_ff) out[ellq(cnt >> 1)] = pend | (s << 16); }
            else pend = s;
            ++cnt;
        };
#pragma unroll 1
        for (pass = 0; pass < (MM == 0 ? 1 : (int)npass); ++pass) {
        cur_cls = (clspack >> (2u * (uint32_t)pass)) & 3u;
#pragma unroll 1
        for (int dx = -1; dx <= 1; ++dx) {
#pragma unroll 1
            for (int dy = -1; dy <= 1; ++dy) {
                const int row = ((lx + dx) * HY + (ly + dy)) * HZ + (lz - 1);
                const uint32_t b = tc.lstart[row], e = tc.lstart[row + 3];
                if (V == 0) {
                    for (uint32_t s = b; s < e; ++s) {
                        const float4 pj = Lp[s];
                        const float d2 = dist2_exact(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
                        if (d2 <= c.sc.h2 && (!multi || c.ff_ok[mi * c.nmodels + Lm[s]])) append(s);
                    }
                } else {
                    // Distance tests first, into one bit per candidate; then only the accepted candidates — a sixth of them — go
                    // through the append path.  Four candidates per trip: one ds_read_b128 per axis plane brings (x0 x1 x2 x3) as two
                    // aligned register pairs, and the eight subtractions, squares and sums of dist2_exact are v_pk_add_f32 /
                    // v_pk_mul_f32 on those pairs (each packed operation is the same IEEE operation on both halves; this file is compiled
                    // without contraction, so d^2 is bit for bit what the reference compares).  Measured: 16 % fewer VALU instructions
                    // (SQ_INSTS_VALU 63 -> 52.7 M per launch) and a quarter fewer LDS bytes for 3 % of the kernel's time — a packed f32
                    // instruction occupies the SIMD like the two scalar ones it replaces (MI355X_MICROARCH.md prices it the same way),
                    // so the eight arithmetic operations of a test are a floor; without the append loop the kernel takes 97 of its
                    // 122 us, without the list stores 111 (profiles/r03_experiments/r03n_nbr_tile.log).
                    // The row is walked from the 4-aligned slot at or below its start; the bits before the start and past the end
                    // are cleared afterwards (the planes are padded, whatever lies there is compared and dropped).
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    // The accept bit of a test comes out of integer arithmetic: d^2 and h^2 are non-negative floats, whose order is
                    // the order of their bit patterns, so the sign of bits(d^2) - (bits(h^2) + 1) is "d^2 <= h^2"; v_alignbit_b32
                    // shifts it into the mask (mask = mask << 1 | sign) — 2 VALU per test instead of compare + select + or, and no
                    // VCC round trip.  The first candidate ends up in the highest bit: one bit reversal per 32 candidates.
                    const f2 pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, piz = {pi.z, pi.z};
                    const uint32_t h2b = __float_as_uint(c.sc.h2) + 1u;
                    for (uint32_t base = b & ~3u; base < e; base += 32u) {
                        const uint32_t nc = min(e - base, 32u), nq = (nc + 3u) >> 2;
                        uint32_t rev = 0u;
                        const f4* __restrict__ x4 = reinterpret_cast<const f4*>(Lx + base);
                        const f4* __restrict__ y4 = reinterpret_cast<const f4*>(Ly + base);
                        const f4* __restrict__ z4 = reinterpret_cast<const f4*>(Lz + base);
                        for (uint32_t q = 0; q < nq; ++q) {
                            const f4 X = x4[q], Y = y4[q], Z = z4[q];
                            const f2 dxa = pix - X.xy, dxb = pix - X.zw, dya = piy - Y.xy, dyb = piy - Y.zw, dza = piz - Z.xy, dzb = piz - Z.zw;
                            const f2 da = (dxa * dxa + dya * dya) + dza * dza, db = (dxb * dxb + dyb * dyb) + dzb * dzb;
                            rev = __builtin_amdgcn_alignbit(rev, __float_as_uint(da.x) - h2b, 31);
                            rev = __builtin_amdgcn_alignbit(rev, __float_as_uint(da.y) - h2b, 31);
                            rev = __builtin_amdgcn_alignbit(rev, __float_as_uint(db.x) - h2b, 31);
                            rev = __builtin_amdgcn_alignbit(rev, __float_as_uint(db.y) - h2b, 31);
                        }
                        // candidate k of the chunk sits in bit 4 nq - 1 - k: reverse, then drop the unused low end
                        uint32_t mask = __builtin_bitreverse32(rev) >> (32u - 4u * nq);
                        if (base < b) mask &= ~((1u << (b - base)) - 1u);
                        mask &= nc >= 32u ? 0xffffffffu : ((1u << nc) - 1u);
                        if (paired) {
                            // two accepted candidates per trip = one finished list dword per trip: the lanes of a wave leave this
                            // loop after max(ceil(bits / 2)) trips instead of max(bits), and no trip branches on the parity of cnt
                            if ((cnt & 1u) && mask) {  // complete the dword the previous chunk left half full
                                const uint32_t s = base + (uint32_t)__builtin_ctz(mask);
                                mask &= mask - 1u;
                                if ((cnt >> 1) < c.cap_ff) out[ellq(cnt >> 1)] = pend | (s << 16);
                                ++cnt;
                            }
                            while (mask & (mask - 1u)) {  // at least two bits left
                                const uint32_t s0 = base + (uint32_t)__builtin_ctz(mask);
                                mask &= mask - 1u;
                                const uint32_t s1 = base + (uint32_t)__builtin_ctz(mask);
                                mask &= mask - 1u;
                                if ((cnt >> 1) < c.cap_ff) out[ellq(cnt >> 1)] = s0 | (s1 << 16);
                                cnt += 2u;
                            }
                            if (mask) { pend = base + (uint32_t)__builtin_ctz(mask); ++cnt; }
                        } else {
                            while (mask) {
                                const uint32_t s = base + (uint32_t)__builtin_ctz(mask);
                                mask &= mask - 1u;
                                if (ff_allowed(s)) append(s);
                            }
                        }
                    }
                }
                if (t.SB && pass == 0) {
                    const uint32_t bb = tc.blstart[row], be = tc.blstart[row + 3];
                    for (uint32_t s = bb; s < be; ++s) {
                        const float4 pj = Bp[s];
                        const float d2 = dist2_exact(pi.x - pj.x, pi.y - pj.y, pi.z - pj.z);
                        const uint32_t bm = __float_as_uint(Bv[s].w);
                        if (d2 <= c.sc.h2 && (V != 0 ? ((fbmask >> bm) & 1u) != 0u : c.fb_ok[mi * c.nbmodels + bm] != 0)) {
                            if (cntb & 1u) { if ((cntb >> 1) < c.cap_fb) outb[ellq(cntb >> 1)] = pendb | (s << 16); }
                            else pendb = s;
                            ++cntb;
                        }
                    }
                }
            }
        }
        if (pass == 0) cnta = cnt;
        if (MM == 2) {
            if (pass <= 1) cpack = cnt * 0x10001u;
            else if (pass == 2) cpack = (cpack & 0xffffu) | (cnt << 16);
        }
        }  // passes
        // an odd list is padded with the particle's own slot (for_each_ff2: the self contact adds nothing to gradient sums)
        const int hself = (lx * HY + ly) * HZ + lz;
        self_slot = tc.lstart[hself] + (i - tc.gstart[hself]);
        if ((cnt & 1u) && (cnt >> 1) < c.cap_ff) out[ellq(cnt >> 1)] = pend | (self_slot << 16);
        if ((cntb & 1u) && (cntb >> 1) < c.cap_fb) outb[ellq(cntb >> 1)] = pendb;
        // (a list longer than the capacity was cut: the consumers must not walk past the rows that exist; the statistics below
        // keep the true lengths, from which the host sees the overflow, grows the capacity and repeats the pass)
        c.nff[i] = min(cnt, 2u * c.cap_ff);
        c.nfb[i] = min(cntb, 2u * c.cap_fb);
        if (MM != 0) {
            const uint32_t cap2 = 2u * c.cap_ff;
            c.nffb[i] = min(cnt, cap2) - min(cnta, cap2);  // (0 in a tile of one mass: cnta == cnt)
            if (MM == 2) {
                const uint32_t cb = min(cpack & 0xffffu, cap2), cc = min(cpack >> 16, cap2);
                c.nffc[i] = (cc - cb) | ((min(cnt, cap2) - cc) << 16);
            }
        }
        sum_ff += cnt; sum_fb += cntb;
        max_ff = max(max_ff, cnt); max_fb = max(max_fb, cntb);
        if (!is_ghost(c, i)) { own_ff += cnt; own_fb += cntb; }  // (a decomposed run reports the contacts of the particles it owns)
        }
        // Pad every list of the slice with self contacts up to the longest one: the gradient passes then run a
        // wave-uniform trip count with no per-lane predicates (a lane with a shorter list would idle anyway).
        const uint32_t nq = (cnt + 1) >> 1;
        const uint32_t nq_max = min(wave_max_u32(nq), c.cap_ff);
        if (active) {
            const uint32_t pad = self_slot | (self_slot << 16);
            for (uint32_t q = nq; q < nq_max; ++q) out[ellq(q)] = pad;
        }
    });
    // per-tile statistics (integer, order independent)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum_ff += (uint32_t)__shfl_xor((int)sum_ff, o, WAVE);
        sum_fb += (uint32_t)__shfl_xor((int)sum_fb, o, WAVE);
        own_ff += (uint32_t)__shfl_xor((int)own_ff, o, WAVE);
        own_fb += (uint32_t)__shfl_xor((int)own_fb, o, WAVE);
    }
    max_ff = wave_max_u32(max_ff); max_fb = wave_max_u32(max_fb);
    const uint32_t wv = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    if (lane == 0) { red[0][wv] = sum_ff; red[1][wv] = sum_fb; red[2][wv] = max_ff; red[3][wv] = max_fb; red[4][wv] = own_ff; red[5][wv] = own_fb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        TileListStats st{0, 0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < nw; ++k) {
            st.sum_ff += red[0][k]; st.sum_fb += red[1][k]; st.own_ff += red[4][k]; st.own_fb += red[5][k];
            st.max_ff = max(st.max_ff, red[2][k]); st.max_fb = max(st.max_fb, red[3][k]);
        }
        tile_stats[t.slot] = st;
    }
}
// fold the per-tile list statistics: out = {ncontacts_ff, ncontacts_fb} (u64) and {max_ff, max_fb} (u32)
__global__ __launch_bounds__(BLOCK) void k_list_stats(const TileListStats* __restrict__ ts, uint32_t ntiles,
                                                      unsigned long long* totals2, uint32_t* maxima2, unsigned long long* own2) {
    __shared__ unsigned long long sred[4][BLOCK / WAVE];
    __shared__ uint32_t mred[2][BLOCK / WAVE];
    unsigned long long a = 0, b = 0, oa = 0, ob = 0;
    uint32_t ma = 0, mb = 0;
    for (uint32_t k = threadIdx.x; k < ntiles; k += BLOCK) {
        const TileListStats s = ts[k];
        a += s.sum_ff; b += s.sum_fb; oa += s.own_ff; ob += s.own_fb; ma = max(ma, s.max_ff); mb = max(mb, s.max_fb);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, WAVE); b += __shfl_xor(b, o, WAVE);
        oa += __shfl_xor(oa, o, WAVE); ob += __shfl_xor(ob, o, WAVE);
    }
    ma = wave_max_u32(ma); mb = wave_max_u32(mb);
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (lane == 0) { sred[0][wv] = a; sred[1][wv] = b; sred[2][wv] = oa; sred[3][wv] = ob; mred[0][wv] = ma; mred[1][wv] = mb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ta = 0, tb = 0, toa = 0, tob = 0; uint32_t xa = 0, xb = 0;
        for (int k = 0; k < BLOCK / WAVE; ++k) {
            ta += sred[0][k]; tb += sred[1][k]; toa += sred[2][k]; tob += sred[3][k]; xa = max(xa, mred[0][k]); xb = max(xb, mred[1][k]);
        }
        totals2[0] = ta; totals2[1] = tb; maxima2[0] = xa; maxima2[1] = xb;
        if (own2) { own2[0] = toa; own2[1] = tob; }
    }
}

size_t tile_list_stats_bytes(uint32_t ntiles) { return (size_t)ntiles * sizeof(TileListStats); }
void launch_nbr_build(const StepCtx& c, const TileLds& L, void* tile_stats, unsigned long long* totals2, uint32_t* maxima2,
                      unsigned long long* own2, hipStream_t s) {
    if (c.n == 0) return;
    TileListStats* ts = static_cast<TileListStats*>(tile_stats);
    if (c.nmodels > 32u || c.nbmodels > 32u) {  // (the bit-mask group tests of V = 1 hold 32 models)
        SALVA_LAUNCH_TILE((k_nbr_tile<0, 0>), c, L, L.bytes(20, 32, 4, true) + 64u, s, c, ts);  // (worlds with a few masses have at most 32 fluids)
    } else {
        // what V = 1 carves: the cell tables, three 4-byte planes of (S + 8 rounded to 4) slots, the model ids when there is more
        // than one fluid, two 16-byte boundary arrays
        // (a function of the TileLds it is handed: the launch macro evaluates it once per launch class)
        auto nbr_lds = [](const TileLds& T, uint32_t nmodels) {
            auto r16 = [](uint32_t b) { return (b + 15u) & ~15u; };
            const uint32_t plane = ((T.max_halo_fluid + 8u + 63u) & ~63u) + 64u;
            return r16(TILE_TABLE_BYTES) + 3u * r16(plane * 4u) + (nmodels > 1 ? r16(T.max_halo_fluid * 4u) : 0u) + 2u * T.max_halo_boundary * 16u + 64u;
        };
        if (!c.two_mass) SALVA_LAUNCH_TILE((k_nbr_tile<1, 0>), c, L, nbr_lds(L, c.nmodels), s, c, ts);
        else if (c.nmass <= 2u) SALVA_LAUNCH_TILE((k_nbr_tile<1, 1>), c, L, nbr_lds(L, c.nmodels), s, c, ts);
        else SALVA_LAUNCH_TILE((k_nbr_tile<1, 2>), c, L, nbr_lds(L, c.nmodels), s, c, ts);
    }
    // (totals2 == nullptr: the statistics are folded by the end-of-step publication, World::publish_enqueue — one launch less per step)
    if (totals2) k_list_stats<<<1, BLOCK, 0, s>>>(static_cast<const TileListStats*>(tile_stats), c.nlaunch, totals2, maxima2, own2);
}

}  // namespace salva
