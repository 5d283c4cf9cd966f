// dist.hip — kernels of the x-slab domain decomposition (multi-GPU): particle migration, ghost-plane construction and
// per-pass ghost refresh.  No counterpart in the reference (single process, shared memory); see comm.h and DESIGN.md §6.
//
// A rank owns the particles whose cell x-coordinate lies in [lo, hi].  Each step, before the grid is built:
//   phase 1  owned particles that left the slab (cx < lo or cx > hi) are removed and sent to that neighbour; last
//            step's ghosts are dropped; arrivals are appended as owned
//   phase 2  owned particles in the TWO edge planes of each face are copied to the neighbour, which appends them as
//            ghosts.  Two planes, although the interaction range is one: a ghost of the inner plane then has its whole
//            neighbourhood on this rank, so whatever a pass computes for it from refreshed inputs is right, and only every
//            second pass of a solver iteration needs an exchange (world.hip: the refresh_* calls say which).  Sender and receiver remember the slot of every such particle (gtag), so later refreshes of a
//            single field are a gather into a dense buffer, one sendrecv, and a scatter — no searching, no sorting.
#include <hipcub/hipcub.hpp>

#include "dist.h"
#include "tile.h"

namespace salva {

// scan element: how many particles before this one are kept / sent to lo / sent to hi
struct Sel3 {
    uint32_t keep, lo, hi, pad;
    __host__ __device__ Sel3 operator+(const Sel3& o) const { return Sel3{keep + o.keep, lo + o.lo, hi + o.hi, 0u}; }
};

// mode 1 (migration): keep = owned and still inside (or beyond an open end); lo/hi = owned and left through that face.
// mode 2 (ghost planes): keep = everything; lo/hi = owned and in the edge plane facing that neighbour.
struct DistSel {
    float h;
    int lo, hi, has_lo, has_hi, mode, nbr_lo_lo, nbr_hi_hi;
};
__device__ __forceinline__ Sel3 classify(uint32_t i, const float4* __restrict__ posm, const uint32_t* __restrict__ gtag, const DistSel& p,
                                         uint32_t* flags) {
    Sel3 s{0, 0, 0, 0};
    const bool ghost = (gtag[i] & GTAG_GHOST) != 0;
    if (p.mode == 2) s.keep = 1;
    if (!ghost) {
        bool bad = false;
        const int cx = cell_coord(posm[i].x, p.h, bad);
        if (p.mode == 1) {
            // a leaver goes to the adjacent rank; flag 4 if even that rank's slab ([nbr_lo_lo, lo - 1] / [hi + 1, nbr_hi_hi],
            // open ends = INT_MIN / INT_MAX) does not hold its cell: it would be owned where nobody mirrors it
            if (p.has_lo && cx < p.lo) { s.lo = 1; if (flags && cx < p.nbr_lo_lo) atomicOr(flags, 4u); }
            else if (p.has_hi && cx > p.hi) { s.hi = 1; if (flags && cx > p.nbr_hi_hi) atomicOr(flags, 4u); }
            else s.keep = 1;
        } else {
            // two planes per face (GHOST_PLANES); <= / >= : an open-ended first / last slab may hold particles beyond its planes
            if (p.has_lo && cx <= p.lo + (GHOST_PLANES - 1)) s.lo = 1;
            if (p.has_hi && cx >= p.hi - (GHOST_PLANES - 1)) s.hi = 1;
        }
    }
    return s;
}

// Two-level exclusive scan of the three selections, in particle order (the order of migrants and ghosts must not depend on
// scheduling: it decides the order of equal-key particles after the cell sort, i.e. the summation order of every later
// pass).  Level 1: counts per block of BLOCK particles; level 2: one workgroup scans the block counts; the pack kernel
// classifies again and ranks within its block with wave ballots.  12 bytes per BLOCK particles of scan traffic instead of
// the 48 bytes per particle of a device-wide scan over per-particle records.
__global__ __launch_bounds__(BLOCK) void k_dist_count(uint32_t n, const float4* __restrict__ posm, const uint32_t* __restrict__ gtag,
                                                      DistSel p, Sel3* __restrict__ blk, uint32_t* flags) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    Sel3 s{0, 0, 0, 0};
    if (i < n) s = classify(i, posm, gtag, p, flags);
    const uint32_t keep = (uint32_t)__syncthreads_count((int)s.keep), lo = (uint32_t)__syncthreads_count((int)s.lo),
                   hi = (uint32_t)__syncthreads_count((int)s.hi);
    if (threadIdx.x == 0) blk[blockIdx.x] = Sel3{keep, lo, hi, 0u};
}
constexpr int SCAN_THREADS = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void k_dist_scan_blocks(uint32_t nblocks, const Sel3* __restrict__ blk, Sel3* __restrict__ off) {
    __shared__ uint32_t sh[3][SCAN_THREADS];
    const uint32_t per = (nblocks + SCAN_THREADS - 1) / SCAN_THREADS;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, nblocks);
    Sel3 acc{0, 0, 0, 0};
    for (uint32_t b = b0; b < b1; ++b) acc = acc + blk[b];
    sh[0][threadIdx.x] = acc.keep; sh[1][threadIdx.x] = acc.lo; sh[2][threadIdx.x] = acc.hi;
    __syncthreads();
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {  // inclusive Hillis-Steele over the per-thread sums
        uint32_t a = 0, b = 0, c = 0;
        if ((int)threadIdx.x >= d) { a = sh[0][threadIdx.x - d]; b = sh[1][threadIdx.x - d]; c = sh[2][threadIdx.x - d]; }
        __syncthreads();
        sh[0][threadIdx.x] += a; sh[1][threadIdx.x] += b; sh[2][threadIdx.x] += c;
        __syncthreads();
    }
    Sel3 run{sh[0][threadIdx.x] - acc.keep, sh[1][threadIdx.x] - acc.lo, sh[2][threadIdx.x] - acc.hi, 0u};  // exclusive
    for (uint32_t b = b0; b < b1; ++b) {
        const Sel3 v = blk[b];
        off[b] = run;
        run = run + v;
    }
    if (threadIdx.x == SCAN_THREADS - 1) off[nblocks] = Sel3{sh[0][threadIdx.x], sh[1][threadIdx.x], sh[2][threadIdx.x], 0u};
}

// owned particles per cell plane (load balancing): hist[cx - base], planes outside [base, base + len) are clamped to the ends
__global__ __launch_bounds__(BLOCK) void k_plane_hist(uint32_t n, const float4* __restrict__ posm, const uint32_t* __restrict__ gtag, float h,
                                                      int base, int len, unsigned long long* __restrict__ hist) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (gtag[i] & GTAG_GHOST)) return;
    bool bad = false;
    int p = cell_coord(posm[i].x, h, bad) - base;
    p = p < 0 ? 0 : (p >= len ? len - 1 : p);
    atomicAdd(&hist[p], 1ull);
}
void launch_plane_hist(uint32_t n, const float4* posm, const uint32_t* gtag, float h, int base, int len, unsigned long long* hist,
                       hipStream_t s) {
    if (n) k_plane_hist<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, posm, gtag, h, base, len, hist);
}

// Move / copy the selected particles.  `off` = exclusive scan of the per-block counts (off[nblocks] = totals).
__global__ __launch_bounds__(BLOCK) void k_dist_pack(uint32_t n, DistArrays in, DistArrays out, DistSel sp, const Sel3* __restrict__ off,
                                                     DistRec* __restrict__ send_lo, DistRec* __restrict__ send_hi) {
    __shared__ uint32_t wc[3][BLOCK / WAVE];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    Sel3 s{0, 0, 0, 0};
    if (i < n) s = classify(i, in.posm, in.gtag, sp, nullptr);
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long bk = __ballot(s.keep != 0), bl = __ballot(s.lo != 0), bh = __ballot(s.hi != 0);
    if (lane == 0) { wc[0][wave] = (uint32_t)__popcll(bk); wc[1][wave] = (uint32_t)__popcll(bl); wc[2][wave] = (uint32_t)__popcll(bh); }
    __syncthreads();
    Sel3 p = off[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) { p.keep += wc[0][w]; p.lo += wc[1][w]; p.hi += wc[2][w]; }
    p.keep += (uint32_t)__popcll(bk & below); p.lo += (uint32_t)__popcll(bl & below); p.hi += (uint32_t)__popcll(bh & below);
    if (i >= n) return;
    const int mode = sp.mode;
    if (mode == 2 && !(s.lo | s.hi)) { in.gtag[i] = 0u; return; }  // (not mirrored: nothing to read or move)
    const float4 pm = in.posm[i], v = in.vel[i], d = in.dv[i];
    const uint32_t m = in.model[i], g = in.gid[i];
    if (s.lo) send_lo[p.lo] = DistRec{pm, v, d, m, g, 0u, 0u};
    if (s.hi) send_hi[p.hi] = DistRec{pm, v, d, m, g, 0u, 0u};
    if (mode == 1) {
        if (s.keep) {
            out.posm[p.keep] = pm; out.vel[p.keep] = v; out.dv[p.keep] = d; out.model[p.keep] = m; out.gid[p.keep] = g;
            out.gtag[p.keep] = 0u;
        }
    } else {
        // in place: remember the send slot (a particle is mirrored to at most one side: slabs are >= 2 GHOST_PLANES thick)
        uint32_t t = 0u;
        if (s.lo) t = GTAG_BORDER_LO | p.lo;
        else if (s.hi) t = GTAG_BORDER_HI | p.hi;
        in.gtag[i] = t;
    }
}

// Append received records after the first `base` particles.
__global__ __launch_bounds__(BLOCK) void k_dist_unpack(uint32_t count, uint32_t base, const DistRec* __restrict__ recv, DistArrays out,
                                                       uint32_t tag_bits) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= count) return;
    const DistRec r = recv[k];
    const uint32_t i = base + k;
    out.posm[i] = r.posm; out.vel[i] = r.vel; out.dv[i] = r.dv; out.model[i] = r.model; out.gid[i] = r.gid;
    out.gtag[i] = tag_bits ? (tag_bits | k) : 0u;
}

// After the cell sort: where did every tagged particle end up?
__global__ __launch_bounds__(BLOCK) void k_dist_lists(uint32_t n, const uint32_t* __restrict__ gtag, uint32_t* __restrict__ send_lo_idx,
                                                      uint32_t* __restrict__ send_hi_idx, uint32_t* __restrict__ ghost_lo_idx,
                                                      uint32_t* __restrict__ ghost_hi_idx) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = gtag[i];
    if (!(t & (GTAG_BORDER_LO | GTAG_BORDER_HI))) return;
    const uint32_t slot = t & GTAG_SLOT_MASK;
    if (t & GTAG_GHOST) {
        if (t & GTAG_BORDER_LO) ghost_lo_idx[slot] = i; else ghost_hi_idx[slot] = i;
    } else {
        if (t & GTAG_BORDER_LO) send_lo_idx[slot] = i; else send_hi_idx[slot] = i;
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gather_idx(uint32_t count, const uint32_t* __restrict__ idx, const T* __restrict__ src,
                                                      T* __restrict__ dst) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < count) dst[k] = src[idx[k]];
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_scatter_idx(uint32_t count, const uint32_t* __restrict__ idx, const T* __restrict__ src,
                                                       T* __restrict__ dst) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < count) dst[idx[k]] = src[k];
}

// ---------------------------------------------------------------------------------------------------- launchers
size_t dist_scan_temp_bytes(uint32_t) { return 0; }  // (the two-level scan needs no scratch beyond sel / pos)
size_t dist_sel_bytes(uint32_t n) { return ((size_t)div_up(n ? n : 1, BLOCK) + 1) * sizeof(Sel3); }
static DistSel dist_sel(float h, int lo, int hi, bool has_lo, bool has_hi, int mode, int nbr_lo_lo, int nbr_hi_hi) {
    return DistSel{h, lo, hi, has_lo ? 1 : 0, has_hi ? 1 : 0, mode, nbr_lo_lo, nbr_hi_hi};
}
void launch_dist_select(uint32_t n, const float4* posm, const uint32_t* gtag, float h, int lo, int hi, bool has_lo, bool has_hi,
                        int mode, int nbr_lo_lo, int nbr_hi_hi, void* sel, void* pos, void*, size_t, uint32_t* flags,
                        uint32_t totals_host[3], hipStream_t s) {
    Sel3* blk = static_cast<Sel3*>(sel);
    Sel3* off = static_cast<Sel3*>(pos);
    const uint32_t nblocks = n ? div_up(n, BLOCK) : 0u;
    if (n) k_dist_count<<<nblocks, BLOCK, 0, s>>>(n, posm, gtag, dist_sel(h, lo, hi, has_lo, has_hi, mode, nbr_lo_lo, nbr_hi_hi), blk, flags);
    k_dist_scan_blocks<<<1, SCAN_THREADS, 0, s>>>(nblocks, blk, off);
    Sel3 tot;
    SALVA_HIP_CHECK(hipMemcpyAsync(&tot, off + nblocks, sizeof(Sel3), hipMemcpyDeviceToHost, s));
    SALVA_HIP_CHECK(hipStreamSynchronize(s));
    totals_host[0] = tot.keep; totals_host[1] = tot.lo; totals_host[2] = tot.hi;
}
void launch_dist_pack(uint32_t n, DistArrays in, DistArrays out, float h, int lo, int hi, bool has_lo, bool has_hi, int mode, int nbr_lo_lo,
                      int nbr_hi_hi, const void* pos, DistRec* send_lo, DistRec* send_hi, hipStream_t s) {
    if (n) k_dist_pack<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, in, out, dist_sel(h, lo, hi, has_lo, has_hi, mode, nbr_lo_lo, nbr_hi_hi),
                                                         static_cast<const Sel3*>(pos), send_lo, send_hi);
}
void launch_dist_unpack(uint32_t count, uint32_t base, const DistRec* recv, DistArrays out, uint32_t tag_bits, hipStream_t s) {
    if (count) k_dist_unpack<<<div_up(count, BLOCK), BLOCK, 0, s>>>(count, base, recv, out, tag_bits);
}
void launch_dist_lists(uint32_t n, const uint32_t* gtag, uint32_t* send_lo_idx, uint32_t* send_hi_idx, uint32_t* ghost_lo_idx,
                       uint32_t* ghost_hi_idx, hipStream_t s) {
    if (n) k_dist_lists<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, gtag, send_lo_idx, send_hi_idx, ghost_lo_idx, ghost_hi_idx);
}
void launch_gather_f32(uint32_t count, const uint32_t* idx, const float* src, float* dst, hipStream_t s) {
    if (count) k_gather_idx<float><<<div_up(count, BLOCK), BLOCK, 0, s>>>(count, idx, src, dst);
}
void launch_scatter_f32(uint32_t count, const uint32_t* idx, const float* src, float* dst, hipStream_t s) {
    if (count) k_scatter_idx<float><<<div_up(count, BLOCK), BLOCK, 0, s>>>(count, idx, src, dst);
}
void launch_gather_idx_f4(uint32_t count, const uint32_t* idx, const float4* src, float4* dst, hipStream_t s) {
    if (count) k_gather_idx<float4><<<div_up(count, BLOCK), BLOCK, 0, s>>>(count, idx, src, dst);
}
void launch_scatter_idx_f4(uint32_t count, const uint32_t* idx, const float4* src, float4* dst, hipStream_t s) {
    if (count) k_scatter_idx<float4><<<div_up(count, BLOCK), BLOCK, 0, s>>>(count, idx, src, dst);
}

}  // namespace salva
