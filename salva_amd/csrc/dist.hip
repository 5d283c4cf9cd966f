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
__global__ __launch_bounds__(BLOCK) void k_dist_flags(uint32_t n, const float4* __restrict__ posm, const uint32_t* __restrict__ gtag,
                                                      float h, int lo, int hi, int has_lo, int has_hi, int mode, int nbr_lo_lo,
                                                      int nbr_hi_hi, Sel3* __restrict__ sel, uint32_t* flags) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Sel3 s{0, 0, 0, 0};
    const bool ghost = (gtag[i] & GTAG_GHOST) != 0;
    if (mode == 2) s.keep = 1;
    if (!ghost) {
        bool bad = false;
        const int cx = cell_coord(posm[i].x, h, bad);
        if (mode == 1) {
            // a leaver goes to the adjacent rank; flag 4 if even that rank's slab ([nbr_lo_lo, lo - 1] / [hi + 1, nbr_hi_hi],
            // open ends = INT_MIN / INT_MAX) does not hold its cell: it would be owned where nobody mirrors it
            if (has_lo && cx < lo) { s.lo = 1; if (cx < nbr_lo_lo) atomicOr(flags, 4u); }
            else if (has_hi && cx > hi) { s.hi = 1; if (cx > nbr_hi_hi) atomicOr(flags, 4u); }
            else s.keep = 1;
        } else {
            // two planes per face (GHOST_PLANES); <= / >= : an open-ended first / last slab may hold particles beyond its planes
            if (has_lo && cx <= lo + (GHOST_PLANES - 1)) s.lo = 1;
            if (has_hi && cx >= hi - (GHOST_PLANES - 1)) s.hi = 1;
        }
    }
    sel[i] = s;
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

// Move / copy the selected particles.  `pos` is the exclusive scan of `sel` (pos[n] = totals).
__global__ __launch_bounds__(BLOCK) void k_dist_pack(uint32_t n, DistArrays in, DistArrays out, const Sel3* __restrict__ sel,
                                                     const Sel3* __restrict__ pos, int mode, DistRec* __restrict__ send_lo,
                                                     DistRec* __restrict__ send_hi) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Sel3 s = sel[i], p = pos[i];
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
size_t dist_scan_temp_bytes(uint32_t n) {
    size_t b = 0;
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, b, (const Sel3*)nullptr, (Sel3*)nullptr, hipcub::Sum(), Sel3{0, 0, 0, 0}, (int)n);
    return b;
}
size_t dist_sel_bytes(uint32_t n) { return (size_t)(n + 1) * sizeof(Sel3); }

void launch_dist_select(uint32_t n, const float4* posm, const uint32_t* gtag, float h, int lo, int hi, bool has_lo, bool has_hi,
                        int mode, int nbr_lo_lo, int nbr_hi_hi, void* sel, void* pos, void* temp, size_t temp_bytes, uint32_t* flags,
                        uint32_t totals_host[3], hipStream_t s) {
    Sel3* se = static_cast<Sel3*>(sel);
    Sel3* po = static_cast<Sel3*>(pos);
    SALVA_HIP_CHECK(hipMemsetAsync(se + n, 0, sizeof(Sel3), s));
    if (n) k_dist_flags<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, posm, gtag, h, lo, hi, has_lo ? 1 : 0, has_hi ? 1 : 0, mode, nbr_lo_lo, nbr_hi_hi, se, flags);
    SALVA_HIP_CHECK(hipcub::DeviceScan::ExclusiveScan(temp, temp_bytes, se, po, hipcub::Sum(), Sel3{0, 0, 0, 0}, (int)(n + 1), s));
    Sel3 tot;
    SALVA_HIP_CHECK(hipMemcpyAsync(&tot, po + n, sizeof(Sel3), hipMemcpyDeviceToHost, s));
    SALVA_HIP_CHECK(hipStreamSynchronize(s));
    totals_host[0] = tot.keep; totals_host[1] = tot.lo; totals_host[2] = tot.hi;
}
void launch_dist_pack(uint32_t n, DistArrays in, DistArrays out, const void* sel, const void* pos, int mode, DistRec* send_lo,
                      DistRec* send_hi, hipStream_t s) {
    if (n) k_dist_pack<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, in, out, static_cast<const Sel3*>(sel), static_cast<const Sel3*>(pos), mode,
                                                         send_lo, send_hi);
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
