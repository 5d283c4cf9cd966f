// nbr_loops.h — per-particle neighbour iteration over the sliced-ELL contact lists.
//
// One lane owns one particle (gather formulation: contacts are directed, contacts.rs:40-55, so no atomics are
// needed for fluid quantities).  Entry k of lane l of wave s lives at nbr[slice_off[s] + 64 k + l]: the 64
// lanes of a wave read one contiguous 256-byte line per k.  The index stream is prefetched one entry ahead
// so the dependent gather of iteration k overlaps the index load of k+1.
#pragma once
#include "common.h"
#include "device_types.h"

namespace salva {

template <typename F>
__device__ __forceinline__ void for_each_nbr(const uint32_t* __restrict__ nbr, const uint64_t* __restrict__ slice_off,
                                             uint32_t i, uint32_t cnt, F&& f) {
    if (cnt == 0) return;
    const uint32_t* __restrict__ p = nbr + slice_off[i / WAVE] + (i & (WAVE - 1));
    uint32_t jn = p[0];
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t j = jn;
        if (k + 1 < cnt) jn = p[(size_t)(k + 1) * WAVE];
        f(j);
    }
}

template <typename F>
__device__ __forceinline__ void for_each_ff(const StepCtx& c, uint32_t i, F&& f) {
    for_each_nbr(c.nbr_ff, c.slice_ff, i, c.nff[i], f);
}
template <typename F>
__device__ __forceinline__ void for_each_fb(const StepCtx& c, uint32_t i, F&& f) {
    if (c.nb == 0) return;
    for_each_nbr(c.nbr_fb, c.slice_fb, i, c.nfb[i], f);
}

// Per-model block reduction of a per-particle error term into c.partials[block][model]
// (par_reduce_sum!, lib.rs:75-83; the per-fluid average is taken by k_finalize_error).
__device__ __forceinline__ void reduce_error(const StepCtx& c, unsigned blk, float err, uint32_t mi, bool active,
                                             float* red) {
    if (c.nmodels == 1) {
        const float s = block_sum(active ? err : 0.0f, red);
        if (threadIdx.x == 0) c.partials[blk] = s;
    } else {
        for (uint32_t m = 0; m < c.nmodels; ++m) {
            const float s = block_sum((active && mi == m) ? err : 0.0f, red);
            if (threadIdx.x == 0) c.partials[(size_t)blk * c.nmodels + m] = s;
        }
    }
}

// Boundary::apply_force (boundary.rs:62-67): forces accumulate in canonical boundary order.
__device__ __forceinline__ void apply_boundary_force(const StepCtx& c, uint32_t jb, uint32_t bmodel, float fx, float fy,
                                                     float fz) {
    if (c.bforce == nullptr || !c.bwants[bmodel]) return;
    float* f = reinterpret_cast<float*>(&c.bforce[c.bperm[jb]]);
    atomicAdd(f + 0, fx);
    atomicAdd(f + 1, fy);
    atomicAdd(f + 2, fz);
}

}  // namespace salva
