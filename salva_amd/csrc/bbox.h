// bbox.h — atomic-free block reduction of cell bounding boxes (min xyz / max xyz as int32).
// Device-scope atomics on six shared words serialise at ~13 ns each (measured: 1 ms for 10^6 particles); instead
// every block writes its six partial bounds and a one-block kernel folds them.
#pragma once
#include <climits>

#include "common.h"

namespace salva {
#ifdef __HIPCC__
// mn/mx: per-thread bounds (INT_MAX / INT_MIN when the thread has no point).  red: 6 * (blockDim/64) ints of LDS.
__device__ __forceinline__ void block_bbox_store(int (&mn)[3], int (&mx)[3], int* red, int32_t* out6) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min_i32(mn[a]); mx[a] = wave_max_i32(mx[a]); }
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[6 * wid + a] = mn[a]; red[6 * wid + 3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        int v = red[a];
        for (int k = 1; k < nw; ++k) v = (a < 3) ? min(v, red[6 * k + a]) : max(v, red[6 * k + a]);
        out6[a] = v;
    }
    __syncthreads();
}
#endif
}  // namespace salva
