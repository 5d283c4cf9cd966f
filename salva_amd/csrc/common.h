// common.h — host/device utilities shared by the libsalva_hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace salva {

constexpr int WAVE = 64;     // gfx950 wavefront
constexpr int BLOCK = 256;   // 4 waves per workgroup, one per SIMD

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define SALVA_HIP_CHECK(expr)                                                                            \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            char _buf[512];                                                                              \
            snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                     __LINE__);                                                                          \
            throw ::salva::HipError(-1, _buf);                                                           \
        }                                                                                                \
    } while (0)

// Growable device buffer (never shrinks; contents are NOT preserved across growth unless asked).
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // returns true if the buffer was reallocated
    bool ensure(size_t n, hipStream_t stream = nullptr, bool preserve = false, float slack = 1.0f) {
        if (n <= cap) return false;
        size_t ncap = (size_t)((double)n * slack);
        if (ncap < n) ncap = n;
        if (ncap < 64) ncap = 64;
        T* np = nullptr;
        SALVA_HIP_CHECK(hipMalloc((void**)&np, ncap * sizeof(T)));
        if (preserve && p && cap) {
            SALVA_HIP_CHECK(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, stream));
            SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = ncap;
        return true;
    }
    size_t bytes() const { return cap * sizeof(T); }
};

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------------
// Device helpers
// ---------------------------------------------------------------------------------------------------

// MI355X dispatches workgroup b to XCD b % 8, each XCD with a private 4 MiB L2.  Particles are sorted by
// cell, so neighbouring workgroups gather from overlapping cache lines: give each XCD one contiguous
// eighth of the blocks (bijective for any grid size).  Placement only affects speed, never results.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nblocks, int enabled) {
    if (!enabled) return b;
    const unsigned xcd = b & 7u, rank = b >> 3;
    const unsigned q = nblocks >> 3, r = nblocks & 7u;
    const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + rank;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned t = (unsigned)__shfl_xor((int)v, o, 64);
        v = v > t ? v : t;
    }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o, 64); v = v < t ? v : t; }
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o, 64); v = v > t ? v : t; }
    return v;
}

// Deterministic block sum (fixed tree): every thread must call; result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float* lds /* >= BLOCK/WAVE floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    float r = 0.0f;
    if (threadIdx.x == 0) {
        const int nw = blockDim.x / WAVE;
        for (int k = 0; k < nw; ++k) r += lds[k];
    }
    __syncthreads();
    return r;
}

__device__ __forceinline__ float3 f3(const float4& a) { return make_float3(a.x, a.y, a.z); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ax * bx + ay * by + az * bz;
}
#endif  // __HIPCC__

}  // namespace salva
