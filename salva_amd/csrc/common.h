// common.h — host/device utilities shared by the libsalva_hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <type_traits>
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

// The kernel sources (dfsph.hip, iisph.hip, forces.hip, visc.hip) are compiled twice: as they are (namespace salva, the cubic
// spline everywhere) and with -DSALVA_OTHER_KERNELS into namespace salva_ok, where sph_math.h also knows the reference's
// other kernels.  Every launcher that ends in a kernel evaluation starts with SALVA_OK_DISPATCH: a world whose solver was
// created with a non-default KernelDensity / KernelGradient is handed to the second compilation, and the default build's
// kernels carry no trace of the choice (no branch, no extra register).
#ifdef SALVA_OTHER_KERNELS
#define SALVA_KNS salva_ok
#define SALVA_OK_DISPATCH(fn, c, ...)
#else
#define SALVA_KNS salva
#define SALVA_OK_DISPATCH(fn, c, ...)                 \
    do {                                              \
        if (((c).sc.kd | (c).sc.kg) != 0) {           \
            salva_ok::fn(c, ##__VA_ARGS__);           \
            return;                                   \
        }                                             \
    } while (0)
#endif

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------------
// Device helpers
// ---------------------------------------------------------------------------------------------------

// MI355X dispatches workgroup b to XCD b % 8, each XCD with a private 4 MiB L2.  Workgroup -> slot mapping of the tile
// kernels: the XCDs take turns at groups of 2^(lg-1) consecutive slots (consecutive slots are neighbouring tiles, so a
// group's halo lines meet in that XCD's L2), which spreads every part of the fluid evenly over the eight XCDs.  One
// contiguous eighth per XCD (rounds 1-2) left the XCD with the fluid's half-empty +x face idle early: 4-6 % per
// kernel at 10^6 and at 8 x 10^6; no remapping at all matches the grouped kernels' times but costs the apply kernels
// their locality (3.66 vs 3.48 ms per settled step; profiles/r03_experiments/r03lm_xcd_groups.log).  Numbering the
// slots so that the sparse tiles run last (a shorter tail in theory) and 4x4x4-tile bricks (more L2 reuse in theory)
// were both measured slower (r03o_heavy_first*, r03k_brick_order*).  Bijective for any grid size (the tail that does not fill 8 whole
// groups maps to itself).  Placement only affects speed, never results.  lg = 0: off.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nblocks, unsigned lg) {
    if (lg == 0u) return b;
    const unsigned sh = lg - 1u, xcd = b & 7u, rank = b >> 3, span = 8u << sh;
    if (b >= (nblocks & ~(span - 1u))) return b;
    return ((((rank >> sh) << 3) | xcd) << sh) | (rank & ((1u << sh) - 1u));
}

// Wave-wide reductions by DPP: four row-local steps (quad swaps, half-row mirror, row mirror — after them every lane of a
// 16-lane row holds the row's result), then the four rows through v_readlane.  `__shfl_xor` compiles to ds_bpermute_b32 — a
// round trip through the LDS crossbar per step, six dependent ones per reduction, ~36 in k_pred_density: ~1.5 k cycles of the
// ~3 k a tile spends outside its pair loop after the staging barrier (profiles/r03_experiments).  Fixed order: deterministic.
// Every lane must call; every lane gets the result.
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce_dpp(T v, Op op) {
    static_assert(sizeof(T) == 4, "32-bit values");
    auto dpp = [](T x, auto ctrl) {
        int i;
        __builtin_memcpy(&i, &x, 4);
        i = __builtin_amdgcn_update_dpp(i, i, decltype(ctrl)::value, 0xf, 0xf, false);
        T r;
        __builtin_memcpy(&r, &i, 4);
        return r;
    };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
    v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
    v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));  // row_mirror
    int i;
    __builtin_memcpy(&i, &v, 4);
    const int i0 = __builtin_amdgcn_readlane(i, 0), i1 = __builtin_amdgcn_readlane(i, 16), i2 = __builtin_amdgcn_readlane(i, 32),
              i3 = __builtin_amdgcn_readlane(i, 48);
    T r0, r1, r2, r3;
    __builtin_memcpy(&r0, &i0, 4); __builtin_memcpy(&r1, &i1, 4); __builtin_memcpy(&r2, &i2, 4); __builtin_memcpy(&r3, &i3, 4);
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce_dpp(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return wave_reduce_dpp(v, [](unsigned a, unsigned b) { return a > b ? a : b; }); }
__device__ __forceinline__ int wave_min_i32(int v) { return wave_reduce_dpp(v, [](int a, int b) { return a < b ? a : b; }); }
__device__ __forceinline__ int wave_max_i32(int v) { return wave_reduce_dpp(v, [](int a, int b) { return a > b ? a : b; }); }

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
