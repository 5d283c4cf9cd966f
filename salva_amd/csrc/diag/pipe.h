// pipe.h — persistent, double-buffered tile pipeline: the second execution skeleton of the neighbour kernels.
//
// Why: with one workgroup per tile (tile.h) a tile's life is a chain of dependent phases — descriptor
// and slot-table fetch, index-gather of the halo, LDS write, barrier, neighbour loops, reduction — and the 66 KB of
// LDS a tile needs admit two tiles per CU, which start together and stay in phase: both wait on memory, then both
// fight for the VALU/LDS.  Measured: 60 % of k_pred_density's time is outside the neighbour loop.
//
// Here ONE workgroup per CU walks a sequence of tiles and keeps two halo buffers in LDS.  While the waves compute
// tile k out of buffer k&1, the halo of tile k+1 streams into the other buffer by LDS-DMA
// (global_load_lds_dwordx4: per-lane gather address, lane-linear LDS destination — exactly the shape of a slot
// table row; no VGPR round trip, no ds_write pass), the own-particle records and list heads of tile k+1 are on
// their way into a second register set, and the slot-table row of tile k+2 is being fetched.  One barrier per tile.
// Every memory latency of a tile is therefore hidden behind the arithmetic of its predecessor; what remains is the
// neighbour loop itself.  Requires the strided slot tables (StepCtx::halo_stride > 0) and 2 x halo in 160 KiB; the
// launchers fall back to the one-tile-per-workgroup kernels otherwise.
#pragma once
#include "tile.h"

namespace salva {

constexpr int PIPE_MAX_WAVES = 12;
constexpr int PIPE_MAX_THREADS = PIPE_MAX_WAVES * WAVE;
constexpr int PIPE_PRE = 4;  // slot-table dwords prefetched per thread (covers halos up to 4 x threads slots)
constexpr uint32_t LDS_BYTES_PER_CU = 160u * 1024u;

// Launch shape of the pipeline kernels of one step (host side).
struct PipeCfg {
    uint32_t threads = 0, num_cus = 256, nlaunch = 0;
    uint32_t scap = 0, sbcap = 0;  // slots per staged fluid / boundary array (multiples of 64: DMA granularity)
    bool enabled = false;
    // bytes of dynamic LDS for a pass that stages nf4 float4 + nf1 float fluid arrays and nb4 float4 boundary arrays
    uint32_t bytes(uint32_t nf4, uint32_t nf1, uint32_t nb4, bool err, uint32_t nbuf = 2u) const {
        return nbuf * (scap * (16u * nf4 + 4u * nf1) + sbcap * 16u * nb4) +
               (err ? 2u * PIPE_MAX_WAVES * MAX_MODELS * (uint32_t)sizeof(float) : 0u) + MAX_MODELS * (uint32_t)sizeof(float);
    }
    bool fits(uint32_t nf4, uint32_t nf1, uint32_t nb4, bool err, uint32_t nbuf = 2u) const {
        return enabled && bytes(nf4, nf1, nb4, err, nbuf) <= LDS_BYTES_PER_CU;
    }
    // persistent grid: as many workgroups as stay resident (LDS and the 32 wave slots of a CU), never more than tiles
    uint32_t grid(uint32_t lds_bytes) const {
        uint32_t per_cu = LDS_BYTES_PER_CU / (lds_bytes ? lds_bytes : 1u);
        const uint32_t by_waves = 2048u / (threads ? threads : 64u);
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1u) per_cu = 1u;
        const uint32_t g = num_cus * per_cu;
        return g < nlaunch ? g : nlaunch;
    }
};

#ifdef __HIPCC__

// Tiles of one workgroup: the XCD that runs workgroup b (b % 8, observed) owns one contiguous eighth of the slots
// (as xcd_block does for the one-tile kernels) and its workgroups sweep it side by side, so that the tiles in flight
// on an XCD at any time are neighbours and share halo lines in that XCD's L2.
struct PipeSeq {
    uint32_t first, step, end;  // slots first, first + step, ... < end
    __device__ __forceinline__ void init(const StepCtx& c) {
        const uint32_t G = gridDim.x, b = blockIdx.x, nl = c.nlaunch;
        uint32_t f = b, st = G, e = nl;
        if (c.xcd && G >= 8u) {
            const uint32_t xcd = b & 7u, q = nl >> 3, rem = nl & 7u;
            const uint32_t base = (xcd < rem) ? xcd * (q + 1u) : rem * (q + 1u) + (xcd - rem) * q;
            f = base + (b >> 3);
            st = (G - xcd + 7u) >> 3;
            e = base + q + (xcd < rem ? 1u : 0u);
        }
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
        step = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
        end = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
    }
    __device__ __forceinline__ bool has(uint32_t k) const { return first + k * step < end; }
    __device__ __forceinline__ uint32_t slot(uint32_t k) const { return first + k * step; }
};

struct PipeInfo { uint32_t slot, own_begin, own_end, slice_base, S, SB; };

// What a pass body sees of the tile it works on.
struct PipeView {
    const float4* Lp;   // staged float4 array 0 (positions + mass)
    const float4* Lw;   // staged float4 array 1 (if any)
    const float* Lk;    // staged float array (if any)
    const float4* Bp;   // boundary (x, y, z, volume)
    const float4* Bv;   // boundary (velocity, model id) — only when the pass stages it
    uint32_t S, SB, slot, own_begin, own_end, slice_base;
    uint64_t hboff;     // offset of this tile's row in bhalo_src
    const float* rho0;  // density0 of every fluid, in LDS (a global load inside a pass body would queue behind the next
                        // tile's DMA: vmcnt retires in order)
};
__device__ __forceinline__ float rho0_of(const StepCtx& c, const PipeView& t, uint32_t model) {
    return (c.nmodels == 1) ? c.rho0_single : t.rho0[model];
}
__device__ __forceinline__ uint32_t boundary_sorted_of_slot(const StepCtx& c, const PipeView& t, uint32_t slot) {
    return c.bhalo_src[t.hboff + slot];
}

// NF4 / NF1: number of float4 / float fluid arrays staged per halo slot; NB4: boundary float4 arrays (1: positions,
// 2: + velocities); ERR: the pass reduces a per-fluid error (TileErr protocol of tile.h, one partial per slot).
//   load_own(i, gslice) -> Own   : the lane's own-particle record (plain loads; prefetched one tile ahead)
//   body(own, i, gslice, active, view, E) : the neighbour sums of one particle (E: TileErr&, unused unless ERR)
template <int NF4, int NF1, int NB4, bool ERR, bool DOUBLE, typename Own, typename LoadOwn, typename Body>
__device__ __forceinline__ void tile_pipeline(const StepCtx& c, uint32_t scap, uint32_t sbcap, const float4* __restrict__ f4a,
                                              const float4* __restrict__ f4b, const float* __restrict__ f1, LoadOwn&& load_own,
                                              Body&& body) {
    PipeSeq seq;
    seq.init(c);
    if (!seq.has(0)) return;
    unsigned char* const pool = tile_smem;
    const uint32_t bufbytes = scap * (16u * NF4 + 4u * NF1) + sbcap * 16u * NB4;
    constexpr uint32_t NBUF = DOUBLE ? 2u : 1u;
    float(*const errtab)[PIPE_MAX_WAVES][MAX_MODELS] =
        reinterpret_cast<float(*)[PIPE_MAX_WAVES][MAX_MODELS]>(pool + NBUF * bufbytes);
    float* const rho0s = reinterpret_cast<float*>(pool + NBUF * bufbytes + (ERR ? 2u * PIPE_MAX_WAVES * MAX_MODELS * 4u : 0u));
    if (threadIdx.x < c.nmodels) rho0s[threadIdx.x] = c.rho0_tab[threadIdx.x];  // visible after the first barrier
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & (WAVE - 1);
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / WAVE)), nw = nt / WAVE;

    // per-slot sizes: one uint4 (k_tile_halo_fill), fetched as an ordinary vector load one tile ahead and made
    // wave-uniform when it is consumed
    auto decode = [&](uint32_t slot, const uint4& raw) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.w);
        return PipeInfo{slot, (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.y),
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.z), w & 0xffffu, w >> 16};
    };
    struct Idx { uint32_t f[PIPE_PRE]; uint32_t b; };
    auto load_idx = [&](uint32_t slot) {
        Idx x;
        const uint32_t* __restrict__ src = c.halo_src + (size_t)slot * c.halo_stride;
#pragma unroll
        for (int k = 0; k < PIPE_PRE; ++k) {
            const uint32_t s = tid + (uint32_t)k * nt;
            x.f[k] = (s < c.halo_stride) ? src[s] : 0u;
        }
        x.b = (NB4 > 0 && tid < c.bhalo_stride) ? c.bhalo_src[(size_t)slot * c.bhalo_stride + tid] : 0u;
        return x;
    };
    struct Bufs { float4 *Lp, *Lw; float* Lk; float4 *Bp, *Bv; };
    auto bufs = [&](uint32_t b) {
        unsigned char* B = pool + b * bufbytes;
        Bufs r;
        r.Lp = reinterpret_cast<float4*>(B);
        r.Lw = r.Lp + (NF4 > 1 ? scap : 0u);
        r.Lk = reinterpret_cast<float*>(B + scap * 16u * NF4);
        r.Bp = reinterpret_cast<float4*>(B + scap * (16u * NF4 + 4u * NF1));
        r.Bv = r.Bp + (NB4 > 1 ? sbcap : 0u);
        return r;
    };
    // start the halo of tile `ti` towards buffer b: every wave DMAs the 64-slot chunks wv, wv + nw, ...
    auto issue = [&](const PipeInfo& ti, const Idx& x, uint32_t b) {
        const Bufs L = bufs(b);
        auto chunk = [&](uint32_t s0, uint32_t g) {
            glds16(f4a + g, L.Lp + s0);
            if (NF4 > 1) glds16(f4b + g, L.Lw + s0);
            if (NF1 > 0) glds4(f1 + g, L.Lk + s0);
        };
#pragma unroll
        for (int k = 0; k < PIPE_PRE; ++k) {
            const uint32_t s0 = (wv + (uint32_t)k * nw) * WAVE;
            if (s0 < ti.S) chunk(s0, (s0 + lane < ti.S) ? x.f[k] : ti.own_begin);
        }
        for (uint32_t s0 = (wv + (uint32_t)PIPE_PRE * nw) * WAVE; s0 < ti.S; s0 += nt) {  // halos beyond the prefetch window
            const uint32_t s = s0 + lane;
            chunk(s0, (s < ti.S) ? c.halo_src[(size_t)ti.slot * c.halo_stride + s] : ti.own_begin);
        }
        if (NB4 > 0) {
            for (uint32_t s0 = wv * WAVE, k = 0; s0 < ti.SB; s0 += nt, ++k) {
                const uint32_t s = s0 + lane;
                uint32_t g = 0u;
                if (s < ti.SB) g = (k == 0) ? x.b : c.bhalo_src[(size_t)ti.slot * c.bhalo_stride + s];
                glds16(c.bposv + g, L.Bp + s0);
                if (NB4 > 1) glds16(c.bvel + g, L.Bv + s0);
            }
        }
    };
    // the own-particle record of this wave's first slice of tile `ti` (loaded unconditionally: idle lanes and waves read
    // the tile's first particle, cf. Tile::first_own)
    auto first_own = [&](const PipeInfo& ti) {
        const uint32_t nsl = (ti.own_end - ti.own_begin + WAVE - 1) / WAVE;
        uint32_t i = ti.own_begin + wv * WAVE + lane, gs = ti.slice_base + wv;
        if (!(wv < nsl && i < ti.own_end)) { i = ti.own_begin; gs = ti.slice_base; }
        return load_own(i, gs);
    };
    auto flush_err = [&](uint32_t slot, uint32_t eb) {
        if (tid < c.nmodels) {
            float s = 0.0f;
            for (uint32_t w = 0; w < nw; ++w) s += errtab[eb][w][tid];
            c.partials[(size_t)slot * c.nmodels + tid] = s;
        }
    };

    // Registers loaded one tile ahead are consumed only after the explicit wait at the top of the next iteration.  hipcc
    // does not see that wait: `landed` makes it place its own (then redundant) wait for them at the same spot, instead of
    // a conservative vmcnt(0) at their first use — which would sit behind the DMA issued in between and drain it.
    auto landed = [](auto& v) {
        static_assert(sizeof(v) % 4 == 0, "dword-sized records only");
        uint32_t* d = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (unsigned q = 0; q < sizeof(v) / 4; ++q) asm volatile("" : "+v"(d[q]));
    };

    auto compute = [&](const PipeInfo& cur, const Own& own_cur, uint32_t b, uint32_t eb) {
        const Bufs L = bufs(b);
        PipeView v;
        v.Lp = L.Lp; v.Lw = L.Lw; v.Lk = L.Lk; v.Bp = L.Bp; v.Bv = L.Bv;
        v.S = cur.S; v.SB = cur.SB; v.slot = cur.slot; v.own_begin = cur.own_begin; v.own_end = cur.own_end;
        v.slice_base = cur.slice_base;
        v.hboff = (uint64_t)cur.slot * c.bhalo_stride;
        v.rho0 = rho0s;
        TileErr E;
        E.tab = errtab[eb];
        if (ERR && lane < c.nmodels) E.tab[wv][lane] = 0.0f;
        const uint32_t nsl = (cur.own_end - cur.own_begin + WAVE - 1) / WAVE;
        uint32_t s = wv;
        if (s < nsl) {
            const uint32_t i = cur.own_begin + s * WAVE + lane;
            body(own_cur, i, cur.slice_base + s, i < cur.own_end, v, E);
            s += nw;
        }
        for (; s < nsl; s += nw) {  // tiles with more slices than the workgroup has waves
            const uint32_t i = cur.own_begin + s * WAVE + lane, gs = cur.slice_base + s;
            const bool active = i < cur.own_end;
            const Own p = load_own(active ? i : cur.own_begin, gs);
            body(p, i, gs, active, v, E);
        }
    };

    if (!DOUBLE) {
        // Single buffer, several workgroups per CU: a workgroup's own phases stay serial (DMA -> barrier -> neighbour
        // loops -> barrier), the co-resident workgroups overlap each other as the one-tile kernels do — but a tile
        // costs no workgroup launch, its sizes and slot-table row were fetched during the previous tile's arithmetic,
        // and its halo arrives by DMA: what is exposed per tile is one memory round trip instead of three plus a launch.
        uint4 raw = c.slot_info[seq.slot(0)];
        Idx x = load_idx(seq.slot(0));
        for (uint32_t k = 0;; ++k) {
            const unsigned long long T0 = c.dbg ? __builtin_readcyclecounter() : 0ull;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            landed(raw); landed(x);
            const PipeInfo cur = decode(seq.slot(k), raw);
            // the own-particle record goes out before the barrier: its registers are free as soon as this wave is done
            // with tile k-1, and its latency then overlaps the wait for the slower waves
            Own own_cur = first_own(cur);
            if (k > 0) {
                // everybody is done reading the buffer (tile k-1).  A raw barrier: __syncthreads() would also wait for
                // the loads just issued.
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (ERR) flush_err(seq.slot(k - 1), (k - 1) & 1u);
            }
            const unsigned long long T1 = c.dbg ? __builtin_readcyclecounter() : 0ull;
            issue(cur, x, 0u);
            const unsigned long long T2 = c.dbg ? __builtin_readcyclecounter() : 0ull;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            landed(own_cur);
            const unsigned long long T3 = c.dbg ? __builtin_readcyclecounter() : 0ull;
            __syncthreads();
            const unsigned long long T4 = c.dbg ? __builtin_readcyclecounter() : 0ull;
            const bool has1 = seq.has(k + 1);
            const uint32_t slot_n = has1 ? seq.slot(k + 1) : cur.slot;
            raw = c.slot_info[slot_n];
            x = load_idx(slot_n);
            compute(cur, own_cur, 0u, k & 1u);
            if (c.dbg && lane == 0) {  // per wave: phase stamps of this tile (SALVA_HIP_TILE_TIMING)
                unsigned long long* d = c.dbg + ((size_t)cur.slot * PIPE_MAX_WAVES + wv) * 8;
                d[0] = T0; d[1] = T1; d[2] = T2; d[3] = T3; d[4] = T4; d[5] = __builtin_readcyclecounter();
                d[6] = cur.S; d[7] = cur.own_end - cur.own_begin;
            }
            if (!has1) {
                if (ERR) { __syncthreads(); flush_err(cur.slot, k & 1u); }
                return;
            }
        }
    }

    uint4 raw = c.slot_info[seq.slot(0)];
    Idx x = load_idx(seq.slot(0));
    landed(raw); landed(x);
    PipeInfo cur = decode(seq.slot(0), raw);
    issue(cur, x, 0u);
    Own own_cur = first_own(cur);
    bool has1 = seq.has(1);
    uint32_t slot_n = has1 ? seq.slot(1) : cur.slot;
    raw = c.slot_info[slot_n];
    x = load_idx(slot_n);
    uint32_t k = 0;
    for (;; ++k) {
        // every load this wave has in flight belongs to tile k (halo DMA, own record) or describes tile k+1 (sizes, slot
        // table row): wait for them, then meet the other waves — their DMA shares have landed too, and nobody still reads
        // the other buffer (tile k-1)
        const unsigned long long T0 = c.dbg ? __builtin_readcyclecounter() : 0ull;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed(own_cur); landed(raw); landed(x);
        const unsigned long long T1 = c.dbg ? __builtin_readcyclecounter() : 0ull;
        __syncthreads();
        const unsigned long long T2 = c.dbg ? __builtin_readcyclecounter() : 0ull;
        if (ERR && k > 0) flush_err(seq.slot(k - 1), (k - 1) & 1u);
        const PipeInfo nxt = decode(slot_n, raw);
        const bool has2 = seq.has(k + 2);
        if (has1) issue(nxt, x, (k + 1) & 1u);
        Own own_nxt = first_own(nxt);  // (re-reads tile k's record when there is no tile k+1)
        const uint32_t slot_nn = has2 ? seq.slot(k + 2) : slot_n;
        raw = c.slot_info[slot_nn];
        x = load_idx(slot_nn);
        const unsigned long long T3 = c.dbg ? __builtin_readcyclecounter() : 0ull;
        compute(cur, own_cur, k & 1u, k & 1u);
        if (c.dbg && lane == 0) {
            unsigned long long* d = c.dbg + ((size_t)cur.slot * PIPE_MAX_WAVES + wv) * 8;
            d[0] = T0; d[1] = T1; d[2] = T2; d[3] = T3; d[4] = T3; d[5] = __builtin_readcyclecounter();
            d[6] = cur.S; d[7] = cur.own_end - cur.own_begin;
        }
        if (!has1) break;
        cur = nxt; own_cur = own_nxt; slot_n = slot_nn; has1 = has2;
    }
    if (ERR) {
        __syncthreads();
        flush_err(cur.slot, k & 1u);
    }
}

#endif  // __HIPCC__
}  // namespace salva
