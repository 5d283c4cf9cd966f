// tile.h — LDS-staged cell tiles: the execution skeleton of every neighbour kernel.
//
// Why: a one-lane-per-particle gather straight from global memory issues ~25 M L2 requests per pass at 10^6
// particles (measured: L1 hit rate 45 %, L2 request rate ~80 % of the tag-lookup ceiling, VALU 20-40 % busy —
// profiles/r01a_v1_gather).  Here one workgroup owns one tile of 4x4x4 grid cells; it copies the particles of the
// 6x6x6 halo box (own tile + one cell all round, the interaction range being exactly one cell, contacts.rs:164-165)
// into LDS with coalesced index-gathers, and the per-particle neighbour loops then read 16-byte records from LDS by
// slot.  Neighbour lists store 16-bit LDS slots (two per dword), so a pass streams 2 B per contact instead of 4.
//
// Particle order = tile-major cell key (tile linear index * 64 + cell-in-tile), x slowest.  A tile's own particles
// are one contiguous index range; its halo is at most 216 cells, each a contiguous range (cell table with
// lower-bound semantics).  The tile grid is anchored at the minimum corner of the occupied cells' bounding box; the
// boundary particles are sorted on the same grid, so fluid and boundary tables agree on tile membership.
//
// What bounds these kernels and what was tried: DESIGN.md §3.3.
#pragma once
#include "common.h"
#include "device_types.h"

namespace salva {

#ifndef SALVA_TX
#define SALVA_TX 4
#endif
constexpr int TX = SALVA_TX, TY = 4, TZ = 4;           // cells per tile
constexpr int TCELLS = TX * TY * TZ;                   // 64
constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;   // halo box
constexpr int HCELLS = HX * HY * HZ;                   // 216
constexpr int TILE_MAX_WAVES = 12;                      // workgroup = one wave per 64-particle slice of the average non-empty
constexpr int TILE_MAX_THREADS = TILE_MAX_WAVES * WAVE; // tile, clamped to [4, 12] waves (a tile holds 512 particles on the 2r
                                                       // lattice: 8 waves; fuller tiles loop over their extra slices)

__host__ __device__ inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// ELL layout of a slice's list block (cap dwords per lane, cap a multiple of 4): dword q of lane l lives at
// ((q >> 2) * WAVE + l) * 4 + (q & 3) — four consecutive dwords of a lane are contiguous, so a list head goes to registers
// with 16-byte loads: 5 VMEM instructions per lane instead of 20, and the issue of those (~70 cycles each, measured) sits
// on every tile's critical path before the staging barrier.  With p = block + 4 * lane, dword q is p[ellq(q)].
__host__ __device__ inline uint32_t ellq(uint32_t q) { return (q >> 2) * (4u * (uint32_t)WAVE) + (q & 3u); }

constexpr uint32_t TILE_TABLE_BYTES = 4 * (2 * (HCELLS + 1) + 2 * HCELLS) + 16;

// Launch geometry / dynamic-LDS budget of the tile kernels of one step: staged arrays are sized for the largest
// halo, the workgroup for the fullest tile.
struct TileLds {
    uint32_t max_halo_fluid = 0, max_halo_boundary = 0, threads = 4 * WAVE;
    // largest (fluid halo padded to 64) + (boundary halo) of any ONE tile: the tile with the fullest fluid halo sits inside
    // the fluid and the tile with the most boundary particles at a wall, so this is well below the sum of the two maxima —
    // 10 KB of LDS per workgroup in the bench scene, the difference between one and two resident tiles per CU for the
    // 36-byte-per-slot force kernels
    uint32_t max_sum = 0;
    // largest (fluid halo) + (boundary halo) of any one tile WITHOUT padding (0: unknown — a speculative pass): the slot count of
    // the plane layouts (stage_p3), which are filled through registers and need no 64-slot granularity
    uint32_t max_raw = 0;
    uint32_t ds_level = 0;  // SALVA_HIP_DS_LEVEL (pairs.h pick_ds*): 0 in production
    uint32_t raw_slots() const { return max_raw ? max_raw : max_halo_fluid + max_halo_boundary; }
    uint32_t sum_slots() const {
        const uint32_t worst = ((max_halo_fluid + 63u) & ~63u) + max_halo_boundary;
        return max_sum ? (max_sum < worst ? max_sum : worst) : worst;
    }
    uint32_t bytes(uint32_t bytes_per_fluid_slot, uint32_t bytes_per_boundary_slot, uint32_t narrays,
                   bool with_cell_tables = false) const {
        // (staged fluid arrays are filled by LDS-DMA in chunks of 64 slots: sized to the next multiple of 64)
        const uint32_t separate = ((max_halo_fluid + 63u) & ~63u) * bytes_per_fluid_slot + max_halo_boundary * bytes_per_boundary_slot;
        const uint32_t wide = bytes_per_fluid_slot > bytes_per_boundary_slot ? bytes_per_fluid_slot : bytes_per_boundary_slot;
        const uint32_t joint = sum_slots() * wide;
        return (with_cell_tables ? TILE_TABLE_BYTES : 0u) + (joint < separate ? joint : separate) + 16u * narrays;
    }
};

// Fixed LDS layouts of the four DFSPH solver kernels (dfsph.hip): the second staged array sits at a COMPILE-TIME distance
// from the first, so both LDS reads of a contact take their address from one VGPR (the list entry << 4) plus an immediate
// offset — one VALU instruction per contact instead of three.  Two instantiations: 2448 slots (two such workgroups still
// share a CU's 160 KB) and 3984 (the 16-bit offset field ends at 65535); fuller halos take the runtime-distance
// instantiation (DS = 0).
#ifndef SALVA_FIXED_DS_SMALL
#define SALVA_FIXED_DS_SMALL 2448  // (an odd multiple of 16 slots = of 256 bytes, see P3_DS_THREE below: k_iisph_dij_pj 49.7 -> 43.5 us against 2432)
#endif
constexpr uint32_t FIXED_DS_SMALL = SALVA_FIXED_DS_SMALL, FIXED_DS_LARGE = 3984;
// Plane layout of the evaluate kernels (stage_p3 below): plane k starts at LDS byte k * 8 * DS whatever the halo's size, and the
// workgroup's LDS ends behind the slots it really uses — so ONE instantiation serves every halo of up to DS slots and costs
// 16 DS + 8 n bytes (+ the boundary halo, 16 bytes per slot and array).  2080: three workgroups per CU (3 x 54.6 KB) with up to
// ~330 boundary slots beside a full fluid halo — the compressed column of the bench scene, where the 50-iteration solves live,
// holds 1940-2000 fluid slots per tile; 3328: two; 4095: the offset field of ds_read_b64 ends at 65535.
// The plane distance must be an ODD multiple of 256 bytes (DS an odd multiple of 32): measured, k_pred_density takes 40.2 us with
// planes 69 x 256 bytes apart and 46.4 us with 64 x 256 or 66 x 256 (profiles/r04_experiments/r04c_plane_distance.log) — the three
// ds_read_b64 of a contact, which share their base address, then land in three different 256-byte phases of a 1 KB period
// (0, 256 or 768, 512); at an even multiple two or all three coincide and the reads serialise.  (MI355X_MICROARCH.md documents
// the 64 banks of one 256-byte row, not what lies above them.)
#ifndef SALVA_P3_DS3
#define SALVA_P3_DS3 2080
#endif
constexpr uint32_t P3_DS_THREE = SALVA_P3_DS3, P3_DS_TWO = 3360, P3_DS_ONE = 4064;  // evaluate kernels: fluid halo slots
#ifndef SALVA_P2_DS3
#define SALVA_P2_DS3 2464
#endif
constexpr uint32_t P2_DS_THREE = SALVA_P2_DS3;                               // apply kernels: fluid + boundary halo slots
static_assert((P3_DS_THREE % 64 == 32 && P3_DS_TWO % 64 == 32 && P3_DS_ONE % 64 == 32 && P2_DS_THREE % 64 == 32) || SALVA_P3_DS3 != 2080 || SALVA_P2_DS3 != 2464,
              "plane distances: odd multiples of 256 bytes");
// Parts of a split tile (StepCtx::split_s).  A part owns the cells ux in [ux0, ux0 + len) of its tile — one contiguous particle
// range, the cell key being x-major inside a tile — and stages the halo planes hx in [ux0, ux0 + len + 2) only.  Code = ux0 | (len - 1) << 2,
// 4 bits; the whole tile is ux0 = 0, len = TX.  It rides in the top bits of a tile_ids entry and in slot_desc.w.
constexpr uint32_t TILE_PART_SHIFT = 28;
constexpr uint32_t TILE_PART_WHOLE = (uint32_t)(TX - 1) << 2;
__host__ __device__ inline uint32_t tile_part_code(uint32_t ux0, uint32_t len) { return ux0 | ((len - 1u) << 2); }
// the fluid halo beyond which a tile does not fit the three-tiles-per-CU plane layouts (P3_DS_THREE)
constexpr uint32_t TILE_SPLIT_S = P3_DS_THREE;
// the sparse class (StepCtx::slot_order): a slot with at most one slice of own particles and a halo this small
constexpr uint32_t TILE_TINY_S = 192, TILE_TINY_SB = 64, TILE_TINY_THREADS = 64;
constexpr uint32_t TILE_ERR_BYTES = 12u * 32u * 4u;  // TileErr table (TILE_MAX_WAVES x MAX_MODELS floats), carved from the pool

#ifdef __HIPCC__

// hipFuncSetAttribute costs tens of microseconds of host time: raise a kernel's dynamic-LDS ceiling only when a
// launch actually needs more than was granted before (48 KiB is the default).
void raise_tile_lds_limit(const void* kernel, uint32_t bytes);  // world.hip
template <typename K>
inline void ensure_tile_lds(K kernel, uint32_t bytes) {
    if (bytes > 48u * 1024u) raise_tile_lds_limit(reinterpret_cast<const void*>(kernel), bytes);
}
// One launch per pass — or one per launch class of the step (StepCtx::slot_order): the full slots with the kernel, workgroup and LDS
// the call site asks for; the light ones (StepCtx::nlight) with `kernel_light` — the family's instantiation for its smallest
// compile-time layout, `ds_light` — and the LDS the SAME size expression yields for the light class's bounds; the sparse ones
// (StepCtx::ntiny) with `kernel_tiny` (the run-time-layout instantiation: a compile-time plane distance would pin the LDS request at
// tens of KB), TILE_TINY_THREADS threads and the LDS of a halo of TILE_TINY_S + TILE_TINY_SB slots.  The expression and the kernel
// arguments are re-read under shadowed names: every call site calls them `c`, `L` and (where it has one) `ds`.
inline TileLds tile_tiny_lds(const TileLds& L) {
    TileLds t = L;
    t.max_halo_fluid = TILE_TINY_S; t.max_halo_boundary = L.max_halo_boundary ? TILE_TINY_SB : 0u;
    t.max_sum = ((TILE_TINY_S + 63u) & ~63u) + t.max_halo_boundary; t.max_raw = TILE_TINY_S + t.max_halo_boundary;
    t.threads = TILE_TINY_THREADS;
    return t;
}
// the light class: a halo every kernel family serves from its smallest compile-time layout — the plane layouts' three-per-CU
// distances (P3_DS_THREE fluid slots; P2_DS_THREE fluid + boundary slots) and FIXED_DS_SMALL for both uses of the 16-byte layouts
// (pairs.h pw_slots: padded fluid + boundary; pk_slots: padded fluid + 2 x boundary)
__host__ __device__ inline bool tile_is_light(uint32_t s, uint32_t sb) {
    return s <= P3_DS_THREE && ((s + 63u) & ~63u) + 2u * sb <= FIXED_DS_SMALL;
}
inline TileLds tile_light_lds(const TileLds& L) {
    TileLds t = L;
    t.max_halo_fluid = L.max_halo_fluid < P3_DS_THREE ? L.max_halo_fluid : P3_DS_THREE;
    t.max_halo_boundary = L.max_halo_boundary < FIXED_DS_SMALL / 2u ? L.max_halo_boundary : FIXED_DS_SMALL / 2u;
    t.max_sum = (L.max_sum && L.max_sum < FIXED_DS_SMALL) ? L.max_sum : FIXED_DS_SMALL;
    t.max_raw = (L.max_raw && L.max_raw < FIXED_DS_SMALL) ? L.max_raw : FIXED_DS_SMALL;
    return t;
}
#define SALVA_LAUNCH_TILE_3(kernel, kernel_light, ds_light, kernel_tiny, c, L, lds, s, ...)         \
    do {                                                                                           \
        if ((c).n && (c).nlaunch) {                                                                \
            if (((c).ntiny == 0u && (c).nlight == 0u) || (c).slot_order == nullptr) {              \
                const uint32_t _lds = (lds);                                                       \
                ::salva::ensure_tile_lds(kernel, _lds);                                            \
                kernel<<<(c).nlaunch, (L).threads, _lds, s>>>(__VA_ARGS__);                        \
            } else {                                                                               \
                const uint32_t _ntiny = (c).ntiny, _nlight = (c).nlight;                           \
                const uint32_t _nfull = (c).nlaunch - _nlight - _ntiny;                            \
                ::salva::StepCtx _c2 = (c);                                                        \
                const ::salva::TileLds _lt = ::salva::tile_tiny_lds(L);                            \
                const ::salva::TileLds _ll = ::salva::tile_light_lds(L);                           \
                const unsigned _thr = (L).threads;                                                 \
                if (_nfull) {                                                                      \
                    const uint32_t _lds = (lds);                                                   \
                    ::salva::ensure_tile_lds(kernel, _lds);                                        \
                    _c2.slot_base = 0u;                                                            \
                    { const ::salva::StepCtx c = _c2; kernel<<<_nfull, _thr, _lds, s>>>(__VA_ARGS__); } \
                }                                                                                  \
                if (_nlight) {                                                                     \
                    _c2.slot_base = _nfull;                                                        \
                    const ::salva::TileLds L = _ll;                                                \
                    const uint32_t ds = (ds_light); (void)ds;                                      \
                    const ::salva::StepCtx c = _c2;                                                \
                    const uint32_t _lds = (lds);                                                   \
                    ::salva::ensure_tile_lds(kernel_light, _lds);                                  \
                    kernel_light<<<_nlight, _thr, _lds, s>>>(__VA_ARGS__);                         \
                }                                                                                  \
                if (_ntiny) {                                                                      \
                    _c2.slot_base = _nfull + _nlight;                                              \
                    const ::salva::TileLds L = _lt;                                                \
                    const uint32_t ds = 0u; (void)ds;                                              \
                    const ::salva::StepCtx c = _c2;                                                \
                    const uint32_t _lds = (lds);                                                   \
                    ::salva::ensure_tile_lds(kernel_tiny, _lds);                                   \
                    kernel_tiny<<<_ntiny, TILE_TINY_THREADS, _lds, s>>>(__VA_ARGS__);              \
                }                                                                                  \
            }                                                                                      \
            SALVA_HIP_CHECK(hipGetLastError());                                                    \
        }                                                                                          \
    } while (0)
#define SALVA_LAUNCH_TILE(kernel, c, L, lds, s, ...) SALVA_LAUNCH_TILE_3(kernel, kernel, 0u, kernel, c, L, lds, s, __VA_ARGS__)

// key of cell (cx,cy,cz) (absolute cell coords) in grid g, or inside = false.  (mx, my, mz): the folding masks of the grid the
// COORDINATES belong to (device_types.h TileGrid) — g's own, except where a tile of the folded fluid grid looks up the cells of the
// boundary grid, which is never folded itself but is then addressed modulo the fluid grid's periods (they are at least as long as
// the boundary grid is wide: at most one of its cells answers).
__device__ __forceinline__ uint32_t tile_key_m(const TileGrid& g, int cx, int cy, int cz, bool& inside, uint32_t mx, uint32_t my, uint32_t mz) {
    const uint32_t ix = ((uint32_t)cx - (uint32_t)g.ox) & mx, iy = ((uint32_t)cy - (uint32_t)g.oy) & my, iz = ((uint32_t)cz - (uint32_t)g.oz) & mz;
    inside = ix < (uint32_t)(g.ntx * TX) && iy < (uint32_t)(g.nty * TY) && iz < (uint32_t)(g.ntz * TZ);
    const uint32_t tile = ((ix / TX) * (uint32_t)g.nty + (iy / TY)) * (uint32_t)g.ntz + (iz / TZ);
    return tile * TCELLS + (((ix % TX) * TY + (iy % TY)) * TZ + (iz % TZ));
}
__device__ __forceinline__ uint32_t tile_key(const TileGrid& g, int cx, int cy, int cz, bool& inside) {
    return tile_key_m(g, cx, cy, cz, inside, g.mx, g.my, g.mz);
}

// floor(x / h) exactly as hgrid.rs:41-43 (IEEE f32 division, then floor), clamped to +-2^30; NaN -> bad.
__device__ __forceinline__ int cell_coord(float x, float h, bool& bad) {
    float f = floorf(__fdiv_rn(x, h));
    if (!(f == f)) { bad = true; f = 0.0f; }
    f = fminf(fmaxf(f, -1073741824.0f), 1073741824.0f);
    return (int)f;
}

extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];

// A staged float4 read from LDS with all 16 bytes: where the caller never looks at .w the compiler narrows the read to
// ds_read_b96, which occupies the CU's LDS pipe for 8 cycles per wave instead of ds_read_b128's 4 (MI355X_MICROARCH.md, LDS
// table) — the difference between 165 and 119 us for the neighbour search, whose loop is little else than LDS reads.
__device__ __forceinline__ float4 lds_f4(const float4* p) {
    float4 v = *p;
    asm volatile("" : "+v"(v.w));
    return v;
}

// LDS reads by BYTE ADDRESS.  Dynamic LDS starts at byte 0 in a kernel that declares no static __shared__ variable, so the
// first staged array of such a kernel starts at LDS address 0 and `slot << 4` IS the address: the compiler then folds any
// compile-time displacement into the instruction's offset field (it cannot do that through the tile_smem symbol, whose
// address it only learns after instruction selection).  Kernels that use these call lds_base_check() once: the comparison
// folds away at compile time, or — if someone adds a static __shared__ array — becomes an unconditional trap that the first
// GPU test hits (and `grep s_trap` on the ISA shows at build time).
typedef float lds_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_base_check() {
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)tile_smem != 0u) __builtin_trap();
}
__device__ __forceinline__ float4 lds_ld16(uint32_t byte_addr) {
    const lds_v4f v = *(const __attribute__((address_space(3))) lds_v4f*)(uintptr_t)byte_addr;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float lds_ld4(uint32_t byte_addr) {
    return *(const __attribute__((address_space(3))) float*)(uintptr_t)byte_addr;
}
typedef float lds_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lds_v2f lds_ld8(uint32_t byte_addr) {
    return *(const __attribute__((address_space(3))) lds_v2f*)(uintptr_t)byte_addr;
}
__device__ __forceinline__ void lds_st8(uint32_t byte_addr, float a, float b) {
    *(__attribute__((address_space(3))) lds_v2f*)(uintptr_t)byte_addr = lds_v2f{a, b};
}

// LDS-DMA (global_load_lds): every lane names its own global source, the 64 lanes' data land in LDS side by side from a
// wave-uniform base — the shape of an index-gather into consecutive halo slots.  No VGPR round trip, no ds_write pass.
// Completion is counted on vmcnt: wait for it before the barrier that publishes the staged data.
__device__ __forceinline__ void glds16(const float4* __restrict__ src, float4* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* __restrict__ src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}


// ---------------------------------------------------------------------------------------------------
// Tile: what every tile kernel needs.  The halo of a tile (the particles of its 6x6x6 cell box, in halo-cell order)
// is described once per step by k_tile_halo_fill as a flat table of sorted particle indices
// (halo_src[halo_off[tile] + slot]); staging an array into LDS is then one coalesced index read plus one gather per
// slot, all slots in flight at once.
// ---------------------------------------------------------------------------------------------------
struct Tile {
    unsigned char* pool;  // staged arrays, 16-byte aligned
    uint32_t pool_used;
    uint32_t S, SB;       // halo slot counts (fluid / boundary)
    uint32_t tile;        // index in the dense tile grid
    uint32_t part;        // which x-planes of it this slot owns (tile_part_code; TILE_PART_WHOLE unless the tile was split)
    uint32_t slot;        // index among the launched (non-empty) tiles: per-workgroup outputs (error partials, list stats)
    uint32_t own_begin, own_end, slice_base;
    uint64_t hoff, hboff; // offsets of this tile's slot tables in halo_src / bhalo_src
    int hcx, hcy, hcz;    // absolute cell coords of halo cell (0,0,0)
    // slot-table entries fetched speculatively by setup(), before the tile's sizes are known (strided tables only):
    // the index fetch then overlaps the descriptor fetch and staging is two dependent round trips instead of three
    static constexpr int PRE = 4;
    uint32_t pre0, pre1, pre2, pre3, preb;
    bool skip;  // this launch is not the one that handles the tile (StepCtx::phase): leave every output alone
    float mass;   // what the plane-layout kernels multiply their finished sums by: StepCtx::mass_uniform, or — two-mass worlds — the
                  // mass of the first segment of THIS tile's lists (StepCtx::tile_mass_bits)
    float massb;  // two-mass worlds: the mass of the second segment, 0 = this tile's halo holds one mass only
    float massc, massd;  // worlds with three / four masses: the masses of the third and fourth segment, 0 = there is none

    __device__ __forceinline__ bool empty() const { return own_begin == own_end; }
    __device__ __forceinline__ bool skipped() const { return skip; }

    // geometry only (k_tile_count: before the per-slot prefix table exists).  Launched over an upper bound of the slot
    // count (the host does not know it yet): returns false for the surplus workgroups.
    __device__ __forceinline__ bool setup_geom(const StepCtx& c) {
        pool = tile_smem;
        pool_used = 0;
        skip = false;
        mass = massb = massc = massd = 0.0f;
        slot = blockIdx.x;
        S = SB = 0; slice_base = 0; hoff = hboff = 0;
        pre0 = pre1 = pre2 = pre3 = preb = 0u;
        own_begin = own_end = 0; tile = 0; hcx = hcy = hcz = 0;
        part = TILE_PART_WHOLE;
        if (slot >= c.tile_rank[c.ntiles]) return false;
        const uint32_t id = c.tile_ids[slot];
        tile = id & ((1u << TILE_PART_SHIFT) - 1u);
        part = id >> TILE_PART_SHIFT;
        geometry(c);
        return true;
    }
    __device__ __forceinline__ uint32_t part_ux0() const { return part & 3u; }
    __device__ __forceinline__ uint32_t part_len() const { return (part >> 2) + 1u; }
    __device__ __forceinline__ void geometry(const StepCtx& c) {
        const TileGrid& g = c.gf;
        const uint32_t ttz = tile % g.ntz, tty = (tile / g.ntz) % g.nty, ttx = tile / (g.ntz * g.nty);
        own_begin = g.cell_start[(size_t)tile * TCELLS + part_ux0() * (TY * TZ)];
        own_end = g.cell_start[(size_t)tile * TCELLS + (part_ux0() + part_len()) * (TY * TZ)];
        hcx = g.ox + (int)ttx * TX - 1; hcy = g.oy + (int)tty * TY - 1; hcz = g.oz + (int)ttz * TZ - 1;
    }

    // workgroup k of a launch works on slot k = the k-th non-empty tile (XCD-remapped so that neighbours share an L2)
    // `stagger`: a 10^6-particle launch is ~4 rounds of resident tiles deep and its first round starts in lock step — every
    // CU's tiles index, then stage (HBM busy, LDS idle), then sum (the reverse) at the same time, and the rounds after it
    // inherit the rhythm.  Holding the first round's workgroups back in four phases (0 / 4 / 8 / 12 k cycles by rank)
    // interleaves staging and summing from the start: -2..3 % on a settled step (k_divergence 50.2 -> 48.6 us); pointless
    // for launches of fewer than two rounds and for k_nbr_tile, whose phases are not memory / LDS halves (it measured
    // +2.5 %), which pass false (profiles/r03_experiments/r03qr_stagger.log).  Timing only.
    __device__ __forceinline__ void setup(const StepCtx& c, bool stagger = true) {
        if (stagger && gridDim.x >= 1024u) {
            const unsigned rk = blockIdx.x >> 3;
            if (rk < 64u)
                for (unsigned i = 0; i < ((rk >> 4) & 3u); ++i) __builtin_amdgcn_s_sleep(64);
        }
        const uint32_t b = xcd_block(blockIdx.x, gridDim.x, c.xcd);
        setup_at(c, c.slot_order ? c.slot_order[c.slot_base + b] : b);  // (two launch classes: StepCtx::slot_order)
    }
    __device__ __forceinline__ void setup_at(const StepCtx& c, uint32_t at_slot) {
        pool = tile_smem;
        pool_used = 0;
        slot = at_slot;
        // A chained step whose chain broke upstream (device_types.h StepCtx::gate): nothing to do.  The word is fetched here and looked
        // at behind the other loads of the set-up — tested first, it puts a dependent round trip (a word the previous kernel wrote) in
        // front of every tile's life: +8-10 us per free-fall step when it went in (profiles/r06_experiments/r06a_chain.log).
        const uint32_t gate_word = c.gate ? (gate_words_closed(c.gate[0], c.gate[1], c.gate_stage) ? 0u : 1u) : 1u;
        const uint4 desc = c.slot_desc[slot];
        mass = c.two_mass ? __uint_as_float(c.tile_mass_bits[slot]) : c.mass_uniform;
        massb = c.two_mass ? __uint_as_float(c.tile_massb_bits[slot]) : 0.0f;
        massc = massd = 0.0f;
        if (c.tile_masscd_bits) { const uint2 cd = c.tile_masscd_bits[slot]; massc = __uint_as_float(cd.x); massd = __uint_as_float(cd.y); }
        pre0 = pre1 = pre2 = pre3 = preb = 0u;
        if (c.halo_stride) {
            const uint32_t* __restrict__ src = c.halo_src + (size_t)slot * c.halo_stride;
            const uint32_t s0 = threadIdx.x, nt = blockDim.x, lim = c.halo_stride;
            if (s0 < lim) pre0 = src[s0];
            if (s0 + nt < lim) pre1 = src[s0 + nt];
            if (s0 + 2 * nt < lim) pre2 = src[s0 + 2 * nt];
            if (s0 + 3 * nt < lim) pre3 = src[s0 + 3 * nt];
            if (c.bhalo_stride && s0 < c.bhalo_stride) preb = c.bhalo_src[(size_t)slot * c.bhalo_stride + s0];
        }
        tile = desc.x; own_begin = desc.y; own_end = desc.z; part = desc.w;
        {
            const TileGrid& g = c.gf;
            const uint32_t ttz = tile % g.ntz, tty = (tile / g.ntz) % g.nty, ttx = tile / (g.ntz * g.nty);
            hcx = g.ox + (int)ttx * TX - 1; hcy = g.oy + (int)tty * TY - 1; hcz = g.oz + (int)ttz * TZ - 1;
        }
        skip = false;
        if (c.phase) {
            const bool border = hcx <= c.ghost_lo_cx || hcx + HX - 1 >= c.ghost_hi_cx;
            skip = border != (c.phase == 2);
        }
        const TileAcc a0 = c.tile_off[slot], a1 = c.tile_off[slot + 1];
        slice_base = a0.nsl;
        S = (uint32_t)(a1.s - a0.s);
        SB = (uint32_t)(a1.sb - a0.sb);
        hoff = c.halo_stride ? (uint64_t)slot * c.halo_stride : a0.s;
        hboff = c.halo_stride ? (uint64_t)slot * c.bhalo_stride : a0.sb;
        if (c.spec) {  // speculative pass: stay inside what the host allocated (device_types.h)
            S = min(S, c.halo_cap);
            SB = min(SB, c.bhalo_cap);
            if (!c.halo_stride) {
                S = (uint32_t)min((uint64_t)S, c.halo_len > hoff ? c.halo_len - hoff : 0ull);
                SB = (uint32_t)min((uint64_t)SB, c.bhalo_len > hboff ? c.bhalo_len - hboff : 0ull);
            }
            if (slot >= c.tile_rank[c.ntiles] || a1.nsl > c.nslices_cap) { own_end = own_begin; S = SB = 0; }
        }
        if (gate_word == 0u) { skip = true; own_end = own_begin; S = SB = 0; }
    }

    template <typename T>
    __device__ __forceinline__ T* carve(uint32_t count) {
        T* p = reinterpret_cast<T*>(pool + pool_used);
        pool_used += (count * (uint32_t)sizeof(T) + 15u) & ~15u;
        return p;
    }

    // f(slot, sorted_index) for every fluid halo slot; 4 independent index loads / gathers in flight per thread
    template <typename F>
    __device__ __forceinline__ void for_halo(const StepCtx& c, F&& f) const {
        const uint32_t* __restrict__ src = c.halo_src + hoff;
        if (c.halo_stride) {
            const uint32_t s0 = threadIdx.x, nt = blockDim.x;
            if (s0 < S) f(s0, pre0);
            if (s0 + nt < S) f(s0 + nt, pre1);
            if (s0 + 2 * nt < S) f(s0 + 2 * nt, pre2);
            if (s0 + 3 * nt < S) f(s0 + 3 * nt, pre3);
            for (uint32_t s = s0 + PRE * nt; s < S; s += nt) f(s, src[s]);
        } else {
#pragma unroll 4
            for (uint32_t s = threadIdx.x; s < S; s += blockDim.x) f(s, src[s]);
        }
    }
    template <typename F>
    __device__ __forceinline__ void for_halo_boundary(const StepCtx& c, F&& f) const {
        const uint32_t* __restrict__ src = c.bhalo_src + hboff;
        if (c.halo_stride) {
            if (threadIdx.x < SB) f(threadIdx.x, preb);
            for (uint32_t s = threadIdx.x + blockDim.x; s < SB; s += blockDim.x) f(s, src[s]);
            return;
        }
#pragma unroll 2
        for (uint32_t s = threadIdx.x; s < SB; s += blockDim.x) f(s, src[s]);
    }

    // Stage global per-particle arrays into LDS (one pass over the slot table for all of them).  No barrier inside: follow
    // with staged_barrier().  With strided slot tables (the indices are in registers already, setup()) the copy is LDS-DMA
    // in 64-slot chunks; otherwise global_load + ds_write.
    template <typename T>
    static __device__ __forceinline__ void dma_chunk(const T* __restrict__ src, T* lds_wave_base) {
        static_assert(sizeof(T) == 16 || sizeof(T) == 4, "LDS-DMA moves 4 or 16 bytes per lane");
        if constexpr (sizeof(T) == 16) glds16(reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(lds_wave_base));
        else glds4(reinterpret_cast<const float*>(src), reinterpret_cast<float*>(lds_wave_base));
    }
    // f(chunk_base_slot, source_index) for every 64-slot chunk this wave copies (wave-uniform control flow)
    template <typename F>
    __device__ __forceinline__ void for_halo_chunks(const StepCtx& c, F&& f) const {
        const uint32_t nt = blockDim.x, lane = threadIdx.x & (WAVE - 1), nw = nt / WAVE;
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
        const uint32_t pre[PRE] = {pre0, pre1, pre2, pre3};
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const uint32_t s0 = (wv + (uint32_t)k * nw) * WAVE;
            if (s0 < S) f(s0, (s0 + lane < S) ? pre[k] : own_begin);
        }
        for (uint32_t s0 = (wv + (uint32_t)PRE * nw) * WAVE; s0 < S; s0 += nt)
            f(s0, (s0 + lane < S) ? c.halo_src[hoff + s0 + lane] : own_begin);
    }
    __device__ __forceinline__ uint32_t stage_cap(const StepCtx& c) const { return c.halo_stride ? ((S + 63u) & ~63u) : S; }
    template <typename T0>
    __device__ __forceinline__ void stage(const StepCtx& c, const T0* __restrict__ s0, const T0*& d0) {
        T0* a = carve<T0>(stage_cap(c));
        if (c.halo_stride) for_halo_chunks(c, [&](uint32_t b, uint32_t g) { dma_chunk(s0 + g, a + b); });
        else for_halo(c, [&](uint32_t s, uint32_t g) { a[s] = s0[g]; });
        d0 = a;
    }
    template <typename T0, typename T1>
    __device__ __forceinline__ void stage(const StepCtx& c, const T0* __restrict__ s0, const T1* __restrict__ s1,
                                          const T0*& d0, const T1*& d1) {
        T0* a = carve<T0>(stage_cap(c)); T1* b = carve<T1>(stage_cap(c));
        if (c.halo_stride) for_halo_chunks(c, [&](uint32_t q, uint32_t g) { dma_chunk(s0 + g, a + q); dma_chunk(s1 + g, b + q); });
        else for_halo(c, [&](uint32_t s, uint32_t g) { a[s] = s0[g]; b[s] = s1[g]; });
        d0 = a; d1 = b;
    }
    template <typename T0, typename T1, typename T2>
    __device__ __forceinline__ void stage(const StepCtx& c, const T0* __restrict__ s0, const T1* __restrict__ s1,
                                          const T2* __restrict__ s2, const T0*& d0, const T1*& d1, const T2*& d2) {
        T0* a = carve<T0>(stage_cap(c)); T1* b = carve<T1>(stage_cap(c)); T2* e = carve<T2>(stage_cap(c));
        if (c.halo_stride)
            for_halo_chunks(c, [&](uint32_t q, uint32_t g) { dma_chunk(s0 + g, a + q); dma_chunk(s1 + g, b + q); dma_chunk(s2 + g, e + q); });
        else for_halo(c, [&](uint32_t s, uint32_t g) { a[s] = s0[g]; b[s] = s1[g]; e[s] = s2[g]; });
        d0 = a; d1 = b; d2 = e;
    }
    template <typename T0, typename T1, typename T2, typename T3>
    __device__ __forceinline__ void stage(const StepCtx& c, const T0* __restrict__ s0, const T1* __restrict__ s1,
                                          const T2* __restrict__ s2, const T3* __restrict__ s3, const T0*& d0,
                                          const T1*& d1, const T2*& d2, const T3*& d3) {
        T0* a = carve<T0>(stage_cap(c)); T1* b = carve<T1>(stage_cap(c)); T2* e = carve<T2>(stage_cap(c)); T3* f = carve<T3>(stage_cap(c));
        if (c.halo_stride)
            for_halo_chunks(c, [&](uint32_t q, uint32_t g) {
                dma_chunk(s0 + g, a + q); dma_chunk(s1 + g, b + q); dma_chunk(s2 + g, e + q); dma_chunk(s3 + g, f + q);
            });
        else for_halo(c, [&](uint32_t s, uint32_t g) { a[s] = s0[g]; b[s] = s1[g]; e[s] = s2[g]; f[s] = s3[g]; });
        d0 = a; d1 = b; d2 = e; d3 = f;
    }
    // ---- fixed-base layouts (lds_ld16 / lds_ld4): call on a fresh pool (nothing carved yet) ----
    // P | W: two 16-byte arrays, P at byte 0 and W at byte `dist`; each holds the fluid halo in slots [0, cap) and the
    // boundary halo in slots [cap, cap + SB) (b0 / b1: bposv / bvel; pass nullptr for b1 to leave W's tail unused).
    // Requires dist >= (cap + SB) * 16.  Afterwards: *bp / *bv = the boundary parts, pool_used = the end of W.
    template <typename T>
    __device__ __forceinline__ void stage_pw(const StepCtx& c, const T* __restrict__ s0, const T* __restrict__ s1, uint32_t dist,
                                             const float4*& bp, const float4*& bv, bool with_bv) {
        static_assert(sizeof(T) == 16, "16-byte records");
        const uint32_t cap = stage_cap(c);
        T* a = reinterpret_cast<T*>(pool);
        T* b = reinterpret_cast<T*>(pool + dist);
        if (c.halo_stride) for_halo_chunks(c, [&](uint32_t q, uint32_t g) { dma_chunk(s0 + g, a + q); dma_chunk(s1 + g, b + q); });
        else for_halo(c, [&](uint32_t s, uint32_t g) { a[s] = s0[g]; b[s] = s1[g]; });
        float4* ba = reinterpret_cast<float4*>(a) + cap;
        float4* bb = reinterpret_cast<float4*>(b) + cap;
        if (with_bv) for_halo_boundary(c, [&](uint32_t s, uint32_t g) { ba[s] = c.bposv[g]; bb[s] = c.bvel[g]; });
        else for_halo_boundary(c, [&](uint32_t s, uint32_t g) { ba[s] = c.bposv[g]; });
        bp = ba; bv = bb;
        pool_used = dist + (cap + SB) * 16u;
    }
    // P | K: one 16-byte array at byte 0 (fluid halo in [0, cap), then bposv in [cap, cap + SB), then bvel in
    // [cap + SB, cap + 2 SB)) and one 4-byte array at byte `dist` (fluid halo only).  Requires dist >= (cap + 2 SB) * 16.
    __device__ __forceinline__ void stage_pk(const StepCtx& c, const float4* __restrict__ s0, const float* __restrict__ s1, uint32_t dist,
                                             const float4*& bp, const float4*& bv) {
        const uint32_t cap = stage_cap(c);
        float4* a = reinterpret_cast<float4*>(pool);
        float* b = reinterpret_cast<float*>(pool + dist);
        if (c.halo_stride) for_halo_chunks(c, [&](uint32_t q, uint32_t g) { dma_chunk(s0 + g, a + q); dma_chunk(s1 + g, b + q); });
        else for_halo(c, [&](uint32_t s, uint32_t g) { a[s] = s0[g]; b[s] = s1[g]; });
        float4* ba = a + cap;
        float4* bb = a + cap + SB;
        for_halo_boundary(c, [&](uint32_t s, uint32_t g) { ba[s] = c.bposv[g]; bb[s] = c.bvel[g]; });
        bp = ba; bv = bb;
        pool_used = dist + cap * 4u;
    }
    // ---- P3: the evaluate kernels' layout when every particle has the same mass (StepCtx::mass_uniform) ----
    // Three 8-byte planes XY | ZU | VW, (U, V, W) = w = v + dv, at LDS bytes 0, dist8, 2 dist8: the fluid halo, slots [0, S).  The
    // boundary halo (few slots, wall tiles only) stays in 16-byte arrays behind the last plane (stage_boundary).
    // 24 bytes per slot against the 32 of P | W (the mass is a constant of the launch, the model id is not used): a
    // 2100-slot halo is 50 KB, and THREE workgroups share a CU's 160 KB where two did — the third resident tile is what overlaps
    // one tile's index / stage / barrier phases with another's pair loop (+11-13 % measured in round 3 on a scene
    // whose halos fitted).  A contact then costs three ds_read_b64 (2 LDS cycles each) instead of two ds_read_b128 (4 each).
    // Filled through registers — two 16-byte loads and three ds_write_b64 per slot: LDS-DMA moves 4 or 16 bytes per lane, so it
    // could only fill 4-byte planes, with three times the staging instructions.  No padding to 64 slots is needed this way.
    // Requires dist8 >= S * 8, a multiple of 8.  Afterwards pool_used = the end of the third plane.
    __device__ __forceinline__ void p3_store(uint32_t slot, uint32_t dist8, const float4& a, const float4& b) const {
        lds_st8(slot * 8u, a.x, a.y);
        lds_st8(slot * 8u + dist8, a.z, b.x);
        lds_st8(slot * 8u + 2u * dist8, b.y, b.z);
    }
    __device__ __forceinline__ void stage_p3(const StepCtx& c, const float4* __restrict__ posm, const float4* __restrict__ w, uint32_t dist8) {
        const uint32_t nt = blockDim.x, s0 = threadIdx.x;
        if (c.halo_stride) {
            // every load of the thread's (up to) four slots is issued before the first store waits for one
            const uint32_t pre[PRE] = {pre0, pre1, pre2, pre3};
            float4 a[PRE], b[PRE];
#pragma unroll
            for (int k = 0; k < PRE; ++k) {
                const uint32_t sl = s0 + (uint32_t)k * nt;
                const uint32_t g = sl < S ? pre[k] : own_begin;
                a[k] = posm[g]; b[k] = w[g];
            }
#pragma unroll
            for (int k = 0; k < PRE; ++k) {
                const uint32_t sl = s0 + (uint32_t)k * nt;
                if (sl < S) p3_store(sl, dist8, a[k], b[k]);
            }
            for (uint32_t sl = s0 + PRE * nt; sl < S; sl += nt) {
                const uint32_t g = c.halo_src[hoff + sl];
                p3_store(sl, dist8, posm[g], w[g]);
            }
        } else {
            for_halo(c, [&](uint32_t sl, uint32_t g) { p3_store(sl, dist8, posm[g], w[g]); });
        }
        pool_used = 2u * dist8 + ((S * 8u + 15u) & ~15u);  // (the boundary halo follows as 16-byte arrays: stage_boundary)
    }
    // ---- P2: the apply kernels' layout under the same condition: two 8-byte planes XY | ZK (K = kappa) at LDS bytes 0 and dist8;
    // the boundary halo in slots [S, S + SB) as (x, y) | (z, V_b).  16 bytes per slot against the 20 of P | K, and two
    // ds_read_b64 per contact instead of a ds_read_b128 and a ds_read_b32.  pool_used = the end of the second plane afterwards.
    __device__ __forceinline__ void stage_p2(const StepCtx& c, const float4* __restrict__ posm, const float* __restrict__ k, uint32_t dist8) {
        const uint32_t nt = blockDim.x, s0 = threadIdx.x;
        if (c.halo_stride) {
            const uint32_t pre[PRE] = {pre0, pre1, pre2, pre3};
            float4 a[PRE];
            float b[PRE];
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const uint32_t sl = s0 + (uint32_t)q * nt;
                const uint32_t g = sl < S ? pre[q] : own_begin;
                a[q] = posm[g]; b[q] = k[g];
            }
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const uint32_t sl = s0 + (uint32_t)q * nt;
                if (sl < S) { lds_st8(sl * 8u, a[q].x, a[q].y); lds_st8(sl * 8u + dist8, a[q].z, b[q]); }
            }
            for (uint32_t sl = s0 + PRE * nt; sl < S; sl += nt) {
                const uint32_t g = c.halo_src[hoff + sl];
                const float4 p = posm[g];
                lds_st8(sl * 8u, p.x, p.y); lds_st8(sl * 8u + dist8, p.z, k[g]);
            }
        } else {
            for_halo(c, [&](uint32_t sl, uint32_t g) {
                const float4 p = posm[g];
                lds_st8(sl * 8u, p.x, p.y); lds_st8(sl * 8u + dist8, p.z, k[g]);
            });
        }
        for_halo_boundary(c, [&](uint32_t sl, uint32_t g) {
            const float4 p = c.bposv[g];
            lds_st8((S + sl) * 8u, p.x, p.y); lds_st8((S + sl) * 8u + dist8, p.z, p.w);
        });
        pool_used = dist8 + (S + SB) * 8u;
    }
    // the barrier that publishes the staged halo: the DMA of this wave has landed (vmcnt), then everybody's has
    static __device__ __forceinline__ void staged_barrier() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    __device__ __forceinline__ void stage_boundary(const StepCtx& c, const float4*& bp) {
        float4* a = carve<float4>(SB);
        for_halo_boundary(c, [&](uint32_t s, uint32_t g) { a[s] = c.bposv[g]; });
        bp = a;
    }
    __device__ __forceinline__ void stage_boundary(const StepCtx& c, const float4*& bp, const float4*& bv) {
        float4* a = carve<float4>(SB); float4* b = carve<float4>(SB);
        for_halo_boundary(c, [&](uint32_t s, uint32_t g) { a[s] = c.bposv[g]; b[s] = c.bvel[g]; });
        bp = a; bv = b;
    }

    // This wave's first slice: own-particle index / global slice; false if the wave has no slice or the lane is idle —
    // i / gs then name the tile's first particle / slice, so that the caller can load unconditionally (a conditional
    // aggregate load makes the compiler park the record in scratch memory: 80 B/lane of dead HBM stores).
    __device__ __forceinline__ bool first_own(uint32_t& i, uint32_t& gs) const {
        const uint32_t s = threadIdx.x / WAVE, nsl = (own_end - own_begin + WAVE - 1) / WAVE;
        i = own_begin + s * WAVE + (threadIdx.x & (WAVE - 1));
        gs = slice_base + s;
        const bool ok = s < nsl && i < own_end;
        if (!ok) { i = own_begin; gs = slice_base; }
        return ok;
    }
    // Like for_own, but the per-particle inputs of the wave's first slice were loaded before the staging barrier
    // (`pre0`), so their latency overlaps the halo copy; later slices (tiles fuller than the workgroup) load on demand.
    template <typename Pre, typename LoadOwn, typename Body>
    __device__ __forceinline__ void for_own_pre(const Pre& pre0, LoadOwn&& load_own, Body&& body) const {
        const uint32_t nsl = (own_end - own_begin + WAVE - 1) / WAVE;
        const uint32_t lane = threadIdx.x & (WAVE - 1), nw = blockDim.x / WAVE;
        uint32_t s = threadIdx.x / WAVE;
        if (s < nsl) {
            const uint32_t i = own_begin + s * WAVE + lane;
            body(pre0, i, slice_base + s, i < own_end);
            s += nw;
        }
        for (; s < nsl; s += nw) {  // tiles fuller than the workgroup
            const uint32_t i = own_begin + s * WAVE + lane, gs = slice_base + s;
            const bool active = i < own_end;
            const Pre p = load_own(active ? i : own_begin, gs);
            body(p, i, gs, active);
        }
    }

    // Visit the tile's own particles, one wave per 64-particle slice: f(i, global_slice, active).
    template <typename F>
    __device__ __forceinline__ void for_own(F&& f) const {
        const uint32_t nsl = (own_end - own_begin + WAVE - 1) / WAVE;
        const uint32_t lane = threadIdx.x & (WAVE - 1), nw = blockDim.x / WAVE;
        for (uint32_t s = threadIdx.x / WAVE; s < nsl; s += nw) {
            const uint32_t i = own_begin + s * WAVE + lane;
            f(i, slice_base + s, i < own_end);
        }
    }
};

// Per-cell slot tables of a tile's halo box, built in LDS from the global cell tables.  Needed only where the halo
// is traversed cell by cell: the per-step table builders (k_tile_count / k_tile_halo_fill) and the neighbour-list
// builder.  lstart[h] = first slot of halo cell h = (hx*HY + hy)*HZ + hz ; gstart[h] = its first sorted index.
struct TileCells {
    uint32_t *lstart, *gstart, *blstart, *bgstart;

    // tables live at the start of dynamic LDS; returns the number of bytes they occupy
    __device__ __forceinline__ void build(const StepCtx& c, Tile& t, bool clamp_to_staged = false) {
        uint32_t* tab = t.carve<uint32_t>(2 * (HCELLS + 1) + 2 * HCELLS);
        lstart = tab; gstart = tab + (HCELLS + 1); blstart = gstart + HCELLS; bgstart = blstart + (HCELLS + 1);
        for (int h = threadIdx.x; h < HCELLS; h += blockDim.x) {  // (one trip at >= HCELLS threads; the sparse class runs 64)
            const int hz = h % HZ, hy = (h / HZ) % HY, hx = h / (HZ * HY);
            // (a part of a split tile stages the x-planes around its own cells only: the others hold no neighbour of its particles)
            const bool plane = (uint32_t)hx >= t.part_ux0() && (uint32_t)hx < t.part_ux0() + t.part_len() + 2u;
            bool in;
            const uint32_t k = tile_key(c.gf, t.hcx + hx, t.hcy + hy, t.hcz + hz, in);
            uint32_t b = 0, e = 0;
            if (in && plane) { b = c.gf.cell_start[k]; e = c.gf.cell_start[k + 1]; }
            gstart[h] = b; lstart[h] = e - b;
            b = e = 0;
            if (c.nb && plane) {
                const uint32_t kb = tile_key_m(c.gb, t.hcx + hx, t.hcy + hy, t.hcz + hz, in, c.gf.mx, c.gf.my, c.gf.mz);
                if (in) { b = c.gb.cell_start[kb]; e = c.gb.cell_start[kb + 1]; }
            }
            bgstart[h] = b; blstart[h] = e - b;
        }
        __syncthreads();
        if (threadIdx.x < WAVE) {  // exclusive prefix of the HCELLS counts by one wave, PER lane each
            constexpr int PER = (HCELLS + WAVE - 1) / WAVE;
            constexpr int NL = (HCELLS + PER - 1) / PER;  // lanes that hold cells (the last one maybe fewer than PER)
            const int l = threadIdx.x;
            uint32_t a[PER], b[PER], sa = 0, sb = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const bool in = PER * l + k < HCELLS;
                a[k] = in ? lstart[PER * l + k] : 0u;
                b[k] = in ? blstart[PER * l + k] : 0u;
                sa += a[k]; sb += b[k];
            }
            uint32_t ia = sa, ib = sb;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const uint32_t ta = (uint32_t)__shfl_up((int)ia, o, WAVE), tb = (uint32_t)__shfl_up((int)ib, o, WAVE);
                if (l >= o) { ia += ta; ib += tb; }
            }
            uint32_t ea = ia - sa, eb = ib - sb;
            if (l < NL) {
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    if (PER * l + k < HCELLS) { lstart[PER * l + k] = ea; blstart[PER * l + k] = eb; }
                    ea += a[k]; eb += b[k];
                }
            }
            if (l == NL - 1) { lstart[HCELLS] = ia; blstart[HCELLS] = ib; }
        }
        __syncthreads();
        if (clamp_to_staged && c.spec) {  // speculative pass: no cell range may reach beyond the slots that were staged
            for (int q = threadIdx.x; q <= HCELLS; q += blockDim.x) {
                lstart[q] = min(lstart[q], t.S);
                blstart[q] = min(blstart[q], t.SB);
            }
            __syncthreads();
        }
    }
};

// Neighbour iteration: 16-bit LDS slots, two per dword, sliced-ELL per 64-particle slice.
// `load(slot)` fetches a neighbour's staged record(s) from LDS and `compute(record)` folds it in.  The two (or four)
// loads of an iteration are issued before the first compute, so the LDS latency of one contact overlaps the
// arithmetic of another; the odd tail reads slot 0 and discards it.
template <typename L, typename C>
__device__ __forceinline__ void for_each_slot(const uint32_t* __restrict__ nbr, uint32_t cap, uint32_t gslice,
                                              uint32_t cnt, L&& load, C&& compute) {
    if (cnt == 0) return;
    const uint32_t* __restrict__ p = nbr + (size_t)gslice * cap * WAVE + 4u * (threadIdx.x & (WAVE - 1));
    const uint32_t nq = (cnt + 1) >> 1;
    uint32_t q = 0;
    uint32_t n0 = p[0], n1 = (nq > 1) ? p[ellq(1)] : 0u;
    for (; q + 2 <= nq; q += 2) {
        const uint32_t a = n0, b = n1;
        if (q + 2 < nq) n0 = p[ellq(q + 2)];
        if (q + 3 < nq) n1 = p[ellq(q + 3)];
        const auto d0 = load(a & 0xffffu);
        const auto d1 = load(a >> 16);
        const auto d2 = load(b & 0xffffu);
        const auto d3 = load(b >> 16);
        compute(d0);
        compute(d1);
        compute(d2);
        if (2 * q + 3 < cnt) compute(d3);
    }
    if (q < nq) {  // one dword left
        const uint32_t a = n0;
        const auto d0 = load(a & 0xffffu);
        const auto d1 = load(a >> 16);
        compute(d0);
        if (2 * q + 1 < cnt) compute(d1);
    }
}
// the first two dwords of a particle's fluid-fluid list, loadable before the staging barrier (the ELL block of every
// slice is allocated, so the read is always in bounds; its content is ignored when the list is shorter)
struct ListHead { uint32_t n0, n1; };
__device__ __forceinline__ ListHead list_head(const StepCtx& c, uint32_t gslice) {
    const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
    return ListHead{p[0], p[ellq(1)]};
}
template <typename L, typename C>
__device__ __forceinline__ void for_each_ff(const StepCtx& c, uint32_t gslice, uint32_t cnt, const ListHead& lh, L&& load,
                                            C&& compute) {
    if (cnt == 0) return;
    const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
    const uint32_t nq = (cnt + 1) >> 1;
    uint32_t q = 0;
    uint32_t n0 = lh.n0, n1 = lh.n1;
    for (; q + 2 <= nq; q += 2) {
        const uint32_t a = n0, b = n1;
        if (q + 2 < nq) n0 = p[ellq(q + 2)];
        if (q + 3 < nq) n1 = p[ellq(q + 3)];
        const auto d0 = load(a & 0xffffu);
        const auto d1 = load(a >> 16);
        const auto d2 = load(b & 0xffffu);
        const auto d3 = load(b >> 16);
        compute(d0);
        compute(d1);
        compute(d2);
        if (2 * q + 3 < cnt) compute(d3);
    }
    if (q < nq) {
        const uint32_t a = n0;
        const auto d0 = load(a & 0xffffu);
        const auto d1 = load(a >> 16);
        compute(d0);
        if (2 * q + 1 < cnt) compute(d1);
    }
}
// Pair form for the gradient passes: compute2(A, B) handles two contacts at once (packed f32).  An odd list is padded
// with the particle's own slot by k_nbr_tile — the self contact has d = 0 and contributes exactly nothing to any
// gradient sum — so there is no validity test anywhere in the loop.
// The first LIST_REGS dwords (2 contacts each) of a particle's list, loaded into registers BEFORE the staging barrier:
// a list dword fetched inside the neighbour loop costs a full HBM/L2 round trip per iteration, which is longer than
// the arithmetic of the iteration — measured at 15 of 54 us in k_pred_density.  Every ELL row has at least LIST_REGS
// dwords allocated (cap_ff >= LIST_REGS), so the loads are unconditional; what lies beyond a lane's own list is ignored.
constexpr int LIST_REGS = 20;
struct ListRegs { uint32_t d[LIST_REGS]; };
__device__ __forceinline__ ListRegs list_regs(const StepCtx& c, uint32_t gslice) {
    const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
    static_assert(LIST_REGS % 4 == 0, "the list head is fetched in 16-byte pieces");
    const uint4* __restrict__ p4 = reinterpret_cast<const uint4*>(p);
    ListRegs r;
#pragma unroll
    for (int k = 0; k < LIST_REGS / 4; ++k) {
        const uint4 v = p4[(size_t)k * WAVE];
        r.d[4 * k] = v.x; r.d[4 * k + 1] = v.y; r.d[4 * k + 2] = v.z; r.d[4 * k + 3] = v.w;
    }
    return r;
}
// `nq` = dwords of the longest list in the slice (slice_list_dwords below): wave-uniform, so every branch here is scalar.
// AHEAD: the tail beyond the registers is fetched one dword ahead (right for the one-tile kernels).  The pipeline kernels
// pass false: a load still pending at the end of the (rare) tail would make hipcc place a conservative vmcnt(0) on the
// common path too, which there sits behind the next tile's DMA and drains it.
// Four-contact form: compute4(A, B, C, D) handles the two packed pairs of a full step at once, so that the body can
// interleave the two (independent) dependency chains stage by stage in source order; compute2 takes the odd last dword.
template <bool AHEAD = true, typename L, typename C4, typename C2>
__device__ __forceinline__ void for_each_ff4(const StepCtx& c, uint32_t gslice, uint32_t nq, const ListRegs& lr, L&& load,
                                             C4&& compute4, C2&& compute2) {
#pragma unroll
    for (int k = 0; k < LIST_REGS; k += 2) {
        if ((uint32_t)(k + 1) < nq) {
            const uint32_t a = lr.d[k], b = lr.d[k + 1];
            const auto d0 = load(a & 0xffffu);
            const auto d1 = load(a >> 16);
            const auto d2 = load(b & 0xffffu);
            const auto d3 = load(b >> 16);
            compute4(d0, d1, d2, d3);
        } else if ((uint32_t)k < nq) {
            const uint32_t a = lr.d[k];
            const auto d0 = load(a & 0xffffu);
            const auto d1 = load(a >> 16);
            compute2(d0, d1);
        }
    }
    if (nq > (uint32_t)LIST_REGS) {
        const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
        if (AHEAD) {
            uint32_t nx = p[ellq(LIST_REGS)];
            for (uint32_t q = LIST_REGS; q < nq; ++q) {
                const uint32_t a = nx;
                if (q + 1 < nq) nx = p[ellq(q + 1)];
                const auto d0 = load(a & 0xffffu);
                const auto d1 = load(a >> 16);
                compute2(d0, d1);
            }
        } else {
            for (uint32_t q = LIST_REGS; q < nq; ++q) {
                const uint32_t a = p[ellq(q)];
                const auto d0 = load(a & 0xffffu);
                const auto d1 = load(a >> 16);
                compute2(d0, d1);
            }
        }
    }
}
// List entry -> what `load` is handed.  OFF = false: the 16-bit slot.  OFF = true: the slot's BYTE OFFSET in a 16-byte-strided
// LDS array (slot * 16) from ONE instruction: v_mad_u32_u16 multiplies a 16-bit half of its first operand (op_sel picks the
// half) by 16 — the plain C form `(a & 0xffff) << 4` is canonicalised to shift + and, two instructions.
#ifndef SALVA_NO_MAD16
__device__ __forceinline__ uint32_t entry_off16_lo(uint32_t a) { uint32_t o; asm("v_mad_u32_u16 %0, %1, 16, 0" : "=v"(o) : "v"(a)); return o; }
__device__ __forceinline__ uint32_t entry_off16_hi(uint32_t a) { uint32_t o; asm("v_mad_u32_u16 %0, %1, 16, 0 op_sel:[1,0,0,0]" : "=v"(o) : "v"(a)); return o; }
#else
__device__ __forceinline__ uint32_t entry_off16_lo(uint32_t a) { return (a & 0xffffu) << 4; }
__device__ __forceinline__ uint32_t entry_off16_hi(uint32_t a) { return (a >> 16) << 4; }
#endif
__device__ __forceinline__ uint32_t entry_off8_lo(uint32_t a) { uint32_t o; asm("v_mad_u32_u16 %0, %1, 8, 0" : "=v"(o) : "v"(a)); return o; }
__device__ __forceinline__ uint32_t entry_off8_hi(uint32_t a) { uint32_t o; asm("v_mad_u32_u16 %0, %1, 8, 0 op_sel:[1,0,0,0]" : "=v"(o) : "v"(a)); return o; }
// OFF: 0 = the slot, 1 (true) = slot * 16, 2 = slot * 8 (the 8-byte planes of stage_p3)
template <int OFF> __device__ __forceinline__ uint32_t entry_lo(uint32_t a) { return OFF == 1 ? entry_off16_lo(a) : OFF == 2 ? entry_off8_lo(a) : (a & 0xffffu); }
template <int OFF> __device__ __forceinline__ uint32_t entry_hi(uint32_t a) { return OFF == 1 ? entry_off16_hi(a) : OFF == 2 ? entry_off8_hi(a) : (a >> 16); }

// FUSED: a step over two list dwords (four contacts) is ONE basic block — the two packed chains are independent, and only
// inside one block can the scheduler interleave them (each chain alone is ~20 dependent packed operations deep).
// NARROW: one list dword (two contacts) in flight instead of two (four) — 12 VGPRs fewer in the plane layouts, for kernels whose
// per-lane state would otherwise cost them a resident tile (k_iisph_next_pressure_p3).
template <bool AHEAD = true, bool FUSED = false, int OFF = 0, bool NARROW = false, typename L, typename C2>
__device__ __forceinline__ void for_each_ff2(const StepCtx& c, uint32_t gslice, uint32_t nq, const ListRegs& lr, L&& load,
                                             C2&& compute2) {
    if (NARROW) {
#pragma unroll
        for (int k = 0; k < LIST_REGS; ++k) {
            if ((uint32_t)k < nq) {
                const uint32_t a = lr.d[k];
                const auto d0 = load(entry_lo<OFF>(a));
                const auto d1 = load(entry_hi<OFF>(a));
                compute2(d0, d1);
            }
        }
    } else {
#pragma unroll
    for (int k = 0; k < LIST_REGS; k += 2) {
        if (FUSED) {
            if ((uint32_t)(k + 1) < nq) {
                const uint32_t a = lr.d[k], b = lr.d[k + 1];
                const auto d0 = load(entry_lo<OFF>(a));
                const auto d1 = load(entry_hi<OFF>(a));
                const auto d2 = load(entry_lo<OFF>(b));
                const auto d3 = load(entry_hi<OFF>(b));
                compute2(d0, d1);
                compute2(d2, d3);
            } else if ((uint32_t)k < nq) {
                const uint32_t a = lr.d[k];
                const auto d0 = load(entry_lo<OFF>(a));
                const auto d1 = load(entry_hi<OFF>(a));
                compute2(d0, d1);
            }
        } else if ((uint32_t)k < nq) {
            const uint32_t a = lr.d[k];
            const bool two = (uint32_t)(k + 1) < nq;
            const uint32_t b = two ? lr.d[k + 1] : a;
            const auto d0 = load(entry_lo<OFF>(a));
            const auto d1 = load(entry_hi<OFF>(a));
            const auto d2 = load(entry_lo<OFF>(b));
            const auto d3 = load(entry_hi<OFF>(b));
            compute2(d0, d1);
            if (two) compute2(d2, d3);
        }
    }
    }
    if (nq > (uint32_t)LIST_REGS) {  // unusually long lists: the rest comes from memory, one dword ahead
        const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
        if (AHEAD) {
            uint32_t nx = p[ellq(LIST_REGS)];
            for (uint32_t q = LIST_REGS; q < nq; ++q) {
                const uint32_t a = nx;
                if (q + 1 < nq) nx = p[ellq(q + 1)];
                const auto d0 = load(entry_lo<OFF>(a));
                const auto d1 = load(entry_hi<OFF>(a));
                compute2(d0, d1);
            }
        } else {
            for (uint32_t q = LIST_REGS; q < nq; ++q) {
                const uint32_t a = p[ellq(q)];
                const auto d0 = load(entry_lo<OFF>(a));
                const auto d1 = load(entry_hi<OFF>(a));
                compute2(d0, d1);
            }
        }
    }
}
// The same walk with the list position handed to the body: compute2(A, B, q) gets the dword index q (entries 2q and 2q + 1) — a
// compile-time constant in the unrolled part.  For the loops whose summand depends on WHERE in the list a contact sits (two-mass
// worlds: the lighter class first, the heavier behind it; pairs.h).
template <int OFF, typename L, typename C2>
__device__ __forceinline__ void for_each_ff2_indexed(const StepCtx& c, uint32_t gslice, uint32_t nq, const ListRegs& lr, L&& load, C2&& compute2) {
#pragma unroll
    for (int k = 0; k < LIST_REGS; k += 2) {
        if ((uint32_t)k < nq) {
            const uint32_t a = lr.d[k];
            const bool two = (uint32_t)(k + 1) < nq;
            const uint32_t b = two ? lr.d[k + 1] : a;
            const auto d0 = load(entry_lo<OFF>(a));
            const auto d1 = load(entry_hi<OFF>(a));
            const auto d2 = load(entry_lo<OFF>(b));
            const auto d3 = load(entry_hi<OFF>(b));
            compute2(d0, d1, (uint32_t)k);
            if (two) compute2(d2, d3, (uint32_t)(k + 1));
        }
    }
    if (nq > (uint32_t)LIST_REGS) {
        const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
        uint32_t nx = p[ellq(LIST_REGS)];
        for (uint32_t q = LIST_REGS; q < nq; ++q) {
            const uint32_t a = nx;
            if (q + 1 < nq) nx = p[ellq(q + 1)];
            const auto d0 = load(entry_lo<OFF>(a));
            const auto d1 = load(entry_hi<OFF>(a));
            compute2(d0, d1, q);
        }
    }
}
// The first FB_REGS dwords of a particle's fluid-boundary list, loadable before the staging barrier like ListRegs (every
// ELL row has cap_fb >= FB_REGS dwords).  Without boundaries nbr_fb is a dummy: the loads then go to the fluid list.
constexpr int FB_REGS = 8;
struct FbRegs { uint32_t d[FB_REGS]; };
__device__ __forceinline__ FbRegs fb_regs(const StepCtx& c, uint32_t gslice) {
    const uint32_t* __restrict__ p = (c.nb ? c.nbr_fb + (size_t)gslice * c.cap_fb * WAVE : c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE) +
                                     4u * (threadIdx.x & (WAVE - 1));
    static_assert(FB_REGS % 4 == 0, "fetched in 16-byte pieces");
    const uint4* __restrict__ p4 = reinterpret_cast<const uint4*>(p);
    FbRegs r;
#pragma unroll
    for (int k = 0; k < FB_REGS / 4; ++k) {
        const uint4 v = p4[(size_t)k * WAVE];
        r.d[4 * k] = v.x; r.d[4 * k + 1] = v.y; r.d[4 * k + 2] = v.z; r.d[4 * k + 3] = v.w;
    }
    return r;
}
// f(slot) over the fluid-boundary contacts of a particle whose list head is held in registers
template <typename F>
__device__ __forceinline__ void for_each_fb_regs(const StepCtx& c, uint32_t SB, uint32_t gslice, uint32_t cnt, const FbRegs& fr, F&& f) {
    if (SB == 0 || cnt == 0) return;
    const uint32_t nq = (cnt + 1) >> 1;
#pragma unroll
    for (int k = 0; k < FB_REGS; ++k) {
        if ((uint32_t)k < nq) {
            const uint32_t a = fr.d[k];
            f(a & 0xffffu);
            if (2u * (uint32_t)k + 1u < cnt) f(a >> 16);
        }
    }
    if (nq > (uint32_t)FB_REGS) {
        const uint32_t* __restrict__ p = c.nbr_fb + (size_t)gslice * c.cap_fb * WAVE + 4u * (threadIdx.x & (WAVE - 1));
        for (uint32_t q = FB_REGS; q < nq; ++q) {
            const uint32_t a = p[ellq(q)];
            f(a & 0xffffu);
            if (2u * q + 1u < cnt) f(a >> 16);
        }
    }
}
// Exact-count traversal with the list head in registers (the passes that use W itself: a padding entry would count).
// `ListOwn` is loaded before the staging barrier (first_own / for_own_pre), so no list dword is fetched inside the loop
// unless the list is longer than 2 * LIST_REGS contacts.
struct ListOwn { uint32_t cnt; ListRegs lr; };
__device__ __forceinline__ ListOwn list_own(const StepCtx& c, uint32_t i, uint32_t gslice) { return ListOwn{c.nff[i], list_regs(c, gslice)}; }
template <typename L, typename C>
__device__ __forceinline__ void for_each_ff_regs(const StepCtx& c, uint32_t gslice, const ListOwn& o, L&& load, C&& compute) {
    const uint32_t cnt = o.cnt;
#pragma unroll
    for (int k = 0; k < LIST_REGS; ++k) {
        if (2u * (uint32_t)k < cnt) {
            const uint32_t a = o.lr.d[k];
            const auto d0 = load(a & 0xffffu);
            const auto d1 = load(a >> 16);  // (an odd list is padded with the particle's own slot: always a valid read)
            compute(d0);
            if (2u * (uint32_t)k + 1u < cnt) compute(d1);
        }
    }
    if (cnt > 2u * (uint32_t)LIST_REGS) {
        const uint32_t* __restrict__ p = c.nbr_ff + (size_t)gslice * c.cap_ff * WAVE + 4u * (threadIdx.x & (WAVE - 1));
        const uint32_t nq = (cnt + 1u) >> 1;
        uint32_t nx = p[ellq(LIST_REGS)];
        for (uint32_t q = LIST_REGS; q < nq; ++q) {
            const uint32_t a = nx;
            if (q + 1 < nq) nx = p[ellq(q + 1)];
            const auto d0 = load(a & 0xffffu);
            const auto d1 = load(a >> 16);
            compute(d0);
            if (2u * q + 1u < cnt) compute(d1);
        }
    }
}
template <typename F>
__device__ __forceinline__ void for_each_ff_regs(const StepCtx& c, uint32_t gslice, const ListOwn& o, F&& f) {
    for_each_ff_regs(c, gslice, o, [](uint32_t s) { return s; }, f);
}
// Wave-uniform list length of a slice in dwords (k_nbr_tile pads the shorter lists with self contacts).  Call from all lanes.
__device__ __forceinline__ uint32_t slice_list_dwords(uint32_t cnt, bool active) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(active ? ((cnt + 1) >> 1) : 0u));
}
template <typename L, typename C>
__device__ __forceinline__ void for_each_ff(const StepCtx& c, uint32_t i, uint32_t gslice, L&& load, C&& compute) {
    for_each_slot(c.nbr_ff, c.cap_ff, gslice, c.nff[i], load, compute);
}
// single-lambda form (no load/compute split): f(slot)
template <typename F>
__device__ __forceinline__ void for_each_ff(const StepCtx& c, uint32_t i, uint32_t gslice, F&& f) {
    for_each_slot(c.nbr_ff, c.cap_ff, gslice, c.nff[i], [](uint32_t s) { return s; }, f);
}
template <typename F>
__device__ __forceinline__ void for_each_fb(const StepCtx& c, const Tile& t, uint32_t i, uint32_t gslice, F&& f) {
    if (t.SB == 0) return;
    for_each_slot(c.nbr_fb, c.cap_fb, gslice, c.nfb[i], [](uint32_t s) { return s; }, f);
}

__device__ __forceinline__ bool is_ghost(const StepCtx& c, uint32_t i) { return c.gtag && (c.gtag[i] & 0x80000000u); }

// Per-fluid error sums of one tile (par_reduce_sum!, lib.rs:75-83; the per-fluid average is taken by
// k_finalize_error).  Each wave accumulates its slices, in order, into its own LDS row; the rows are then folded in
// wave order, so the result does not depend on scheduling.
constexpr int MAX_MODELS = 32;
struct TileErr {
    float (*tab)[MAX_MODELS];
    __device__ __forceinline__ void init(float (*t)[MAX_MODELS], const StepCtx& c) {
        tab = t;
        const uint32_t lane = threadIdx.x & (WAVE - 1);
        if (lane < c.nmodels) tab[threadIdx.x / WAVE][lane] = 0.0f;
    }
    // wave-uniform call
    __device__ __forceinline__ void add(const StepCtx& c, float err, uint32_t mi, bool active) {
        const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
        if (c.nmodels == 1) {
            const float s = wave_sum(active ? err : 0.0f);
            if (lane == 0) tab[wv][0] += s;
        } else {
            for (uint32_t m = 0; m < c.nmodels; ++m) {
                const float s = wave_sum((active && mi == m) ? err : 0.0f);
                if (lane == 0) tab[wv][m] += s;
            }
        }
    }
    __device__ __forceinline__ void finish(const StepCtx& c, uint32_t tile) {
        __syncthreads();
        if (threadIdx.x < c.nmodels) {
            float s = 0.0f;
            for (int w = 0; w < (int)(blockDim.x / WAVE); ++w) s += tab[w][threadIdx.x];
            c.partials[(size_t)tile * c.nmodels + threadIdx.x] = s;
        }
    }
    static __device__ __forceinline__ void zero(const StepCtx& c, uint32_t tile) {
        if (threadIdx.x < c.nmodels) c.partials[(size_t)tile * c.nmodels + threadIdx.x] = 0.0f;
    }
};

// The same with rows of nmodels floats instead of MAX_MODELS (48 bytes for one fluid instead of 1536: the plane-layout kernels
// count their LDS in hundreds of bytes, tile.h P3_DS_THREE).  Same sums in the same order as TileErr.
struct TileErrC {
    float* tab;
    uint32_t nm;
    static __device__ __forceinline__ uint32_t bytes(const StepCtx& c) { return (TILE_MAX_WAVES * c.nmodels * 4u + 15u) & ~15u; }
    __device__ __forceinline__ void init(float* t, const StepCtx& c) {
        tab = t; nm = c.nmodels;
        const uint32_t lane = threadIdx.x & (WAVE - 1);
        if (lane < nm) tab[(threadIdx.x / WAVE) * nm + lane] = 0.0f;
    }
    __device__ __forceinline__ void add(const StepCtx& c, float err, uint32_t mi, bool active) {
        const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
        if (nm == 1) {
            const float s = wave_sum(active ? err : 0.0f);
            if (lane == 0) tab[wv] += s;
        } else {
            for (uint32_t m = 0; m < nm; ++m) {
                const float s = wave_sum((active && mi == m) ? err : 0.0f);
                if (lane == 0) tab[wv * nm + m] += s;
            }
        }
    }
    __device__ __forceinline__ void finish(const StepCtx& c, uint32_t tile) {
        __syncthreads();
        if (threadIdx.x < nm) {
            float s = 0.0f;
            for (int w = 0; w < (int)(blockDim.x / WAVE); ++w) s += tab[(uint32_t)w * nm + threadIdx.x];
            c.partials[(size_t)tile * nm + threadIdx.x] = s;
        }
    }
};

// Boundary::apply_force (boundary.rs:62-67): forces accumulate in canonical boundary order.  `jb` is the sorted
// boundary index (t.bgstart-relative lookups are done by the caller).  Callers skip ghost particles (is_ghost): in a
// decomposed run the boundary particles near a slab face exist on both ranks, and a reaction force belongs to the rank that
// owns the fluid particle — summed over ranks the forces then equal the single-domain run's.
__device__ __forceinline__ void apply_boundary_force(const StepCtx& c, uint32_t jb_sorted, uint32_t bmodel, float fx,
                                                     float fy, float fz) {
    if (c.bforce == nullptr || !c.bwants[bmodel]) return;
    float* f = reinterpret_cast<float*>(&c.bforce[c.bperm[jb_sorted]]);
    atomicAdd(f + 0, fx);
    atomicAdd(f + 1, fy);
    atomicAdd(f + 2, fz);
}
__device__ __forceinline__ uint32_t boundary_sorted_of_slot(const StepCtx& c, const Tile& t, uint32_t slot) {
    return c.bhalo_src[t.hboff + slot];
}

#endif  // __HIPCC__
}  // namespace salva
