// device_types.h — kernel argument bundles and the launcher interface between world.hip (host
// orchestration) and the kernel translation units (grid.hip, dfsph.hip, iisph.hip, forces.hip).
//
// Data layout in HBM: every per-particle array is SoA over the *cell-sorted* particle
// order; 3-vectors are float4 so that one 16-byte gather fetches everything a neighbour contributes:
//   posm = (x, y, z, mass)            gathered by every neighbour pass
//   w    = (v+dv x, y, z, model id)   gathered by the evaluate passes and by the viscosity/tension forces
//   vel  = (v x, y, z, volume)        streamed only
//   dv   = (dv x, y, z, pressure)     streamed only (pressure = IISPH warm start, carried across steps)
// Neighbour lists are ELL blocks per slice (64 consecutive particles of one tile = one wavefront): entries 2q and
// 2q+1 (16-bit LDS slots of the tile's halo, tile.h) of the particle handled by lane l of slice s live in the dword
// nbr[(s * cap + q) * 64 + l], so a wave reads 256 contiguous bytes per pair of contacts.  `cap` (dwords per
// particle) is fixed per step, which lets the list be written in the same pass that finds the contacts; rows
// beyond a slice's longest list are never touched.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sph_math.h"

namespace salva {

// Dense cell table over the (tile-aligned) bounding box of one particle class.  Cells are keyed tile-major:
// key = tile_linear * TCELLS (64) + cell_in_tile, tile_linear = (tx * nty + ty) * ntz + tz (x slowest), cell_in_tile =
// (ux * TY + uy) * TZ + uz, with (cx,cy,cz) = floor(p / h) (hgrid.rs:41-52) relative to the origin (ox,oy,oz) = the
// minimum corner of the occupied cells' bounding box.  cell_start has ncells+1 entries with lower-bound semantics:
// cell_start[k] = first sorted index whose key >= k, so cell k is [cell_start[k], cell_start[k+1]) and a whole tile
// is [cell_start[64 t], cell_start[64 t + 64)).
//
// Folding (world.hip dims_from_bbox): the reference's grid is a hash map and costs nothing per EMPTY cell; this table costs four bytes
// for every cell of the bounding box, and a few particles that have left the scene (an open tank leaks, a faucet is never stopped)
// blow the box up without bound.  When the box holds far more cells than particles, the axes are therefore FOLDED: the cell
// coordinate relative to the origin is taken modulo a power-of-two period P (mx / my / mz = P - 1; 0xffffffff = the axis is not
// folded), so the table is a torus of P cells in that axis and cells P apart share an entry.  Cells next to each other stay next to
// each other (the halo of a tile wraps around), so every true neighbour is still a candidate; the particles of the other images
// that share a cell are candidates too and fail the exact d^2 <= h^2 test on their true positions like any other non-neighbour —
// the contact SETS are the unfolded grid's.  (The order inside a cell is the order of the previous step's sort, which follows the
// numbering of the tiles: a folded and an unfolded run of the same scene agree bit for bit in their first step and to summation
// order afterwards — like two runs whose boxes have different origins.)
struct TileGrid {
    int ox, oy, oz;
    int ntx, nty, ntz;
    uint32_t mx, my, mz;
    const uint32_t* cell_start;
};

// per-tile sizes, prefix-summed over tiles in one scan
struct TileAcc {
    uint64_t s;     // fluid halo slots
    uint64_t sb;    // boundary halo slots
    uint32_t nsl;   // 64-particle slices
    uint32_t nonempty;  // tiles that own at least one particle
    uint32_t max_s, max_sb, max_nsl;  // running maxima of the three (carried through the same scan)
    uint32_t max_sum;  // running maximum of (fluid halo slots padded to 64) + (boundary halo slots) of one tile (TileLds::max_sum)
    uint32_t max_raw;  // running maximum of (fluid halo slots) + (boundary halo slots) of one tile, no padding: the plane layouts
                       // (tile.h stage_p3), which are filled through registers, slot by slot
    uint32_t wsl;      // sum of (slices of the tile)^2: wsl / nsl = the slice count of the tile the average PARTICLE lives in (world.hip:
                       // the workgroup size)
    uint32_t ntiny;    // slots of the SPARSE class (tile.h TILE_TINY_*: one slice of own particles, a halo of a few cells' worth): a
                       // prefix gives a sparse slot its rank among its kind, the total decides whether they get a launch of their own
    uint32_t heavy;    // slots that are PARTS of a split tile + whole tiles whose fluid halo is beyond the three-tiles-per-CU layouts
                       // (tile.h TILE_SPLIT_S): what World::substep decides the next step's splitting by
    uint32_t nlight;   // slots of the LIGHT class (tile.h tile_is_light: not sparse, and the halo fits the smallest compile-time layout
                       // of every kernel family): when other slots of the step do not, these get a launch of their own on that layout
    __host__ __device__ TileAcc operator+(const TileAcc& o) const {
        return TileAcc{s + o.s, sb + o.sb, nsl + o.nsl, nonempty + o.nonempty, max_s > o.max_s ? max_s : o.max_s,
                       max_sb > o.max_sb ? max_sb : o.max_sb, max_nsl > o.max_nsl ? max_nsl : o.max_nsl,
                       max_sum > o.max_sum ? max_sum : o.max_sum, max_raw > o.max_raw ? max_raw : o.max_raw, wsl + o.wsl, ntiny + o.ntiny, heavy + o.heavy, nlight + o.nlight};
    }
};

// Device-resident control block of an iterative solve (`for i in 0..max { err = evaluate(); if err <= tol && i >= min
// { break }; apply(); }`, dfsph_solver.rs:439-463, :474-502).  The host enqueues several iterations at once; once
// `done` is set every later evaluate / finalize / apply kernel of the batch returns immediately, so the protocol is
// exactly the reference's while the host reads the block back once per batch instead of once per iteration.
struct alignas(16) SolveCtl {
    // --- the first 16 bytes are what the host needs: published to host-mapped memory with ONE 16-byte store per test
    uint32_t done;       // set by k_finalize_error when the break condition holds
    uint32_t iters;      // applies executed (DFSPH) / Jacobi iterations completed (IISPH)
    float err;           // last evaluated error
    uint32_t seq;        // convergence tests executed so far in this solve (the host waits for the count it enqueued)
    // --- parameters of the solve
    float tol;
    uint32_t min_iter;
    uint32_t mode;       // 0: DFSPH protocol (test, then count the apply); 1: IISPH (count the iteration, then test)
    uint32_t pad;
};
static_assert(sizeof(SolveCtl) == 32, "SolveCtl: two 16-byte halves");

// per-tile list statistics of k_nbr_tile (grid.hip), folded by k_list_stats or by the end-of-step publication
struct TileListStats { uint32_t sum_ff, sum_fb, max_ff, max_fb, own_ff, own_fb; };  // own_*: lists of particles this rank OWNS (no ghosts)

// what the end-of-step publication needs to decide Readback::pre_ok
struct PrePub { int32_t on, chained; int32_t bbox[6]; };

struct StepCtx {
    SphConsts sc;
    uint32_t xcd;  // workgroup -> slot mapping: 1 + log2 of the consecutive slots one XCD takes at a time (common.h xcd_block); 0 = off

    // ---- fluid particles, cell-sorted order ----
    uint32_t n;
    float4* posm;
    float4* vel;
    float4* dv;
    float4* acc;
    float4* w;
    float4* normal;      // Akinci normals (xyz, unused)
    uint32_t* model;
    uint32_t* perm;      // sorted index -> canonical (host-order) index; in a multi-GPU run: global particle id
    const uint32_t* gtag; // multi-GPU runs only (else nullptr): ghost / border tags, dist.h
    float* rho;
    float* alpha;
    float* kappa;        // DFSPH: div*alpha or (rho* - rho0)*alpha ; IISPH: pressure p
    float* kappa2;       // IISPH: next pressure
    float* rho_star;     // IISPH predicted density
    float* aii;          // IISPH
    float4* dii;         // IISPH (xyz, unused)
    float4* dijpj;       // IISPH sum_j d_ij p_j (xyz, unused)
    float4* posmr;       // (x, y, z, m / rho) of this step (k_density_alpha): the neighbour record of volume-weighted sums (k_xsph)
    float4* iisph_q;     // IISPH d_ii p_i + sum_j d_ij p_j: what a neighbour contributes to compute_next_pressures in one record
    float4* iisph_pr;    // IISPH (x, y, z, m / rho^2): what a neighbour contributes to compute_dij_pjl besides its pressure (k_iisph_dii)
    uint32_t* nff;       // # fluid-fluid contacts of each particle (self included)
    uint32_t* nfb;       // # fluid-boundary contacts
    uint32_t* nbr_ff;           // packed 16-bit halo slots: slice s owns dwords [s*cap_ff*64, (s+1)*cap_ff*64)
    uint32_t* nbr_fb;
    uint32_t* slice_near;       // per slice: some particle has a neighbour closer than 1e-5 h (k_density_alpha; dfsph.hip)
    uint32_t cap_ff, cap_fb;    // dwords (= pairs of contacts) reserved per particle
    const TileAcc* tile_off;    // [nslots+1] exclusive prefix of per-tile {halo slots, boundary halo slots, slices}, by SLOT
    const uint32_t* halo_src;   // sorted fluid index of every halo slot of every tile (tile-major)
    const uint32_t* bhalo_src;  // sorted boundary index of every boundary halo slot
    uint32_t halo_stride;       // > 0: tile t's slot table starts at t * halo_stride (address known before any load
    uint32_t bhalo_stride;      //      returns); 0: compact tables at tile_off[t].s / .sb
    uint32_t ntiles;            // tiles of the dense grid (only the cell table and one flag per tile are dense)
    // > 0: a tile whose fluid halo holds more than this many particles is cut along x into halves (own cells ux 0-1 | 2-3, four of
    // the six halo planes each) or quarters (one plane of own cells, three halo planes), each part a slot of its own (tile.h
    // Tile::part) — so that a few over-full tiles at the bottom of a tank do not move EVERY tile of every pass to the two-tiles-per-CU
    // layouts (round 6).  0: every non-empty tile is one slot.
    uint32_t split_s;
    // Everything per tile lives in a compact table over the NON-EMPTY tiles ("slots", in dense-index order), and every
    // tile kernel is launched over slots: widely scattered particles then cost nothing per empty tile.
    const uint32_t* tile_ids;   // [nlaunch] dense tile index of slot k | its part code << 28 (tile.h Tile::part)
    const uint4* slot_desc;     // [nlaunch] {dense tile index, first own particle, one past the last, part code}: everything
                                //           Tile::setup needs besides tile_off, in one load that depends on nothing
    const uint4* slot_info;     // [nlaunch] {first own particle, one past the last, first slice, S | SB << 16}: all a solver
                                //           kernel needs to know about a tile's sizes, written by k_tile_halo_fill
    const uint32_t* tile_rank;  // [ntiles+1] exclusive prefix of the tiles' slot counts: FIRST slot of a dense tile; [ntiles] = nlaunch
    // Launch classes per pass (round 6).  SPARSE: when thousands of stray particles own a tile each, those slots —
    // one slice, a halo of a few particles — run in a launch of their own, 64 threads and a few KB of LDS per workgroup (two dozen
    // per CU), instead of holding a full tile's LDS for a list of one entry, three per CU.  LIGHT: when some halos of the step are
    // beyond the three-per-CU layouts (the settled bench scene: half of its tiles), the slots that are not run in a launch of their
    // own on those layouts instead of sharing the two-per-CU launch of the full ones.  slot_order = [the full slots | the light slots |
    // the sparse slots], each kind in slot order (k_tile_halo_fill); a launch works on slot_order[slot_base + mapped block].
    // nullptr: one class, block -> slot directly.
    const uint32_t* slot_order;
    uint32_t slot_base;
    uint32_t ntiny;             // sparse slots of this step when they have their own launches (0: they run with the others)
    uint32_t nlight;            // light slots of this step when they have their own launches (0: they run with the full ones)
    uint32_t nlaunch;           // number of slots launched (= the number of non-empty tiles, or in a speculative pass an upper bound)
    // Speculative passes (World::step): launch shapes and buffers were cut from the previous step's totals, so every
    // tile kernel clamps itself to what it was given — surplus slots are empty, a halo is cut at the staged capacity, a tile
    // whose slices would not fit the list buffer is skipped.  Results are then wrong, and the host (which compares the true
    // totals with its prediction at the end of the step) discards the pass and repeats it with exact sizes.
    uint32_t spec;              // 0: exact sizes (no clamping needed)
    uint32_t halo_cap, bhalo_cap;   // slots a tile may stage
    uint32_t nslices_cap;           // slices the list buffers hold
    uint64_t halo_len, bhalo_len;   // entries of halo_src / bhalo_src
    TileGrid gf;
    // non-null when a DynamicContactSampling boundary pushed particles after the cell keys were taken: the sorted keys, from
    // which the neighbour search takes each particle's (possibly stale) cell instead of its position — the reference does
    // not re-insert pushed particles into its grid either (liquid_world.rs:90-106)
    const uint32_t* stale_keys;

    // ---- boundary particles, cell-sorted order ----
    uint32_t nb;
    float4* bposv;       // (x, y, z, volume V_b)
    float4* bvel;        // (vx, vy, vz, boundary model id)
    uint32_t* bperm;     // sorted -> canonical boundary index
    float4* bforce;      // canonical order accumulators (nullptr if no boundary wants forces)
    const uint8_t* bwants;  // per boundary model: forces requested?
    TileGrid gb;

    // ---- per-model tables ----
    uint32_t nmodels, nbmodels;
    const float* rho0_tab;     // density0 of each fluid model
    float rho0_single;         // = rho0_tab[0]; used when nmodels == 1 (saves a dependent load)
    // > 0: EVERY fluid particle of the working set (ghosts included) has exactly this mass (k_cell_keys reduces min / max of
    // posm.w every step): the evaluate kernels then stage 24 bytes per halo slot instead of 32 (no mass in LDS; tile.h stage_p3)
    // and multiply the finished sum by it.  0: masses differ (non-uniform volumes, fluids of different density0), or unknown.
    float mass_uniform;
    // Worlds with a few particle masses (BASELINE config 4: two fluids of different density0, each with `Fluid::new`'s uniform volumes;
    // round 6: up to FOUR masses): the host knows the masses (`class_mass`, ascending) and which fluid carries which (`cmask`, two bits
    // per fluid).  k_nbr_tile then writes the lists of a tile whose halo holds several with the lightest class first and the heavier
    // ones behind it, class by class (nffb[i] = entries behind the first segment; nffc[i] = lengths of the third and fourth segment)
    // and, per slot, the masses of the segments (0: there is no such segment).  The plane-layout kernels run over ALL tiles in one
    // launch and compute m_a S_all + (m_b - m_a) S_b (+ (m_c - m_b) S_c + (m_d - m_b) S_d), the later sums over the tail segments only
    // and only in the tiles that have them (pairs.h pair_tail_*).
    uint32_t* tile_mass_bits;   // [nlaunch] by slot: bits of the mass of the list's first segment (written by k_nbr_tile)
    uint32_t* tile_massb_bits;  // [nlaunch] by slot: bits of the mass of the second segment, 0 = there is none
    uint2* tile_masscd_bits;    // [nlaunch] by slot: ... of the third and fourth segment (worlds with more than two masses only, else nullptr)
    uint32_t* nffb;             // [n] number of entries behind the first segment of particle i's list
    uint32_t* nffc;             // [n] entries of the third segment | of the fourth << 16 (worlds with more than two masses only)
    uint32_t two_mass;          // 1: the above is on (DFSPH, default kernels, single domain)
    uint32_t nmass;             // number of masses (2 ... 4) when it is
    uint64_t cmask;             // mass class of fluid f: (cmask >> 2 f) & 3
    float class_mass[4];        // the masses, ascending
    uint32_t bvel_zero;        // 1: every boundary velocity is exactly zero (plain boundaries uploaded at rest, the usual tank)
    const uint8_t* ff_ok;      // [nmodels*nmodels] InteractionGroups::test between fluids (diagonal = 1)
    const uint8_t* fb_ok;      // [nmodels*nbmodels]
    const uint8_t* bb_ok;      // [nbmodels*nbmodels] (diagonal = 1)

    // ---- reductions / flags ----
    float* partials;     // [nblocks * nmodels] per-block error sums
    // Speculative divergence applies (dfsph.hip k_divergence_apply, World::dfsph_solve): iteration k of the solve reads record
    // spec_ring[k & 1] and the apply pass's workgroup 0 writes spec_ring[(k + 1) & 1]; w lives in w / w2 by the parity of the
    // committed applies.  spec_k < 0: off (the control block `ctl` and one k_finalize_error launch per iteration).
    int spec_k;
    // 1: a DECOMPOSED solve runs its applies speculatively (round 6): the records are written by k_decide_ring behind the all-reduce,
    // on the main stream, while the apply pass runs beside it on the second stream — the apply's workgroup 0 decides nothing
    uint32_t spec_external;
    SolveCtl* spec_ring;
    SolveCtl* spec_pub;            // host-mapped copy of the first half of the newest record (may be null)
    const uint32_t* model_counts;  // particles per fluid: the denominators of the error averages
    float4* w2;
    uint32_t* flags;     // bit 0: numeric error (zero density / NaN), bit 1: particle outside grid
    uint32_t min_neighbors_for_divergence;
    // Decomposed runs: an evaluate pass may be launched twice — once over the tiles whose halo box touches no ghost plane
    // (phase 1, on a second stream, while the ghosts' fields are still in flight) and once over the rest (phase 2, after
    // the exchange).  0: every tile.  ghost_lo_cx / ghost_hi_cx: the cell planes next to the slab that hold ghosts.
    int32_t phase, ghost_lo_cx, ghost_hi_cx;
    const SolveCtl* ctl;       // non-null inside an iterative solve: kernels return at once when ctl->done
    // Chained steps (World::dfsph_solve, round 6): the host enqueues the divergence solve's batch, the kernels between the two
    // solves, the pressure solve's batch and the end of the step WITHOUT waiting for either solve's outcome.  Everything behind a
    // solve is then "gated": non-null, and the word it points to (Readback::chain_ok) is 0 when a solve upstream did not converge
    // within the batch it was given — such a kernel returns at once, the end-of-step publication says where the chain broke, and
    // the host continues from there the classic way (enqueue, wait, decide).  nullptr: not gated.
    const uint32_t* gate;
    // the kernels of a chained SOLVE carry its number here (1 = divergence, 2 = pressure; 0 = behind every solve): when the gate was
    // shut by the last test of this very solve, the apply pass that test has just counted still runs — only a solve upstream stops them
    uint32_t gate_stage;
#ifdef SALVA_HIP_DIAG
    unsigned long long* dbg;   // kernel-development builds: per-tile phase timestamps (k_pred_density, SALVA_HIP_TILE_TIMING=1)
#endif
};

constexpr uint32_t MASS_SLOTS = 64;  // flag words k_cell_keys spreads "a mass differs from particle 0's" over (grid.hip)

// Result block the host reads back (pinned mirror).
struct Readback {
    float err;            // max over models of (sum / nparticles)
    uint32_t flags;
    int32_t bbox[6];      // fluid cell bbox (min xyz, max xyz)
    int32_t bbbox[6];     // boundary cell bbox
    TileAcc tile_total;   // totals and maxima of halo slots / boundary halo slots / slices over the tiles
    uint64_t ncontacts_bb;
    uint64_t ncontacts_ff, ncontacts_fb;   // } written by k_list_stats and read back in one copy
    uint32_t max_cnt_ff, max_cnt_fb;       // } longest contact lists of the step (capacity check)
    uint32_t dcs_count;                // points emitted by the last DynamicContactSampling pass
    uint32_t cfl_max_bits;             // bits of max |v + a t|^2 over the fluid particles (World::choose_substep, opt-in CFL sub-stepping)
    uint64_t ncontacts_own_ff, ncontacts_own_fb;  // list totals over the particles this rank owns (decomposed runs)
    uint32_t mass_mm[2];  // [0] = bits of particle 0's mass, [1] = the same if no particle's mass differed since the last publication
                          // of the totals (host side of the publication only; on the device: World::mass_slots, grid.hip k_cell_keys)
    uint32_t pad2_[2];
    // chained steps (StepCtx::gate): 1 while every solve of the step has converged within the batch the host enqueued for it; the
    // last convergence test of a batch that fails clears it and leaves the solve's number (1 = divergence, 2 = pressure) in
    // chain_stage.  solve[k] = {done, iters, err, seq} of d_ctl[k] at the end-of-step publication.
    uint32_t chain_ok, chain_stage;
    uint32_t solve[2][4];
    // the grid part of the NEXT step enqueued at the end of this one (World::pre_enqueue_grid): 1 when the box the position update
    // found is the box that part was enqueued for (and the chain held) — the gate of those launches
    uint32_t pre_ok;
    uint32_t pad3_;
};
#ifdef __HIPCC__
// gate[0] = Readback::chain_ok (or pre_ok), gate[1] = the word behind it (chain_stage)
__device__ __forceinline__ bool gate_words_closed(uint32_t ok, uint32_t stage, uint32_t my_stage) { return ok == 0u && !(my_stage != 0u && stage == my_stage); }
__device__ __forceinline__ bool gate_closed(const StepCtx& c) { return c.gate && gate_words_closed(c.gate[0], c.gate[1], c.gate_stage); }
#endif

}  // namespace salva
