// kernels.h — launcher interface implemented by grid.hip / dfsph.hip / iisph.hip / forces.hip.
#pragma once
#include "common.h"
#include "device_types.h"
#include "tile.h"
#ifdef SALVA_HIP_DIAG
#include "diag/pipe.h"
#endif

namespace salva {

// ---------------------------------------------------------------- grid.hip (replaces geometry::HGrid + contacts)
// cell bbox of `n` points (xyz of float4) -> bbox6 = {min x,y,z, max x,y,z}; partials needs 6 * bbox_blocks(n) ints
unsigned bbox_blocks(uint32_t n);
void launch_bbox(const float4* pts, uint32_t n, float h, int32_t* partials, int32_t* bbox6, uint32_t* flags, hipStream_t s);
void launch_bbox_final(const int32_t* partials, unsigned nblocks, int32_t* bbox6, hipStream_t s, const uint32_t* gate = nullptr);
// counts != nullptr: the first half of the counting sort (cell_sort) — counts[key] += 1, rank[i] = the value before; idx is not written
void launch_cell_keys(const float4* pts, uint32_t n, float h, TileGrid g, uint32_t* keys, uint32_t* idx,
                      uint32_t* flags, uint32_t* mass_mm, uint32_t* counts, uint32_t* rank, hipStream_t s, const uint32_t* gate = nullptr);
size_t cell_sort_temp_bytes(uint32_t ncells);
void cell_sort(void* temp, size_t temp_bytes, uint32_t n, uint32_t ncells, const uint32_t* keys, const uint32_t* rank, uint32_t* cell_start,
               uint32_t* keys_out, uint32_t* idx_tmp, uint32_t* idx_out, hipStream_t s, const uint32_t* gate = nullptr);
size_t sort_pairs_temp_bytes(uint32_t n, int end_bit);
void sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in,
                uint32_t* idx_out, uint32_t n, int end_bit, hipStream_t s);
void launch_cell_start(const uint32_t* keys_sorted, uint32_t n, uint32_t ncells, uint32_t* cell_start, hipStream_t s);

struct FluidArrays {
    float4 *posm, *vel, *dv;
    uint32_t *model, *perm;
    uint32_t* gtag;  // nullptr outside multi-GPU runs
};
void launch_reorder_fluid(uint32_t n, const uint32_t* idx, FluidArrays in, FluidArrays out, float4* w, hipStream_t s);
void launch_reorder_boundary(uint32_t n, const uint32_t* idx, const float4* bpos_in, const float4* bvel_in,
                             float4* bposv_out, float4* bvel_out, uint32_t* bperm_out, hipStream_t s);
// canonical (host order) staging <-> sorted working set
void launch_stage_to_sorted(uint32_t n, const float4* st_pos /*xyz,vol*/, const float4* st_vel, const float4* st_dv,
                            const uint32_t* st_model, const float* rho0_tab, FluidArrays out, hipStream_t s);
void launch_sorted_to_stage(uint32_t n, FluidArrays in, float4* st_pos, float4* st_vel, float4* st_dv, hipStream_t s);
void launch_unsort_f32(uint32_t n, const uint32_t* perm, const float* in, float* out, hipStream_t s);
void launch_export_contacts(const StepCtx& c, const uint32_t* keys, uint32_t slot, int boundary, const uint64_t* offsets,
                            const uint32_t* model_off, const uint32_t* bmodel_off, uint32_t* out_model, uint32_t* out_j, hipStream_t s);
void launch_export_contacts_local(const StepCtx& c, const uint32_t* keys, int boundary, const uint64_t* offsets, const uint32_t* bmodel_off,
                                  uint32_t* out_model, uint32_t* out_j, hipStream_t s);
void launch_unsort_u32(uint32_t n, const uint32_t* perm, const uint32_t* in, uint32_t* out, hipStream_t s);
void launch_unsort_u32_as_f32(uint32_t n, const uint32_t* perm, const uint32_t* in, float* out, hipStream_t s);
void launch_unsort_f4(uint32_t n, const uint32_t* perm, const float4* in, float4* out, hipStream_t s);
void launch_gather_f4(uint32_t n, const uint32_t* perm, const float4* in, float4* out, hipStream_t s);

// per-tile {halo slots, boundary halo slots, slices} (-> scan_tiles -> tile_off) and their maxima; then the flat
// halo slot tables
// g: the fluid grid WITH its cell table; split_s: StepCtx::split_s (0 = one slot per non-empty tile)
void launch_tile_slots(TileGrid g, uint32_t ntiles, uint32_t* flags, uint32_t* rank, uint32_t* tile_ids, void* temp,
                       size_t temp_bytes, hipStream_t s, const uint32_t* gate = nullptr, uint32_t split_s = 0u);
void launch_tile_count(const StepCtx& c, uint32_t nslots_bound, TileAcc* tile_cnt, uint4* slot_desc, hipStream_t s);
// slot_order != nullptr: also the order of the step's launch classes (StepCtx::slot_order; nlight / ntiny: the light / sparse slots when
// they get launches of their own, else 0)
void launch_tile_halo_fill(const StepCtx& c, uint32_t* halo_src, uint32_t* bhalo_src, uint4* slot_info, hipStream_t s, uint32_t* slot_order = nullptr,
                           uint32_t nlight = 0u, uint32_t ntiny = 0u);
size_t scan_tiles_temp_bytes(uint32_t n);
void scan_tiles(void* temp, size_t temp_bytes, const TileAcc* in, TileAcc* out, uint32_t n, hipStream_t s);
// neighbour lists (per-slice ELL blocks of 16-bit halo slots, capacity c.cap_ff / c.cap_fb dwords per particle), built
// in one pass; totals2 = {ff, fb} contact counts, maxima2 = longest {ff, fb} list (> 2*cap means overflow: rebuild)
size_t tile_list_stats_bytes(uint32_t ntiles);
// own2 (may be null) = the same totals over the particles this rank owns (no ghosts)
void launch_nbr_build(const StepCtx& c, const TileLds& L, void* tile_stats, unsigned long long* totals2, uint32_t* maxima2,
                      unsigned long long* own2, hipStream_t s);
size_t select_flagged_temp_bytes(uint32_t n);
void select_flagged_f4(void* temp, size_t temp_bytes, const float4* in, const uint8_t* flags, float4* out, uint32_t* num_selected,
                       uint32_t n, hipStream_t s);
size_t scan_temp_bytes(uint32_t n);
void scan_u64(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, uint32_t n, hipStream_t s);

// V_b = 1 / sum_b' W_bb' (dfsph_solver.rs:72-96); also counts boundary-boundary contacts
void launch_boundary_volumes(const StepCtx& c, unsigned long long* ncontacts_bb, hipStream_t s);

// ---------------------------------------------------------------- dfsph.hip
unsigned num_blocks(uint32_t n);
// iisph_dt > 0: the IISPH form — d_ii, p = p_prev / 2 and the dij record ride along (no k_iisph_dii), no alpha
void launch_density_alpha(const StepCtx& c, const TileLds& L, float iisph_dt, hipStream_t s);
// density + alpha + the first divergence evaluate of the step in one pass (DFSPH, uniform mass, default kernels): false = not taken
bool launch_density_alpha_div(const StepCtx& c, const TileLds& L, hipStream_t s);
void launch_divergence(const StepCtx& c, const TileLds& L, hipStream_t s);                 // -> kappa = div*alpha, partials
void launch_divergence_apply(const StepCtx& c, const TileLds& L, float inv_dt_prev, hipStream_t s);
void launch_finish_divergence(const StepCtx& c, float gx, float gy, float gz, bool acc_has_user, hipStream_t s);
void launch_integrate(const StepCtx& c, float dt, hipStream_t s);       // dv += acc*dt ; acc = 0 ; w = vel + dv
// visc.hip — DFSPHViscosity
void launch_visc_betas(const StepCtx& c, const TileLds& L, uint32_t model, float* beta, hipStream_t s);
void launch_visc_va(const StepCtx& c, float dt_prev, float4* va, hipStream_t s);
void launch_visc_strain(const StepCtx& c, const TileLds& L, uint32_t model, int mode, float coef, const float4* va,
                        const float* beta, float* target, float4* u0, float4* u1, hipStream_t s);
void launch_visc_accel(const StepCtx& c, const TileLds& L, uint32_t model, float inv_dt_prev, float dt_prev, const float4* u0,
                       const float4* u1, float4* va, hipStream_t s);
void launch_pred_density(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);    // -> kappa = (rho*-rho0)*alpha, partials
void launch_pressure_apply(const StepCtx& c, const TileLds& L, float inv_dt, hipStream_t s);
// x += w dt; per-block cell bounds into bbox_partials (6 * num_blocks(n) ints), folded into bbox6
void launch_update_positions(const StepCtx& c, float dt, int32_t* bbox_partials, int32_t* bbox6, hipStream_t s);
// err = max_m (sum_b partials[b][m] / count[m]) -> ctl->err, then the break test of the solve (nblocks = ntiles)
// gate / close / close_stage: chained steps (device_types.h StepCtx::gate) — skip when *gate == 0; a failed test clears close[0] and
// leaves close_stage in close[1]
void launch_finalize_error(const float* partials, unsigned nblocks, uint32_t nmodels, const uint32_t* model_counts,
                           SolveCtl* ctl, SolveCtl* pub, hipStream_t s, const uint32_t* gate = nullptr, uint32_t* close = nullptr,
                           uint32_t close_stage = 0u, uint32_t skipped = 0u);
// multi-GPU form: per-fluid sums of this rank -> sums[nmodels]; (all-reduce over ranks); break test on the global sums
void launch_sum_partials(const float* partials, unsigned nblocks, uint32_t nmodels, const SolveCtl* ctl, float* sums, hipStream_t s);
void launch_decide_ring(const float* sums, uint32_t nmodels, const uint32_t* model_counts, SolveCtl* ring, int k, SolveCtl* pub, hipStream_t s);
// skipped: tests of earlier iterations that were not launched because they could not end the solve (i < min_iter): counted here
void launch_decide(const float* sums, uint32_t nmodels, const uint32_t* model_counts, SolveCtl* ctl, SolveCtl* pub, hipStream_t s, uint32_t skipped = 0u);

#ifdef SALVA_HIP_DIAG
// diag/sched.hip: permute every list's real entries so that the ds_read_b128 lane groups hit distinct LDS bank quads (or one
// slot).  Measured (profiles/r03_experiments): -6 % per neighbour pass for ~600 us per step — never pays; kept as an experiment.
void launch_list_schedule(const StepCtx& c, const TileLds& L, hipStream_t s);
// ---------------------------------------------------------------- diag/dfsph_pipe.hip (kernel experiments, `make VARIANT=diag`: persistent tile pipeline, pipe.h)
void launch_pred_density_pipe(const StepCtx& c, const PipeCfg& P, float dt, hipStream_t s);
// diagnostics: variant 0 = one tile per workgroup, 1 = + de-phased co-resident tiles (param = sleep in 64-cycle units),
// 2 = pipeline, 3 = one tile per workgroup with LDS-DMA staging
void launch_pred_density_variant(const StepCtx& c, const TileLds& L, const PipeCfg& P, float dt, int variant, uint32_t param,
                                 uint32_t* cu_arrivals, hipStream_t s);
#endif

// ---------------------------------------------------------------- forces.hip
void launch_xsph(const StepCtx& c, const TileLds& L, uint32_t model, float fluid_coeff, float boundary_coeff, float inv_dt_prev, hipStream_t s);
void launch_artificial_viscosity(const StepCtx& c, const TileLds& L, uint32_t model, float fluid_coeff, float boundary_coeff, float alpha,
                                 float beta, float speed_of_sound, hipStream_t s);
void launch_akinci_normals(const StepCtx& c, const TileLds& L, uint32_t model, hipStream_t s);
void launch_he2014_colors(const StepCtx& c, const TileLds& L, uint32_t model, float* colors, hipStream_t s);
void launch_he2014_gradc(const StepCtx& c, const TileLds& L, uint32_t model, const float* colors, float* gradcs, hipStream_t s);
void launch_he2014_forces(const StepCtx& c, const TileLds& L, uint32_t model, float tension, float boundary_tension,
                          const float* gradcs, hipStream_t s);
void launch_wcsph_tension(const StepCtx& c, const TileLds& L, uint32_t model, float tension, hipStream_t s);
void launch_akinci_forces(const StepCtx& c, const TileLds& L, uint32_t model, float tension, float adhesion, hipStream_t s);

// ---------------------------------------------------------------- iisph.hip
void launch_iisph_begin(const StepCtx& c, float gx, float gy, float gz, bool acc_has_user, hipStream_t s);  // acc += g
void launch_iisph_dii(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);            // + p = 0.5 * dv.w
void launch_iisph_pred_density(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);   // rho_star
void launch_iisph_aii(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);
void launch_iisph_dij_pj(const StepCtx& c, const TileLds& L, float dt, const float* p, hipStream_t s);
void launch_iisph_next_pressure(const StepCtx& c, const TileLds& L, float dt, float omega, const float* p, float* p_next, hipStream_t s);
void launch_iisph_velocity_changes(const StepCtx& c, const TileLds& L, float dt, const float* p, hipStream_t s);
void launch_iisph_finish(const StepCtx& c, float dt, const float* p, int32_t* bbox_partials, int32_t* bbox6, hipStream_t s);

}  // namespace salva

// The same launchers as compiled a second time with -DSALVA_OTHER_KERNELS (common.h, SALVA_KNS): what SALVA_OK_DISPATCH
// hands a world with a non-default KernelDensity / KernelGradient to.
namespace salva_ok {
using namespace salva;
void launch_density_alpha(const StepCtx& c, const TileLds& L, float iisph_dt, hipStream_t s);
void launch_divergence(const StepCtx& c, const TileLds& L, hipStream_t s);
void launch_divergence_apply(const StepCtx& c, const TileLds& L, float inv_dt_prev, hipStream_t s);
void launch_pred_density(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);
void launch_pressure_apply(const StepCtx& c, const TileLds& L, float inv_dt, hipStream_t s);
void launch_boundary_volumes(const StepCtx& c, unsigned long long* ncontacts_bb, hipStream_t s);
void launch_iisph_dii(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);
void launch_iisph_pred_density(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);
void launch_iisph_aii(const StepCtx& c, const TileLds& L, float dt, hipStream_t s);
void launch_iisph_dij_pj(const StepCtx& c, const TileLds& L, float dt, const float* p, hipStream_t s);
void launch_iisph_next_pressure(const StepCtx& c, const TileLds& L, float dt, float omega, const float* p, float* p_next, hipStream_t s);
void launch_iisph_velocity_changes(const StepCtx& c, const TileLds& L, float dt, const float* p, hipStream_t s);
void launch_xsph(const StepCtx& c, const TileLds& L, uint32_t model, float fluid_coeff, float boundary_coeff, float inv_dt_prev, hipStream_t s);
void launch_artificial_viscosity(const StepCtx& c, const TileLds& L, uint32_t model, float fluid_coeff, float boundary_coeff, float alpha,
                                 float beta, float speed_of_sound, hipStream_t s);
void launch_akinci_normals(const StepCtx& c, const TileLds& L, uint32_t model, hipStream_t s);
void launch_akinci_forces(const StepCtx& c, const TileLds& L, uint32_t model, float tension, float adhesion, hipStream_t s);
void launch_he2014_colors(const StepCtx& c, const TileLds& L, uint32_t model, float* colors, hipStream_t s);
void launch_he2014_gradc(const StepCtx& c, const TileLds& L, uint32_t model, const float* colors, float* gradcs, hipStream_t s);
void launch_he2014_forces(const StepCtx& c, const TileLds& L, uint32_t model, float tension, float boundary_tension,
                          const float* gradcs, hipStream_t s);
void launch_wcsph_tension(const StepCtx& c, const TileLds& L, uint32_t model, float tension, hipStream_t s);
void launch_visc_betas(const StepCtx& c, const TileLds& L, uint32_t model, float* beta, hipStream_t s);
void launch_visc_strain(const StepCtx& c, const TileLds& L, uint32_t model, int mode, float coef, const float4* va,
                        const float* beta, float* target, float4* u0, float4* u1, hipStream_t s);
void launch_visc_accel(const StepCtx& c, const TileLds& L, uint32_t model, float inv_dt_prev, float dt_prev, const float4* u0,
                       const float4* u1, float4* va, hipStream_t s);
}  // namespace salva_ok
