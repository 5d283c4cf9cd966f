// world.h — host orchestration of the device-resident fluid world (the body of
// /root/reference/src/liquid_world.rs:67-158 re-designed around HBM-resident, cell-sorted SoA state).
#pragma once
#include <string>
#include <map>
#include <memory>
#include <vector>

#include "../../include/salva_hip.h"
#include "common.h"
#include "device_types.h"
#include "comm.h"
#include "dist.h"
#include "kernels.h"

namespace salva {

struct FluidSlot {
    uint64_t n = 0;
    float density0 = 1000.0f;
    // the one volume every particle of this fluid has (`Fluid::new`'s default, fluid.rs:110-120, or a uniform user array), NaN when the
    // volumes differ: what lets the host know the particle masses without looking at the device (World::decide_two_mass)
    float vol_uniform = 0.0f;
    uint32_t memberships = 1u, filter = 0xffffffffu;
    std::vector<SalvaHipForceDesc> forces;
    std::vector<uint32_t> force_iters;  // iterative forces (DFSPHViscosity): iterations / last error of the last step
    std::vector<float> force_errs;
};
struct BoundarySlot {
    uint64_t n = 0;
    uint32_t memberships = 1u, filter = 0xffffffffu;
    bool wants_forces = false;
    bool vel_zero = true;  // every velocity of the last upload was exactly zero (a sampled boundary's are written on the device: treated as moving)
    // ColliderSampling::StaticSampling(points): collider-local sample points (integrations/rapier/fluids_pipeline.rs:36-41)
    std::shared_ptr<DevBuf<float4>> sampling;
    // ColliderSampling::DynamicContactSampling (:42-43): the collider's shape and its last pose; the boundary's particles are
    // re-emitted by every step (World::run_dynamic_sampling), `dyn_src` = host index of the fluid particle behind each
    // (decomposed run: its global id, and `dyn_src_model` its fluid — the particle may live on another rank)
    int dyn_kind = 0;
    SalvaHipShape dyn_shape{};
    SalvaHipHostShape dyn_host{};  // dyn_kind == SALVA_HIP_SHAPE_HOST: the host's compute_aabb / project_point callbacks
    SalvaHipRigidPose dyn_pose{};
    std::shared_ptr<DevBuf<uint32_t>> dyn_src, dyn_src_model;
};

struct GridDims {          // tile-aligned dense grid (tile.h)
    int o[3] = {0, 0, 0};  // cell coords of the first cell, multiples of the tile shape
    int nt[3] = {1, 1, 1}; // tiles per axis
    uint32_t mask[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};  // folded axes: period in cells - 1 (device_types.h TileGrid)
    bool folded() const { return (mask[0] & mask[1] & mask[2]) != 0xffffffffu; }
    TileGrid device(const uint32_t* cell_start) const { return TileGrid{o[0], o[1], o[2], nt[0], nt[1], nt[2], mask[0], mask[1], mask[2], cell_start}; }
    size_t ntiles() const { return (size_t)nt[0] * nt[1] * nt[2]; }
    size_t ncells() const { return ntiles() * TCELLS; }
};

class World {
  public:
    explicit World(const SalvaHipParams& p);
    ~World();

    void set_fluid(uint32_t slot, uint64_t n, const float* pos, const float* vel, const float* vol, const float* acc,
                   const float* dvs, float density0, uint32_t memberships, uint32_t filter, uint32_t dirty);
    void set_fluid_forces(uint32_t slot, const SalvaHipForceDesc* f, uint32_t nf);
    void remove_fluid(uint32_t slot);
    void set_boundary(uint32_t slot, uint64_t n, const float* pos, const float* vel, uint32_t memberships,
                      uint32_t filter, bool wants_forces);
    void remove_boundary(uint32_t slot);
    int step(float dt, const float g[3], SalvaHipStepStats* stats);
    void add_particles(uint32_t slot, uint64_t n_add, const float* pos, const float* vel);
    uint64_t delete_particles(uint32_t slot, const uint8_t* mask);
    uint64_t particles_in_aabb(const float mins[3], const float maxs[3], uint64_t capacity, uint32_t* kinds, uint32_t* slots,
                               uint32_t* indices);
    uint64_t particles_in_shape(const float t[3], const float q[4], const SalvaHipShape& shape, uint64_t capacity, uint32_t* kinds,
                                uint32_t* slots, uint32_t* indices);
    uint64_t particles_in_host_shape(const SalvaHipHostQueryShape& shape, uint64_t capacity, uint32_t* kinds, uint32_t* slots, uint32_t* indices);
    void map_query_hits(std::vector<uint64_t>& keys, uint32_t* kinds, uint32_t* slots, uint32_t* indices);
    void get_fluid(uint32_t slot, float* pos, float* vel);
    void get_force_stats(uint32_t slot, uint32_t force, int32_t* iters, float* err);
    uint64_t get_fluid_contacts(uint32_t slot, int boundary, uint64_t* offsets, uint32_t* j_model, uint32_t* j, uint64_t capacity);
    void get_fluid_field(uint32_t slot, int field, float* out);
    void get_boundary(uint32_t slot, float* volumes, float* forces);
    void set_boundary_sampling(uint32_t slot, uint64_t n, const float* local_points, uint32_t memberships, uint32_t filter);
    void update_boundary_pose(uint32_t slot, const SalvaHipRigidPose& pose);
    void set_boundary_dynamic_sampling(uint32_t slot, const SalvaHipShape& shape, uint32_t memberships, uint32_t filter);
    void set_boundary_dynamic_sampling_host(uint32_t slot, const SalvaHipHostShape& shape, uint32_t memberships, uint32_t filter);
    void clear_boundary_sampling(uint32_t slot);
    uint64_t local_len() const { return n; }
    void get_local(uint32_t* ids, uint32_t* fluid_slots, uint8_t* is_ghost, float* positions, float* velocities, float* densities, float* volumes);
    uint64_t get_local_contacts(int boundary, uint64_t* offsets, uint32_t* j_model, uint32_t* j, uint64_t capacity);
    void force_add_local_accelerations(const float* acc_h);
    void get_dist_timing(double out[4]) const { for (int k = 0; k < 4; ++k) out[k] = dist_times[k]; }
    void get_fluid_async(uint32_t slot, float* pos, float* vel_out);
    void wait_download();
    uint64_t boundary_len(uint32_t slot) const;
    void get_boundary_sources(uint32_t slot, uint32_t* fluid_slots, uint32_t* indices);
    void set_force_callback(SalvaHipForceCallback cb, void* user, SalvaHipWorld* owner) { force_cb = cb; force_user = user; force_owner = owner; }
    void force_get_state(uint32_t slot, float* positions, float* velocities, float* densities);
    void force_add_accelerations(uint32_t slot, const float* acc);
    bool in_force_callback() const { return in_force_cb; }
    // CouplingManager::update_boundaries / transmit_forces inside the substep loop (include/salva_hip.h, salva_hip_set_coupling_callback)
    void set_coupling_callback(SalvaHipCouplingCallback cb, void* user, SalvaHipWorld* owner) { coupling_cb = cb; coupling_user = user; coupling_owner = owner; }
    void set_fluid_field(uint32_t slot, int field, const float* data);
    void get_timestep(float* dt, float* inv_dt) const { if (dt) *dt = dt_prev; if (inv_dt) *inv_dt = inv_dt_prev; }
    void set_timestep(float dt, float inv_dt) { dt_prev = dt; inv_dt_prev = inv_dt; }
    void get_boundary_particles(uint32_t slot, float* positions, float* velocities);
    void get_boundary_wrench(uint32_t slot, const float point[3], float force[3], float torque[3]);
    void clear_boundary_forces(uint32_t slot);
    uint64_t device_bytes() const;
    // multi-GPU: this world owns the cell planes [lo, hi] along x; neighbours are rank-1 / rank+1 of `transport`
    void set_domain(Transport* transport, int lo, int hi, uint32_t gid_offset);
    // collective: re-cut the slabs at cell planes so that every rank owns about the same number of particles; the
    // particles follow in the next step's migration.  Returns this rank's new [lo, hi].
    void rebalance(int32_t* new_lo, int32_t* new_hi);
    void set_timers(bool on) { prm.enable_timers = on ? 1 : 0; }
    void set_cfl(int mode, float coeff, int min_sub, int max_sub);
    const std::vector<float>& last_substeps() const { return substeps; }
    uint64_t get_owned(uint32_t cap, uint32_t* gids, float* pos, float* vel, uint32_t* models);
    uint64_t delete_owned(uint32_t n_ids, const uint32_t* gids);
    uint32_t owned_count() const { return comm ? n_owned : n; }
    float time_pred_density(int reps);
    float time_kernel(int kernel, int reps);
#ifdef SALVA_HIP_DIAG
    // kernel experiments (diag/world_diag.hip, salva_hip_time_variant): time variant `variant` of k_pred_density; *checksum =
    // FNV-1a of the kappa it wrote
    float time_variant(int variant, uint32_t param, int reps, uint64_t* checksum);
    void tile_timing_report();
#endif

    SalvaHipCounters counters{};  // the reference's Counters tree of the last step (counters/mod.rs:17-72)
    SalvaHipParams prm;
    SphConsts sc;
    std::vector<FluidSlot> fluids;
    std::vector<BoundarySlot> bounds;

  private:
    void use_device() const;
    struct QuerySrc query_fluid_source();
    uint64_t collect_query(unsigned int* d_count, uint32_t* d_kind, uint32_t* d_index, uint32_t cap, uint32_t* kinds, uint32_t* slots,
                           uint32_t* indices);
    uint64_t fluid_offset(uint32_t slot) const;
    uint64_t boundary_offset(uint32_t slot) const;
    void ensure_staging_current();
    void upload_tables();
    void build_boundary_grid();
    void resize_boundary_slot(uint32_t slot, uint64_t nn);
    bool has_dynamic_sampling() const;
    void run_dynamic_sampling();   // between the cell keys and the sort (fluids_pipeline.rs:193-259 inside liquid_world.rs:94-103)
    DevBuf<float4> dcs_cand, dcs_out, dcs_proj, dcs_cand2;  // (_proj, _cand2: the host-shape arm)
    std::vector<float> dcs_h_pts, dcs_h_proj;
    std::vector<float4> dcs_h_f4;
    std::vector<uint8_t> dcs_h_inside;
    DevBuf<uint8_t> dcs_flag;
    DevBuf<uint32_t> dcs_num;
    // decomposed run: every rank's emitted points, in rank order (dist_gather_emitted): rows, then one uint32 fluid per row
    DevBuf<unsigned long long> dcs_all;
    uint32_t dist_gather_emitted(const float4* rows, uint32_t cnt, const float4** all_rows, const uint32_t** all_models,
                                 const HipError* local_error = nullptr);
    void ensure_cub_temp(size_t bytes);
    StepCtx make_ctx();
    struct SolveResult { uint32_t iters; float err; };
    template <typename Eval, typename Apply>
    SolveResult run_solve(StepCtx c, int which, float tol, int min_iter, int max_iter, uint32_t mode, Eval&& eval, Apply&& apply,
                          bool spec_apply = false, int chain_stage = 0, int from = 0, bool chain_open = false);
    // Chained steps (device_types.h StepCtx::gate, World::dfsph_solve): both solves of a DFSPH step and everything between and behind
    // them are enqueued without a wait; the end-of-step publication carries their outcome.
    bool chain_allowed() const;
    SolveCtl pre_init1{};          // the pressure solve's initial control block, written by the divergence solve's k_init_ctl launch
    bool pre_init1_valid = false;
    bool chain_off = false;       // SALVA_HIP_NO_CHAIN=1 (A/B, tests)
    bool chain_pending = false;   // this pass was enqueued that way and its outcome has not been read yet
    bool chain_div_pending = false;  // ... the divergence solve included (not while its iteration count is rising)
    int chain_batch[2] = {0, 0};  // iterations enqueued for the divergence / the pressure solve (where a continuation starts)
    float chain_dt_prev = 0.0f, chain_inv_dt_prev = 0.0f;  // TimestepManager::{dt, inv_dt} as the chained attempt found them
    uint64_t chain_steps = 0, chain_breaks = 0;  // passes whose chain held / broke (SALVA_HIP_TILE_TRACE prints them)
    DevBuf<float4> w2;            // the second w buffer of speculative divergence applies (dfsph.hip, spec_decide)
    DevBuf<SolveCtl> spec_ring;   // their two alternating control records
    bool spec_apply_off = false;  // SALVA_HIP_NO_SPEC_APPLY (A/B, tests)
    void wait_stream();  // low-latency wait for the world's stream (spins on an event)
    void run_forces(const StepCtx& c);
    void dfsph_solve(StepCtx& c, float& dt, const float g[3], SalvaHipStepStats& st, int resume = 0);  // dt: in = the step, out = the substep advanced by
    void iisph_solve(StepCtx& c, float& dt, const float g[3], SalvaHipStepStats& st);
    int substep(float& dt, const float g[3], SalvaHipStepStats& st);  // one pass of the reference's substep loop (liquid_world.rs:85-147)
    // Opt-in CFL sub-stepping (salva_hip_set_cfl): TimestepManager's cfl_coeff / min / max_num_substeps (timestep_manager.rs:23-34)
    // and the clamp its compute_substep left commented out (:90-93).  0 = off: one substep per step, as the reference runs.
    int cfl_mode = 0;
    float cfl_coeff = 0.4f;
    int cfl_min_sub = 1, cfl_max_sub = 10;
    float step_total = 0.0f, step_remaining = 0.0f;  // TimestepManager::{total_step_size, remaining_time} of the running step
    std::vector<float> substeps;                     // substep lengths of the last step (salva_hip_get_substeps)
    float choose_substep(const StepCtx& c);
    FluidArrays arrays(int which);
    DistArrays dist_arrays(int which);
    void dist_prepare();        // migration + ghost planes, before the grid is built
    void dist_build_lists();    // after the cell sort
    void refresh_f32(float* field);
    void refresh_f4(float4* field);
    void ensure_particle_capacity(size_t cap);
    void finalize_solve(SolveCtl* ctl, SolveCtl* pub, uint32_t skipped = 0);
    bool solve_owes_apply[3] = {false, false, false};  // the apply behind a batch's last test, left to whoever continues the solve

    hipStream_t stream = nullptr;
    // decomposed runs: evaluate passes over interior tiles run here while the ghost exchange is in flight on `stream`
    hipStream_t stream2 = nullptr;
    // Asynchronous read-back (salva_hip_get_fluid_async / _wait_download): positions / velocities are scattered into host order
    // on the main stream (into dl_dev), copied out by the copy stream behind an event, and — for pageable destinations —
    // handed over from the pinned ring in wait_download.  One download in flight at a time.
    hipStream_t dl_stream = nullptr;
    hipEvent_t ev_dl_ready = nullptr, ev_dl_done = nullptr;
    DevBuf<float> dl_dev[2];
    float* h_dl[2] = {nullptr, nullptr};
    size_t h_dl_cap[2] = {0, 0};
    struct PendingDownload { bool active = false; float* dst[2] = {nullptr, nullptr}; bool staged[2] = {false, false}; size_t bytes = 0; } dl;
    hipEvent_t ev_pre_refresh = nullptr, ev_interior = nullptr;
    // decomposed solves with speculative applies (World::run_solve): evaluate done -> the apply may start on stream2; apply done ->
    // the main stream may refresh what it wrote
    hipEvent_t ev_spec_eval = nullptr, ev_spec_apply = nullptr;
    hipStream_t spec_dist_stream = nullptr;  // non-null while run_solve launches such an apply: where its kernel goes
    bool spec_dist_off = false;              // SALVA_HIP_NO_SPEC_DIST=1 (A/B, tests)
    bool overlap_exchange = true;   // SALVA_HIP_NO_OVERLAP=1 turns it off (diagnostics)
    template <typename Launch> void evaluate_split(const StepCtx& c, int iteration, Launch&& launch);
    uint32_t n = 0, nb = 0;

    // canonical (host order) staging: st_pos = (x,y,z,volume), st_vel = (v,0), st_dv = (dv, pressure), st_acc
    DevBuf<float4> st_pos, st_vel, st_dv, st_acc;
    DevBuf<uint32_t> st_model;
    bool staging_current = true, sorted_valid = false, acc_user = false;

    // cell-sorted working set
    DevBuf<float4> posm[2], vel[2], dv[2];
    DevBuf<uint32_t> model[2], perm[2];
    int cur = 0;
    static constexpr int NUM_SOLVES = 3;  // divergence, pressure, viscosity
    DevBuf<float4> acc, w, normal, dii, dijpj, iisph_q, iisph_pr, posmr;
    DevBuf<float> visc_beta, visc_target;  // DFSPHViscosity scratch: betas [36][n], strain-rate targets [6][n]
    DevBuf<float4> visc_u0, visc_u1, visc_va;
    DevBuf<double> wrench_partial;       // per-block partial sums of salva_hip_get_boundary_wrench
    DevBuf<float> he_colors, he_gradcs;  // He2014SurfaceTension state (he2014_surface_tension.rs:15-16)
    DevBuf<float> rho, alpha, kappa, kappa2, rho_star, aii;
    DevBuf<uint32_t> nff, nfb;
    // What the first part of a step builds from the positions alone — keys, the sort's permutation, the cell table, the non-empty
    // tiles and their halo / slice counts — exists TWICE: the end of a step may enqueue this part of the NEXT step already
    // (World::pre_enqueue_grid, round 6) while the tables of the step that has just run still serve contact exports and queries.
    struct GridTabs {
        DevBuf<uint32_t> keys[2], idx[2], cell_start_f, cell_rank, tile_ids, tile_flags, tile_rank;
        DevBuf<TileAcc> tile_cnt, tile_off;
        DevBuf<uint4> slot_desc;
    };
    GridTabs gtab[2];
    int gsel = 0;  // the set the current step works on
    GridTabs& G() { return gtab[gsel]; }
    DevBuf<uint4> slot_info;
    DevBuf<uint32_t> d_maxhalo, halo_src, bhalo_src;
    uint32_t nlaunch = 0;  // non-empty tiles of the current step = grid size of the solver kernels
    int sched_mode = 0;  // kernel-development builds: 1 = run diag/sched.hip after the list build (SALVA_HIP_SCHED)
    uint32_t last_iters[NUM_SOLVES] = {1u, 1u, 1u};  // iterations of the previous step's divergence / pressure solve (batch sizing)
    uint32_t prev_iters[NUM_SOLVES] = {1u, 1u, 1u};  // ... and of the step before it (a rising count widens a chained batch)
    uint32_t halo_stride = 0, bhalo_stride = 0;  // fixed row stride of the slot tables (0 = compact)
    DevBuf<char> tile_list_stats;
    uint32_t cap_ff = 24, cap_fb = 8;  // ELL capacity (dwords per particle), grown on demand
    DevBuf<uint32_t> nbr_ff, nbr_fb;
    DevBuf<uint32_t> slice_near;   // per slice: a pair closer than 1e-5 h exists (written by k_density_alpha, read by the DFSPH solver kernels)
    DevBuf<int32_t> bbox_partials;
    TileLds lds;
    float mass_uniform = 0.0f;  // StepCtx::mass_uniform of the current step (0: masses differ, or not known)
    // Two-mass worlds (device_types.h StepCtx::two_mass; BASELINE config 4): every fluid has one particle mass (uniform volumes:
    // FluidSlot::vol_uniform) and exactly two different masses occur.  The plane-layout kernels then serve the whole world in one
    // launch per pass, the heavier class as a tail segment of the lists in the tiles that hold both.
    // Round 6: up to four masses — the third and fourth class as further tail segments (tile_masscd_bits, nffc).
    DevBuf<uint32_t> tile_mass_bits, tile_massb_bits, nffb, nffc;
    DevBuf<uint2> tile_masscd_bits;
    uint32_t max_masses = 2;      // SALVA_HIP_MAX_MASSES=3 / 4: opt-in — on the one 10^6-particle scene it was measured on (four columns,
                                  // tools/r06/multi_mass_probe.py) the general kernels are 8-10 % faster than the segments of three and four masses
    bool two_mass_off = false;    // SALVA_HIP_NO_TWO_MASS=1 (A/B, tests): such a world keeps the general kernels
    bool fold_off = false;        // SALVA_HIP_NO_FOLD=1: the fluid grid is never folded (device_types.h TileGrid)
    struct FoldRetry {};          // thrown by substep when the tile totals show a fold that piled the bulk onto itself (World::step retries)
    uint32_t fold_relax = 0;      // how often that happened: the fold rule is loosened eightfold per level, given up at 3
    bool fold_locked = false;     // the looser fold did not fit the cell-table budget: keep the tighter one
    uint32_t fold_forced = 0;     // SALVA_HIP_FOLD_CELLS=P: every axis longer than P cells is folded to exactly P (tests)
    bool two_mass = false;        // this step runs that way
    uint32_t nmass = 0;           // ... with this many masses,
    float mass_classes[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // these, ascending,
    uint64_t mass_cmask = 0;      // and this class per fluid (two bits each)
    bool decide_two_mass();
    // Decomposed runs, timers enabled (salva_hip_enable_counters): HIP event pairs around every ghost refresh (gather -> exchange ->
    // scatter) and every all-reduced convergence test (sum -> all-reduce -> decide) of a step, folded into `dist_times` at its end:
    // {refresh ms, refreshes, test ms, tests} — what an exchange costs INSIDE a decomposed step, waiting for the neighbour included
    std::vector<hipEvent_t> dist_ev;
    size_t dist_ev_used = 0;
    std::vector<std::pair<size_t, int>> dist_ev_pairs;  // (index of the first event of a pair, 0 = refresh / 1 = test)
    double dist_times[4] = {0.0, 0.0, 0.0, 0.0};
    size_t dist_time_begin(int kind);
    void dist_time_end(size_t first);
    void dist_time_fold();
    bool fused_first_divergence = false;  // this step's density pass also ran the divergence solve's first evaluate (dfsph.hip)
    bool flags_clean = false;   // d_flags were cleared by the last end-of-step publication and nothing has run since
    bool check_mass = true;     // this step's k_cell_keys compares the masses (not while a scene is known to hold different ones)
    bool mass_known = false;    // mass_uniform describes the particles as they are (set by a publication, cleared by host edits)
    bool tile_trace = false;    // SALVA_HIP_TILE_TRACE=1: one line of tile statistics per step on stderr
    bool no_fused_div = false;  // SALVA_HIP_NO_FUSED_DIV=1 (A/B): the first divergence evaluate stays a pass of its own
    bool no_planes = false;     // SALVA_HIP_NO_PLANES=1 (A/B): keep the 32-byte-per-slot evaluate kernels
    bool iisph_dii_fused = false;  // this step's density pass wrote d_ii (k_density_alpha<true>): iisph_solve skips k_iisph_dii
    int sort_mode = -1;         // SALVA_HIP_RADIX_SORT: 1 = always the radix sort + k_cell_start, 0 = always the counting sort by cell, unset = by size
#ifdef SALVA_HIP_DIAG
    PipeCfg pipe;          // launch shape of the persistent pipeline kernels of this step (pipe.h)
#endif
    // speculative sizing (World::step): the previous step's table totals, and what the current pass lets the kernels use
    TileAcc pred_tt{};
    uint32_t pred_n = 0;
    bool pred_valid = false, spec_mode = false;
    uint32_t halo_cap = 0xffffffffu, bhalo_cap = 0xffffffffu, nslices_cap = 0xffffffffu;
    uint64_t halo_len = ~0ull, bhalo_len = ~0ull;
    uint64_t spec_misses = 0;  // passes discarded because the prediction did not hold
    bool trust_cap0 = false;     // SALVA_HIP_LIST_CAP0 (tests)
    bool lists_checked = false;  // the list capacity has held once since the last edit of the objects (upload_tables clears it)
    bool defer_off = false;  // SALVA_HIP_NO_DEFER_LISTS: check the list capacity in the middle of the step (read at construction)
    bool spec_off = true, spec_tight = false;  // SALVA_HIP_SPECULATE / SALVA_HIP_SPEC_TIGHT, read at construction
    int num_cus = 256;
    DevBuf<char> cub_temp;
    DevBuf<float> scratch_f;   // staging for AoS up/downloads and field unsorts
    DevBuf<float4> scratch_f4;

    // boundaries: canonical + sorted
    DevBuf<float4> bst_pos, bst_vel;
    DevBuf<float4> bposv, bvel, bforce;
    DevBuf<uint32_t> bperm, cell_start_b, bkeys[2], bidx[2];
    bool b_dirty = true;
    uint64_t ncontacts_bb = 0;

    GridDims gf, gb;
    bool bbox_known = false;

    // per-model tables
    DevBuf<float> rho0_tab;
    DevBuf<uint8_t> ff_ok, fb_ok, bb_ok, bwants;
    DevBuf<uint32_t> model_counts;
    bool tables_dirty = true, any_wants_forces = false;

    // reductions / readback
    DevBuf<float> partials;
    DevBuf<Readback> d_rb;
    DevBuf<uint32_t> mass_slots;  // k_cell_keys' {min, max} pairs of the mass bits (grid.hip), folded and reset by k_publish_readback
    struct FlagsPtr { uint32_t* p = nullptr; } d_flags;  // = &d_rb.p->flags: flags and the next step's box come back in one copy
    DevBuf<unsigned long long> d_counters;
    Readback* h_rb = nullptr;
    // host-mapped copy of the words the host waits for twice per step (tile totals; flags + next box + list statistics): written by
    // a one-wave kernel at the point of the stream where a hipMemcpyAsync + event used to sit, polled by the host (publish_wait)
    struct HostPub { uint32_t seq, pad[3]; Readback rb; };
    HostPub* h_hostpub = nullptr;
    uint32_t hostpub_seq = 0;
    uint32_t publish_enqueue(const TileAcc* totals, bool lists, bool end_of_step, const PrePub* pre = nullptr, const uint32_t* gate = nullptr);
    // The grid part of the next step, enqueued at the end of this one (round 6).  A step starts with ~14 launches of kernels that
    // take a few microseconds each, on an empty queue, right after the host has returned from one step and entered the next: the
    // GPU waits for the host all the way (tools/r06/slow_host.c: 10-15 launches of a free-fall step sit on its critical path).
    // When the particles' cell box did not change over the last step, the end of a step therefore enqueues keys -> counting sort ->
    // non-empty tiles -> per-tile counts -> totals publication for the SAME grid, into the other set of tables (gtab[gsel ^ 1]),
    // gated on the device by Readback::pre_ok = "the box the position update found is that box".  The next step adopts the work
    // when nothing has touched the world in between, and runs its own otherwise.
    struct PreGrid {
        bool valid = false;
        uint32_t n = 0, seq = 0, nslots_bound = 0, ntiles = 0;
        size_t ncf = 0;
        bool check_mass = false;
        uint32_t split_s = 0;
        GridDims gf;
    } pre;
    bool pre_off = false;          // SALVA_HIP_NO_PREGRID=1 (A/B, tests)
    // two launch classes per pass (device_types.h StepCtx::slot_order)
    DevBuf<uint32_t> slot_order;
    // scratch of particles_in_host_shape: candidate kinds / indices / positions, kept between queries
    DevBuf<unsigned int> hq_cnt;
    DevBuf<uint32_t> hq_kind, hq_index;
    DevBuf<float4> hq_pos;
    uint32_t hq_need = 0;
    uint32_t class_ntiny = 0;      // sparse slots of this step, when they run in launches of their own (0: with the others)
    uint32_t class_nlight = 0;     // light slots of this step, when they run in launches of their own (0: with the full ones)
    bool classes_off = false, classes_forced = false, light_on = false;  // SALVA_HIP_NO_CLASSES=1 / SALVA_HIP_CLASSES=1 / SALVA_HIP_LIGHT=1 (the light class: opt-in, it lost)
    // splitting of over-full tiles (device_types.h StepCtx::split_s)
    bool split_off = false;        // SALVA_HIP_NO_SPLIT=1
    uint32_t split_forced = 0;     // SALVA_HIP_SPLIT_S=k: split at k halo particles whatever the statistics say (tests)
    bool split_on = false;         // the previous step's totals say: a few tiles are over-full
    uint32_t split_s_cur = 0;      // what this step's tables are built with
    int32_t bbox_used_last[6] = {0, 0, 0, 0, 0, 0};  // the cell box the previous step ran on
    bool bbox_used_valid = false;
    uint64_t pre_adopted = 0, pre_dropped = 0;
    void pre_enqueue_grid(uint32_t nslots_bound);
    void pre_drop();
    void publish_wait(uint32_t seq, bool totals, bool lists, bool end_of_step);
    // what the next end-of-step publication folds (world.hip Epilogue): the list statistics of this pass, the position update's boxes
    bool fold_stats = false;
    uint32_t fold_bbox_blocks = 0;
    const uint32_t* fold_bbox_gate = nullptr;
    void publish_and_wait(const TileAcc* totals, bool lists, bool end_of_step);
    DevBuf<SolveCtl> d_ctl;      // [0] divergence solve, [1] pressure solve, [2] viscosity solve (DFSPHViscosity)
    SolveCtl* h_ctl = nullptr;   // pinned: [0..NUM_SOLVES) read-back, [NUM_SOLVES..2 NUM_SOLVES) initial values
    SolveCtl* h_pub = nullptr;   // host memory mapped into the device: k_finalize_error publishes every test's outcome here

    // The reference's solver buffers (velocity_changes, IISPH pressures) are positional per fluid SLOT and outlive the object:
    // remove_fluid swap-removes the fluid only, and the buffers are resized / truncated by the next init_with_fluids
    // (dfsph_solver.rs:526-549, iisph_solver.rs:479-501).  Until the next step, `sticky[slot]` holds the content of such a
    // buffer where it differs from what the fluid occupying the slot carries: a fluid moved or created into the slot, and
    // particles added to it, inherit from it exactly as the reference's `resize` would hand it to them.
    struct StickyBuf { std::shared_ptr<DevBuf<float4>> data; uint64_t len = 0; };
    std::map<uint32_t, StickyBuf> sticky;
    SalvaHipForceCallback force_cb = nullptr;
    void* force_user = nullptr;
    SalvaHipWorld* force_owner = nullptr;
    bool in_force_cb = false;
    SalvaHipCouplingCallback coupling_cb = nullptr;
    void* coupling_user = nullptr;
    SalvaHipWorld* coupling_owner = nullptr;
    void call_coupling(int phase, float dt);
    float dt_prev = 0.0f, inv_dt_prev = 0.0f;  // TimestepManager::{dt, inv_dt} persist across steps (timestep_manager.rs:23-34)
    StepCtx last_ctx{};
    float last_dt = 0.0f;
    bool have_last_ctx = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t evc[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // Counters: fluids in grid, boundaries in grid, densities done, custom on / off
    uint64_t spec_passes = 0;
    hipEvent_t ev_sync = nullptr;

    // ---- multi-GPU slab decomposition (comm.h / dist.h)
    Transport* comm = nullptr;  // not owned
    int slab_lo = 0, slab_hi = 0;
    int nbr_lo_lo = INT32_MIN, nbr_hi_hi = INT32_MAX;  // far ends of the neighbours' slabs (open ends: MIN / MAX)
    bool nbr_bounds_valid = false;
    DevBuf<unsigned long long> plane_hist;
    std::vector<uint32_t> global_counts;  // particles per fluid over all ranks (the denominators of the error averages)
    uint32_t gid_offset = 0, n_owned = 0;
    uint64_t gid_next = 0;  // first global id not in use (dist_add_particles)
    void dist_add_particles(uint32_t slot, uint64_t n_add, const float* pos, const float* vel_h);
    void dist_update_counts(const std::vector<long long>& delta);
    bool dist_started = false;
    DevBuf<uint32_t> gtag[2];
    DevBuf<DistRec> xsend_lo, xsend_hi, xrecv_lo, xrecv_hi;
    DevBuf<char> dsel, dpos;
    DevBuf<uint32_t> send_lo_idx, send_hi_idx, ghost_lo_idx, ghost_hi_idx;
    uint32_t nborder_lo = 0, nborder_hi = 0, nghost_lo = 0, nghost_hi = 0;
    DevBuf<float4> fbuf_send, fbuf_recv;
    DevBuf<float> d_sums;
};

}  // namespace salva
