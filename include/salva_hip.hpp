// salva_hip.hpp — header-only C++ mirror of the salva3d host API for the `LiquidWorld::step` path, on top of the C ABI
// (include/salva_hip.h).  The reference is Rust (no cargo/rustc in this image), so the host side above the ABI is
// written in C++ with the reference's names, argument meaning and error behaviour:
//
//   salva::LiquidWorld              /root/reference/src/liquid_world.rs:17-209
//   salva::Fluid / Boundary         src/object/fluid.rs:12-185, src/object/boundary.rs:11-84
//   salva::InteractionGroups        src/object/interaction_groups.rs:6-79
//   salva::DFSPHSolver/IISPHSolver  src/solver/pressure/dfsph_solver.rs:21-70, iisph_solver.rs:21-64 (pub tuning fields)
//   salva::XSPHViscosity / ArtificialViscosity / Akinci2013SurfaceTension   src/solver/{viscosity,surface_tension}/*.rs
//
// `Fluid`'s pub fields stay plain std::vector members the caller may edit between steps (faucet3.rs:69-104,
// heightfield3.rs:40); `LiquidWorld::step` uploads what changed (call `mark_dirty()` after in-place edits of
// positions / velocities / volumes) and refreshes positions / velocities after the step, like the reference whose host
// arrays are always current.  Panics of the reference (assert!/unwrap) become salva::Error exceptions.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <memory>
#include <exception>
#include <stdexcept>
#include <string>
#include <functional>
#include <utility>
#include <vector>

#include "salva_hip.h"

namespace salva {

using Real = float;
using Vec3 = std::array<Real, 3>;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != SALVA_HIP_OK) throw Error(rc, salva_hip_last_error());
}

struct InteractionGroups {  // interaction_groups.rs:64-79
    uint32_t memberships = 1u, filter = 0xffffffffu;
    bool test(const InteractionGroups& rhs) const { return (memberships & rhs.filter) != 0 && (rhs.memberships & filter) != 0; }
};

// ---- solver::NonPressureForce built-ins (nonpressure_force.rs:10-30)
struct NonPressureForce {
    virtual ~NonPressureForce() = default;
    virtual SalvaHipForceDesc desc() const = 0;  // arbitrary user forces need host contact lists: not on the device path
};
struct XSPHViscosity : NonPressureForce {
    Real fluid_viscosity_coefficient, boundary_viscosity_coefficient;
    XSPHViscosity(Real f, Real b) : fluid_viscosity_coefficient(f), boundary_viscosity_coefficient(b) {}
    SalvaHipForceDesc desc() const override {
        SalvaHipForceDesc d{SALVA_HIP_FORCE_XSPH, {fluid_viscosity_coefficient, boundary_viscosity_coefficient}};
        return d;
    }
};
struct ArtificialViscosity : NonPressureForce {
    Real alpha = 1.0f, beta = 0.0f, speed_of_sound = 10.0f;  // artificial_viscosity.rs:29-37
    Real fluid_viscosity_coefficient, boundary_viscosity_coefficient;
    ArtificialViscosity(Real f, Real b) : fluid_viscosity_coefficient(f), boundary_viscosity_coefficient(b) {}
    SalvaHipForceDesc desc() const override {
        SalvaHipForceDesc d{SALVA_HIP_FORCE_ARTIFICIAL,
                            {fluid_viscosity_coefficient, boundary_viscosity_coefficient, alpha, beta, speed_of_sound}};
        return d;
    }
};
struct Akinci2013SurfaceTension : NonPressureForce {
    Real fluid_tension_coefficient, boundary_adhesion_coefficient;
    Akinci2013SurfaceTension(Real t, Real a) : fluid_tension_coefficient(t), boundary_adhesion_coefficient(a) {}
    SalvaHipForceDesc desc() const override {
        SalvaHipForceDesc d{SALVA_HIP_FORCE_AKINCI2013, {fluid_tension_coefficient, boundary_adhesion_coefficient}};
        return d;
    }
};
struct He2014SurfaceTension : NonPressureForce {  // surface_tension/he2014_surface_tension.rs:12-29
    Real fluid_tension_coefficient, boundary_tension_coefficient;
    He2014SurfaceTension(Real t, Real b) : fluid_tension_coefficient(t), boundary_tension_coefficient(b) {}
    SalvaHipForceDesc desc() const override {
        SalvaHipForceDesc d{SALVA_HIP_FORCE_HE2014, {fluid_tension_coefficient, boundary_tension_coefficient}};
        return d;
    }
};
struct WCSPHSurfaceTension : NonPressureForce {  // surface_tension/wcsph_surface_tension.rs:15-28
    Real fluid_tension_coefficient, boundary_tension_coefficient;
    WCSPHSurfaceTension(Real t, Real b) : fluid_tension_coefficient(t), boundary_tension_coefficient(b) {}
    SalvaHipForceDesc desc() const override {
        SalvaHipForceDesc d{SALVA_HIP_FORCE_WCSPH_TENSION, {fluid_tension_coefficient, boundary_tension_coefficient}};
        return d;
    }
};

struct DFSPHViscosity : NonPressureForce {  // viscosity/dfsph_viscosity.rs:85-125
    int min_viscosity_iter = 1, max_viscosity_iter = 50;
    Real max_viscosity_error = 0.01f;
    Real viscosity_coefficient;
    explicit DFSPHViscosity(Real coefficient) : viscosity_coefficient(coefficient) {
        if (!(coefficient >= 0.0f && coefficient <= 1.0f))
            throw std::invalid_argument("The viscosity coefficient must be between 0.0 and 1.0.");  // assert! :104-108
    }
    SalvaHipForceDesc desc() const override {
        SalvaHipForceDesc d{SALVA_HIP_FORCE_DFSPH_VISCOSITY,
                            {viscosity_coefficient, (Real)min_viscosity_iter, (Real)max_viscosity_iter, max_viscosity_error}};
        return d;
    }
};

// ---- pressure solvers: only the pub tuning fields exist on the host, the passes run on the device
struct PressureSolver {
    int kind = SALVA_HIP_SOLVER_DFSPH;
    int min_pressure_iter = 1, max_pressure_iter = 50;
    Real max_density_error = 0.05f;
    int min_divergence_iter = 1, max_divergence_iter = 50;
    Real max_divergence_error = 0.1f;
    int kernel_density = SALVA_HIP_KERNEL_CUBIC_SPLINE, kernel_gradient = SALVA_HIP_KERNEL_CUBIC_SPLINE;
};
// src/kernel/*.rs as tags: the KernelDensity / KernelGradient type parameters of the solvers (dfsph_solver.rs:17-20)
struct CubicSplineKernel { static constexpr int kind = SALVA_HIP_KERNEL_CUBIC_SPLINE; };
struct Poly6Kernel { static constexpr int kind = SALVA_HIP_KERNEL_POLY6; };
struct SpikyKernel { static constexpr int kind = SALVA_HIP_KERNEL_SPIKY; };
struct ViscosityKernel { static constexpr int kind = SALVA_HIP_KERNEL_VISCOSITY; };
template <class KernelDensity = CubicSplineKernel, class KernelGradient = CubicSplineKernel>
struct DFSPHSolverT : PressureSolver {
    DFSPHSolverT() { kind = SALVA_HIP_SOLVER_DFSPH; kernel_density = KernelDensity::kind; kernel_gradient = KernelGradient::kind; }
};
template <class KernelDensity = CubicSplineKernel, class KernelGradient = CubicSplineKernel>
struct IISPHSolverT : PressureSolver {
    IISPHSolverT() { kind = SALVA_HIP_SOLVER_IISPH; kernel_density = KernelDensity::kind; kernel_gradient = KernelGradient::kind; }
};
using DFSPHSolver = DFSPHSolverT<>;
using IISPHSolver = IISPHSolverT<>;

class LiquidWorld;

class Fluid {  // object/fluid.rs
  public:
    std::vector<std::shared_ptr<NonPressureForce>> nonpressure_forces;
    std::vector<Vec3> positions, velocities, accelerations;
    std::vector<Real> volumes;
    Real density0;
    InteractionGroups interaction_groups;

    Fluid(std::vector<Vec3> particle_positions, Real particle_radius, Real density0_, InteractionGroups groups = {})
        : positions(std::move(particle_positions)), density0(density0_), interaction_groups(groups),
          particle_radius_(particle_radius) {
        const size_t n = positions.size();
        velocities.assign(n, Vec3{0, 0, 0});
        accelerations.assign(n, Vec3{0, 0, 0});
        volumes.assign(n, default_particle_volume());
        deleted_.assign(n, false);
    }
    Real particle_radius() const { return particle_radius_; }
    Real default_particle_volume() const { return particle_radius_ * particle_radius_ * particle_radius_ * 6.4f; }  // fluid.rs:110-120
    size_t num_particles() const { return positions.size(); }
    Real particle_mass(size_t i) const { return volumes[i] * density0; }
    void delete_particle_at_next_timestep(size_t i) { deleted_[i] = true; }
    void add_particles(const std::vector<Vec3>& pos, const std::vector<Vec3>* vel = nullptr) {  // fluid.rs:126-150
        if (vel && vel->size() != pos.size()) throw Error(SALVA_HIP_E_INVALID, "The provided positions and velocities arrays must have the same length.");
        positions.insert(positions.end(), pos.begin(), pos.end());
        if (vel) velocities.insert(velocities.end(), vel->begin(), vel->end());
        velocities.resize(positions.size(), Vec3{0, 0, 0});
        accelerations.resize(positions.size(), Vec3{0, 0, 0});
        volumes.resize(positions.size(), default_particle_volume());
        if (appended_from_ == kNone) appended_from_ = deleted_.size();  // [appended_from_, size) is not on the device yet
        deleted_.resize(positions.size(), false);
    }
    void transform_by(const Vec3& translation) {
        for (auto& p : positions) for (int a = 0; a < 3; ++a) p[a] += translation[a];
        dirty_ |= SALVA_HIP_DIRTY_POSITIONS;
    }
    void mark_dirty(uint32_t mask = SALVA_HIP_DIRTY_POSITIONS | SALVA_HIP_DIRTY_VELOCITIES | SALVA_HIP_DIRTY_VOLUMES) { dirty_ |= mask; }

  private:
    friend class LiquidWorld;
    Real particle_radius_;
    std::vector<bool> deleted_;
    uint32_t dirty_ = SALVA_HIP_DIRTY_ALL;
    static constexpr size_t kNone = (size_t)-1;
    size_t appended_from_ = kNone;  // add_particles since the last upload
    bool structural_ = true;        // the whole fluid must be (re)uploaded: new fluid, or edits that the device cannot replay
};

class Boundary {  // object/boundary.rs
  public:
    std::vector<Vec3> positions, velocities;
    std::vector<Real> volumes;          // V_b, refreshed by LiquidWorld::sync_boundary()
    std::vector<Vec3> forces;           // filled when wants_forces (boundary.forces = Some(..))
    bool wants_forces = false;
    InteractionGroups interaction_groups;
    explicit Boundary(std::vector<Vec3> particle_positions, InteractionGroups groups = {})
        : positions(std::move(particle_positions)), interaction_groups(groups) {
        velocities.assign(positions.size(), Vec3{0, 0, 0});
        volumes.assign(positions.size(), 0.0f);
    }
    // ColliderSampling::StaticSampling(points) (integrations/rapier/fluids_pipeline.rs:36-41): when set, positions and
    // velocities are produced on the device from a pose (ColliderCouplingSet below) and `positions` above is only a
    // read-back (LiquidWorld::sync_boundary)
    std::vector<Vec3> sampling;
    // ColliderSampling::DynamicContactSampling (:42-43) for a ball / cuboid / capsule / cylinder collider: kind != 0 makes every step re-emit the
    // boundary's particles on the device from the fluid near the collider (salva_hip_set_boundary_dynamic_sampling);
    // positions / velocities are read back by LiquidWorld::sync_boundary
    SalvaHipShape dynamic_shape{0, {0, 0, 0}};
    static Boundary dynamic_ball(Real radius, InteractionGroups groups = {}) {
        Boundary b({}, groups);
        b.dynamic_shape = SalvaHipShape{SALVA_HIP_SHAPE_BALL, {radius, 0, 0}};
        return b;
    }
    static Boundary dynamic_cuboid(const Vec3& half_extents, InteractionGroups groups = {}) {
        Boundary b({}, groups);
        b.dynamic_shape = SalvaHipShape{SALVA_HIP_SHAPE_CUBOID, {half_extents[0], half_extents[1], half_extents[2]}};
        return b;
    }
    // parry Capsule::new_y(half_height, radius) / Cylinder::new(half_height, radius): axis = the collider's local y
    static Boundary dynamic_capsule(Real half_height, Real radius, InteractionGroups groups = {}) {
        Boundary b({}, groups);
        b.dynamic_shape = SalvaHipShape{SALVA_HIP_SHAPE_CAPSULE, {half_height, radius, 0}};
        return b;
    }
    static Boundary dynamic_cylinder(Real half_height, Real radius, InteractionGroups groups = {}) {
        Boundary b({}, groups);
        b.dynamic_shape = SalvaHipShape{SALVA_HIP_SHAPE_CYLINDER, {half_height, radius, 0}};
        return b;
    }
    // ... and for every other collider shape: the loop runs on the device, `aabb` / `project` (the two parry calls of the loop,
    // see SalvaHipHostShape in salva_hip.h) are called back on the host once per step.  The callbacks and `user` must outlive
    // the boundary's registration.
    SalvaHipHostShape dynamic_host{nullptr, nullptr, nullptr};
    static Boundary dynamic_host_shape(const SalvaHipHostShape& shape, InteractionGroups groups = {}) {
        Boundary b({}, groups);
        b.dynamic_shape = SalvaHipShape{SALVA_HIP_SHAPE_HOST, {0, 0, 0}};
        b.dynamic_host = shape;
        return b;
    }
    size_t num_particles() const { return dynamic_shape.kind ? dynamic_n_ : (sampling.empty() ? positions.size() : sampling.size()); }
    void mark_dirty() { dirty_ = true; }

  private:
    friend class LiquidWorld;
    bool dirty_ = true;
    size_t dynamic_n_ = 0;  // what the last step emitted
};

class LiquidWorld;
// coupling/coupling_manager.rs:8-31
struct CouplingManager {
    virtual ~CouplingManager() = default;
    virtual void update_boundaries(LiquidWorld& world) = 0;
    virtual void transmit_forces(LiquidWorld& world, Real dt) = 0;
};

// ---- multi-GPU: a rank's handle on the slab exchange transport (salva_hip_comm_*; no counterpart in the reference, which is
// single-process).  One process and one LiquidWorld per GPU; see LiquidWorld::set_domain.
class Comm {
  public:
    Comm() = default;
    Comm(Comm&& o) noexcept : h_(o.h_), rank_(o.rank_), size_(o.size_) { o.h_ = nullptr; }
    Comm& operator=(Comm&& o) noexcept {
        if (this != &o) { destroy(); h_ = o.h_; rank_ = o.rank_; size_ = o.size_; o.h_ = nullptr; }
        return *this;
    }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    ~Comm() { destroy(); }
    void destroy() {  // after the worlds that use it
        if (h_) salva_hip_comm_destroy(h_);
        h_ = nullptr;
    }
    // RCCL over xGMI (the default): rank 0 creates the id and distributes it (MPI_Bcast, a file ...)
    static std::array<unsigned char, 128> rccl_unique_id() {
        std::array<unsigned char, 128> id{};
        check(salva_hip_comm_rccl_unique_id(id.data()));
        return id;
    }
    static Comm rccl(int rank, int size, const std::array<unsigned char, 128>& id, int device) {
        Comm c;
        check(salva_hip_comm_rccl_create(rank, size, id.data(), device, &c.h_));
        c.rank_ = rank; c.size_ = size;
        return c;
    }
    // xGMI peer-direct, the ranks of one node: `all_gather(mine) -> every rank's 64-byte handle in rank order` is the caller's
    // all-gather (MPI_Allgather ...); it doubles as the barrier that makes every window exist before anybody writes to it
    using PeerHandle = std::array<unsigned char, SALVA_HIP_PEER_HANDLE_BYTES>;
    template <class AllGather>
    static Comm peer(int rank, int size, int device, AllGather&& all_gather, uint64_t slot_bytes = 64ull << 20) {
        PeerHandle mine{};
        SalvaHipPeerSetup* setup = nullptr;
        check(salva_hip_comm_peer_begin(rank, size, device, slot_bytes, mine.data(), &setup));
        std::vector<PeerHandle> all;
        try {
            all = all_gather(mine);
            if ((int)all.size() != size) throw Error(SALVA_HIP_E_INVALID, "peer transport: one handle per rank, in rank order");
        } catch (...) {
            salva_hip_comm_peer_abort(setup);
            throw;
        }
        Comm c;
        check(salva_hip_comm_peer_connect(setup, all[0].data(), &c.h_));  // consumes `setup`, also when it fails
        c.rank_ = rank; c.size_ = size;
        return c;
    }
    // in-process loopback for tests: drive each rank from its own host thread
    static std::vector<Comm> loopback(int size) {
        std::vector<SalvaHipComm*> hs((size_t)size, nullptr);
        check(salva_hip_comm_loopback_create(size, hs.data()));
        std::vector<Comm> out((size_t)size);
        for (int r = 0; r < size; ++r) { out[r].h_ = hs[r]; out[r].rank_ = r; out[r].size_ = size; }
        return out;
    }
    // collective checks of the transport itself: patterned exchanges + count exchange + both all-reduces; (us per exchange of
    // `bytes` each way with both neighbours, us per four-float all-reduce)
    void selftest(uint64_t max_bytes = 1u << 20, int rounds = 6) { check(salva_hip_comm_selftest(h_, max_bytes, rounds)); }
    std::pair<float, float> time(uint64_t bytes = 64u << 10, int iters = 200) {
        float a = 0, b = 0;
        check(salva_hip_comm_time(h_, bytes, iters, &a, &b));
        return {a, b};
    }
    int rank() const { return rank_; }
    int size() const { return size_; }
    SalvaHipComm* handle() const { return h_; }

  private:
    SalvaHipComm* h_ = nullptr;
    int rank_ = 0, size_ = 1;
};
static_assert(sizeof(Comm::PeerHandle) == SALVA_HIP_PEER_HANDLE_BYTES, "handles are gathered as one contiguous block");

using FluidHandle = size_t;     // dense index; remove_fluid is a swap-remove like ContiguousArena (contiguous_arena.rs)
using BoundaryHandle = size_t;

class LiquidWorld {  // liquid_world.rs
  public:
    LiquidWorld(const PressureSolver& solver, Real particle_radius, Real smoothing_factor, int device = 0)
        : particle_radius_(particle_radius) {
        SalvaHipParams p;
        salva_hip_default_params(&p);
        p.particle_radius = particle_radius;
        p.smoothing_factor = smoothing_factor;
        p.solver = solver.kind;
        p.kernel_density = solver.kernel_density; p.kernel_gradient = solver.kernel_gradient;
        p.min_pressure_iter = solver.min_pressure_iter; p.max_pressure_iter = solver.max_pressure_iter;
        p.max_density_error = solver.max_density_error;
        p.min_divergence_iter = solver.min_divergence_iter; p.max_divergence_iter = solver.max_divergence_iter;
        p.max_divergence_error = solver.max_divergence_error;
        p.device = device;
        check(salva_hip_create(&p, &w_));
    }
    ~LiquidWorld() { salva_hip_destroy(w_); }
    LiquidWorld(const LiquidWorld&) = delete;
    LiquidWorld& operator=(const LiquidWorld&) = delete;

    FluidHandle add_fluid(Fluid f) { fluids_.push_back(std::move(f)); fluids_.back().structural_ = true; return fluids_.size() - 1; }
    BoundaryHandle add_boundary(Boundary b) { boundaries_.push_back(std::move(b)); boundaries_.back().dirty_ = true; return boundaries_.size() - 1; }
    void remove_fluid(FluidHandle h) {
        upload_new_objects();
        if (h < salva_hip_num_fluids(w_)) check(salva_hip_remove_fluid(w_, (uint32_t)h));
        fluids_[h] = std::move(fluids_.back());
        fluids_.pop_back();
    }
    void remove_boundary(BoundaryHandle h) {
        upload_new_objects();
        if (h < salva_hip_num_boundaries(w_)) check(salva_hip_remove_boundary(w_, (uint32_t)h));
        boundaries_[h] = std::move(boundaries_.back());
        boundaries_.pop_back();
    }
    std::vector<Fluid>& fluids() { return fluids_; }
    std::vector<Boundary>& boundaries() { return boundaries_; }
    Real h() const { return salva_hip_h(w_); }
    Real particle_radius() const { return particle_radius_; }
    const SalvaHipStepStats& counters() const { return stats_; }
    // ParticleId of liquid_world.rs:211-280: (is_boundary, handle = slot, particle index)
    struct ParticleId { bool boundary; size_t handle; uint32_t index; };
    // LiquidWorld::particles_intersecting_aabb — liquid_world.rs:210-243
    std::vector<ParticleId> particles_intersecting_aabb(const Vec3& mins, const Vec3& maxs) {
        sync_for_query();
        return run_query([&](uint64_t cap, uint32_t* k, uint32_t* s, uint32_t* i) {
            return salva_hip_particles_intersecting_aabb(w_, mins.data(), maxs.data(), cap, k, s, i);
        });
    }
    // LiquidWorld::particles_intersecting_shape for a ball / cuboid — liquid_world.rs:245-280
    std::vector<ParticleId> particles_intersecting_shape(const Vec3& translation, const std::array<Real, 4>& rotation_ijkw,
                                                         const SalvaHipShape& shape) {
        sync_for_query();
        return run_query([&](uint64_t cap, uint32_t* k, uint32_t* s, uint32_t* i) {
            return salva_hip_particles_intersecting_shape(w_, translation.data(), rotation_ijkw.data(), &shape, cap, k, s, i);
        });
    }
    // ... and for any other shape (the reference's query is generic over parry's `Shape`): `compute_aabb()` = shape.compute_aabb(pos)
    // as (mins, maxs), `distance_to_point(n, points_xyz, out)` = shape.distance_to_point(pos, &pt, true) per point
    std::vector<ParticleId> particles_intersecting_host_shape(std::function<std::pair<Vec3, Vec3>()> compute_aabb,
                                                              std::function<void(uint32_t, const float*, float*)> distance_to_point) {
        sync_for_query();
        struct Ctx { decltype(compute_aabb)* a; decltype(distance_to_point)* d; } ctx{&compute_aabb, &distance_to_point};
        SalvaHipHostQueryShape sh{};
        sh.user = &ctx;
        sh.aabb = [](void* u, float* mins, float* maxs) {
            const auto box = (*static_cast<Ctx*>(u)->a)();
            for (int k = 0; k < 3; ++k) { mins[k] = box.first[k]; maxs[k] = box.second[k]; }
        };
        sh.distance = [](void* u, uint32_t n, const float* pts, float* out) { (*static_cast<Ctx*>(u)->d)(n, pts, out); };
        return run_query([&](uint64_t cap, uint32_t* k, uint32_t* s, uint32_t* i) {
            return salva_hip_particles_intersecting_host_shape(w_, &sh, cap, k, s, i);
        });
    }
    // `world.counters` of the reference (counters/mod.rs:17-72): nsubsteps, step_time, custom, stages, cd, solver
    // Counters::enable / disable (counters/mod.rs:56-72); disabled by default, as in the reference
    void enable_counters(bool enabled = true) { check(salva_hip_enable_counters(w_, enabled ? 1 : 0)); }
    // Opt-in CFL sub-stepping (timestep_manager.rs:36-46 + the clamp the reference left commented out at :90-93).  0 = off (the
    // reference as it runs: one substep per step), 1 = the commented code literally, 2 = the same, cut at the remaining time.
    void set_cfl_substepping(int mode = 1, float cfl_coeff = 0.4f, int min_num_substeps = 1, int max_num_substeps = 10) {
        check(salva_hip_set_cfl(w_, mode, cfl_coeff, min_num_substeps, max_num_substeps));
        cfl_mode_ = mode;
    }
    std::vector<float> substeps() const {  // substep lengths of the last step
        int64_t n = salva_hip_get_substeps(w_, nullptr, 0);  // (the count first: max_num_substeps is the caller's to choose)
        if (n < 0) check((int)n);
        std::vector<float> v((size_t)n);
        if (n) n = salva_hip_get_substeps(w_, v.data(), v.size());
        if (n < 0) check((int)n);
        return v;
    }
    SalvaHipCounters counters_tree() const {
        SalvaHipCounters c{};
        check(salva_hip_get_counters(w_, &c));
        return c;
    }

    // ---- x-slab decomposition (salva_hip_set_domain).  This world becomes the slab of cell planes
    // [cell_lo, cell_hi] (cell = floor(x / h)) of a domain cut along x; call after adding this rank's fluids (the same
    // fluid slots on every rank) and the boundary particles within three cells of its slab, before the first step.
    // `gid_offset` + upload index is a particle's global id.  From then on Fluid::positions / velocities are no longer
    // refreshed (particles migrate between ranks): read them with owned().
    void set_domain(Comm& comm, int cell_lo, int cell_hi, uint32_t gid_offset = 0) {
        for (size_t s = 0; s < fluids_.size(); ++s) upload(fluids_[s], (uint32_t)s);
        for (size_t s = 0; s < boundaries_.size(); ++s) upload(boundaries_[s], (uint32_t)s);
        check(salva_hip_set_domain(w_, comm.handle(), cell_lo, cell_hi, gid_offset));
        decomposed_ = true;
    }
    struct OwnedParticles {
        std::vector<uint32_t> gids, fluid_slots;
        std::vector<Vec3> positions, velocities;
    };
    // the particles this rank owns after the last step, in no particular order
    OwnedParticles owned() {
        OwnedParticles o;
        size_t cap = (size_t)stats_.nparticles + 1024;
        for (;;) {
            o.gids.resize(cap); o.fluid_slots.resize(cap); o.positions.resize(cap); o.velocities.resize(cap);
            const int64_t m = salva_hip_get_owned(w_, (uint32_t)cap, o.gids.data(), o.positions[0].data(), o.velocities[0].data(),
                                                  o.fluid_slots.data());
            if (m < 0) check((int)m);
            if ((size_t)m <= cap) { cap = (size_t)m; break; }
            cap = (size_t)m;
        }
        o.gids.resize(cap); o.fluid_slots.resize(cap); o.positions.resize(cap); o.velocities.resize(cap);
        return o;
    }
    // collective re-cut of the slabs for equal particle counts; returns this rank's new [cell_lo, cell_hi] (the caller
    // re-uploads the boundary particles the new slab needs)
    std::pair<int, int> rebalance() {
        int32_t lo = 0, hi = 0;
        check(salva_hip_rebalance(w_, &lo, &hi));
        return {lo, hi};
    }
    // collective, between the same two steps on every rank (empty where there is nothing to do): particles appended to this
    // rank get the next free global ids; the listed ids this rank owns are gone from the next step on (faucet3.rs:69-104)
    void add_owned(FluidHandle h, const std::vector<Vec3>& positions, const std::vector<Vec3>* velocities = nullptr) {
        if (velocities && velocities->size() != positions.size())
            throw Error(SALVA_HIP_E_INVALID, "The provided positions and velocities arrays must have the same length.");
        const bool any = !positions.empty();
        check(salva_hip_add_particles(w_, (uint32_t)h, positions.size(), any ? positions[0].data() : nullptr,
                                      any && velocities ? (*velocities)[0].data() : nullptr));
    }
    int64_t delete_owned(const std::vector<uint32_t>& gids) {
        const int64_t m = salva_hip_delete_owned(w_, (uint32_t)gids.size(), gids.empty() ? nullptr : gids.data());
        if (m < 0) check((int)m);
        return m;
    }

  private:
    void sync_for_query() {  // a query is not a step: objects are uploaded, pending deletions stay pending
        if (decomposed_) {
            for (size_t s = 0; s < boundaries_.size(); ++s) upload(boundaries_[s], (uint32_t)s);
            return;
        }
        for (size_t s = 0; s < fluids_.size(); ++s) upload(fluids_[s], (uint32_t)s);
        for (size_t s = 0; s < boundaries_.size(); ++s) upload(boundaries_[s], (uint32_t)s);
    }
    template <typename F>
    std::vector<ParticleId> run_query(F&& call) {
        std::vector<uint32_t> k(1024), s(1024), i(1024);
        int64_t total;
        for (;;) {
            total = call((uint64_t)k.size(), k.data(), s.data(), i.data());
            if (total < 0) check((int)total);
            if ((size_t)total <= k.size()) break;
            k.resize(total); s.resize(total); i.resize(total);
        }
        std::vector<ParticleId> out((size_t)total);
        for (size_t q = 0; q < out.size(); ++q) out[q] = ParticleId{k[q] != 0, s[q], i[q]};
        return out;
    }

  public:
    // LiquidWorld::step(dt, gravity) — liquid_world.rs:62-158
    void step(Real dt, const Vec3& gravity) {
        if (decomposed_) {  // the particles live on the device and change owner: read them with owned()
            for (size_t s = 0; s < boundaries_.size(); ++s) upload(boundaries_[s], (uint32_t)s);
            check(salva_hip_step(w_, dt, gravity.data(), &stats_));
            return;
        }
        for (size_t s = 0; s < fluids_.size(); ++s) upload(fluids_[s], (uint32_t)s);
        for (size_t s = 0; s < boundaries_.size(); ++s) upload(boundaries_[s], (uint32_t)s);
        const int rc = salva_hip_step(w_, dt, gravity.data(), &stats_);
        for (size_t s = 0; s < fluids_.size(); ++s) {  // the reference's host arrays are current after every step
            Fluid& f = fluids_[s];
            if (f.num_particles())
                check(salva_hip_get_fluid(w_, (uint32_t)s, f.positions[0].data(), f.velocities[0].data()));
            for (auto& a : f.accelerations) a = Vec3{0, 0, 0};
        }
        check(rc);
    }
    // The working set as it is (salva_hip_get_local): every particle this world holds — on a rank of a decomposed run the owned
    // particles and the ghosts — in the order of the last step's cell sort; what a user NonPressureForce works on there.
    struct LocalView {
        std::vector<uint32_t> ids, fluid_slots;
        std::vector<uint8_t> is_ghost;
        std::vector<Vec3> positions, velocities;
        std::vector<Real> densities, volumes;
    };
    LocalView local_view() {
        const size_t n = (size_t)salva_hip_local_len(w_);
        LocalView v;
        v.ids.resize(n); v.fluid_slots.resize(n); v.is_ghost.resize(n); v.positions.resize(n); v.velocities.resize(n);
        v.densities.resize(n); v.volumes.resize(n);
        if (n)
            check(salva_hip_get_local(w_, v.ids.data(), v.fluid_slots.data(), v.is_ghost.data(), v.positions[0].data(), v.velocities[0].data(),
                                      v.densities.data(), v.volumes.data()));
        return v;
    }
    // CSR contact lists over the local view: j = local index (fluid-fluid) / index in boundary j_model's arrays (fluid-boundary)
    void local_contacts(bool boundary, std::vector<uint64_t>& offsets, std::vector<uint32_t>& j_model, std::vector<uint32_t>& j) {
        offsets.assign((size_t)salva_hip_local_len(w_) + 1, 0);
        const int64_t total = salva_hip_get_local_contacts(w_, boundary ? 1 : 0, offsets.data(), nullptr, nullptr, 0);
        if (total < 0) check((int)total);
        j_model.assign((size_t)total, 0); j.assign((size_t)total, 0);
        if (total) {
            const int64_t rc = salva_hip_get_local_contacts(w_, boundary ? 1 : 0, offsets.data(), j_model.data(), j.data(), (uint64_t)total);
            if (rc < 0) check((int)rc);
        }
    }
    void force_add_local_accelerations(const std::vector<Vec3>& acc) { check(salva_hip_force_add_local_accelerations(w_, acc[0].data())); }
    // Asynchronous read-back of one fluid (salva_hip_get_fluid_async): start it after a step, run the next step, collect the
    // arrays of the earlier state with wait_download().  `positions` / `velocities` must hold num_particles() entries and
    // stay untouched until the wait; pin them with host_register() once for a read-back at PCIe speed.
    void download_async(FluidHandle h, Vec3* positions, Vec3* velocities) {
        check(salva_hip_get_fluid_async(w_, (uint32_t)h, positions ? positions[0].data() : nullptr, velocities ? velocities[0].data() : nullptr));
    }
    void wait_download() { check(salva_hip_wait_download(w_)); }
    void host_register(void* p, size_t bytes) { check(salva_hip_host_register(w_, p, bytes)); }
    static void host_unregister(void* p) { check(salva_hip_host_unregister(p)); }
    // LiquidWorld::step_with_coupling (liquid_world.rs:67-158): update_boundaries -> the substep -> transmit_forces
    void step_with_coupling(Real dt, const Vec3& gravity, CouplingManager& coupling) {
        for (size_t s = 0; s < boundaries_.size(); ++s) upload(boundaries_[s], (uint32_t)s);
        if (cfl_mode_) {
            // CFL sub-stepping: the manager's two calls belong inside the substep loop (liquid_world.rs:94-103, :146) — the library calls back
            struct Ctx { LiquidWorld* w; CouplingManager* c; std::exception_ptr err; } ctx{this, &coupling, nullptr};
            auto thunk = [](void* user, SalvaHipWorld*, int32_t phase, float sub_dt) -> int {
                Ctx& x = *static_cast<Ctx*>(user);
                try {
                    if (phase == 0) x.c->update_boundaries(*x.w);
                    else x.c->transmit_forces(*x.w, sub_dt);
                    return 0;
                } catch (...) { x.err = std::current_exception(); return 1; }  // (never unwind through C)
            };
            check(salva_hip_set_coupling_callback(w_, thunk, &ctx));
            try { step(dt, gravity); } catch (...) {
                salva_hip_set_coupling_callback(w_, nullptr, nullptr);
                if (ctx.err) std::rethrow_exception(ctx.err);
                throw;
            }
            salva_hip_set_coupling_callback(w_, nullptr, nullptr);
            return;
        }
        coupling.update_boundaries(*this);
        step(dt, gravity);
        coupling.transmit_forces(*this, dt);
    }
    // boundary.volumes / boundary.forces (and, for sampled boundaries, positions / velocities) after a step
    void sync_boundary(BoundaryHandle h) {
        Boundary& b = boundaries_[h];
        if (b.dynamic_shape.kind) b.dynamic_n_ = (size_t)salva_hip_boundary_len(w_, (uint32_t)h);
        if (!b.num_particles()) { b.positions.clear(); b.velocities.clear(); b.volumes.clear(); b.forces.clear(); return; }
        b.volumes.resize(b.num_particles());
        if (b.wants_forces) b.forces.resize(b.num_particles());
        check(salva_hip_get_boundary(w_, (uint32_t)h, b.volumes.data(), b.wants_forces ? b.forces[0].data() : nullptr));
        if (!b.sampling.empty() || b.dynamic_shape.kind) {
            b.positions.resize(b.num_particles()); b.velocities.resize(b.num_particles());
            check(salva_hip_get_boundary_particles(w_, (uint32_t)h, b.positions[0].data(), b.velocities[0].data()));
        }
    }
    // the two halves of ColliderCouplingManager for one boundary (fluids_pipeline.rs:160-193 / :266-287)
    void update_boundary_pose(BoundaryHandle h, const SalvaHipRigidPose& pose) {
        upload(boundaries_[h], (uint32_t)h);
        if (pose.has_body) boundaries_[h].wants_forces = pose.is_dynamic != 0;
        check(salva_hip_update_boundary_pose(w_, (uint32_t)h, &pose));
    }
    // `unregister_coupling` as the library sees it (salva_hip_clear_boundary_sampling): the boundary keeps the particles it holds
    // and becomes a plain boundary; call it before the memory behind Boundary::dynamic_host's callbacks / user pointer goes away
    void clear_boundary_sampling(BoundaryHandle h) {
        Boundary& b = boundaries_[h];
        upload(b, (uint32_t)h);
        const uint64_t n = salva_hip_boundary_len(w_, (uint32_t)h);
        b.positions.assign(n, Vec3{0, 0, 0}); b.velocities.assign(n, Vec3{0, 0, 0});
        if (n) check(salva_hip_get_boundary_particles(w_, (uint32_t)h, b.positions[0].data(), b.velocities[0].data()));
        check(salva_hip_clear_boundary_sampling(w_, (uint32_t)h));
        b.sampling.clear();
        b.dynamic_shape = SalvaHipShape{};
        b.dynamic_host = SalvaHipHostShape{nullptr, nullptr, nullptr};
        b.dirty_ = false;  // (the device holds exactly these particles)
    }
    void boundary_wrench(BoundaryHandle h, const Vec3& point, Vec3& force, Vec3& torque) {
        check(salva_hip_get_boundary_wrench(w_, (uint32_t)h, point.data(), force.data(), torque.data()));
    }

  private:
    // Objects added since the last step exist only on the host: a swap-remove must see the same dense sets on both sides.
    // Pending particle deletions stay pending (they are applied at the top of the next step, fluid.rs:88-98).
    void upload_new_objects() {
        for (size_t s = salva_hip_num_fluids(w_); s < fluids_.size(); ++s) {
            Fluid& f = fluids_[s];
            const size_t n = f.num_particles();
            check(salva_hip_set_fluid(w_, (uint32_t)s, n, n ? f.positions[0].data() : nullptr, n ? f.velocities[0].data() : nullptr,
                                      n ? f.volumes.data() : nullptr, n ? f.accelerations[0].data() : nullptr, nullptr, f.density0,
                                      f.interaction_groups.memberships, f.interaction_groups.filter, (uint32_t)SALVA_HIP_DIRTY_ALL));
            f.structural_ = false; f.dirty_ = 0; f.appended_from_ = Fluid::kNone;
        }
        for (size_t s = 0; s < boundaries_.size(); ++s) upload(boundaries_[s], (uint32_t)s);
    }
    void upload(Fluid& f, uint32_t slot) {
        std::vector<Vec3> dv;  // solver.velocity_changes of the surviving particles (init_with_fluids, dfsph_solver.rs:526-561)
        std::vector<Real> pr;  // ... and the pressures IISPH warm-starts from (iisph_solver.rs:35, :499-536)
        bool have_dv = false;
        bool any_deleted = false;
        for (bool d : f.deleted_) any_deleted |= d;
        bool on_device = slot < salva_hip_num_fluids(w_);
        if (!on_device && any_deleted) {
            // never uploaded: upload uncompacted first — the reference resizes the slot's (possibly inherited) solver buffer
            // to the full particle count and filters afterwards (dfsph_solver.rs:543-560); the deletion replays below
            const size_t n = f.num_particles();
            check(salva_hip_set_fluid(w_, slot, n, n ? f.positions[0].data() : nullptr, n ? f.velocities[0].data() : nullptr,
                                      n ? f.volumes.data() : nullptr, n ? f.accelerations[0].data() : nullptr, nullptr, f.density0,
                                      f.interaction_groups.memberships, f.interaction_groups.filter, (uint32_t)SALVA_HIP_DIRTY_ALL));
            f.structural_ = false; f.dirty_ = 0; f.appended_from_ = Fluid::kNone;
            on_device = true;
        }
        if (on_device && !f.structural_ && f.dirty_ && (any_deleted || f.appended_from_ != Fluid::kNone)) {
            // host edits of the particles the device already holds go up first, so that appends and deletions can replay on
            // the device (only there do new particles see what `velocity_changes[slot].resize(n)` would hand them)
            const size_t n0 = f.appended_from_ != Fluid::kNone ? f.appended_from_ : f.num_particles();
            if (n0 == salva_hip_fluid_len(w_, slot)) {
                check(salva_hip_set_fluid(w_, slot, n0, n0 ? f.positions[0].data() : nullptr, n0 ? f.velocities[0].data() : nullptr,
                                          n0 ? f.volumes.data() : nullptr, n0 ? f.accelerations[0].data() : nullptr, nullptr, f.density0,
                                          f.interaction_groups.memberships, f.interaction_groups.filter, f.dirty_));
                f.dirty_ = 0;
            }
        }
        if (on_device && !f.structural_ && !f.dirty_ && (any_deleted || f.appended_from_ != Fluid::kNone)) {
            // replay the edits on the device: append (fluid.rs:126-150), then compact (fluid.rs:88-98) — nothing the
            // fluid already holds travels over PCIe
            if (f.appended_from_ != Fluid::kNone) {
                const size_t a = f.appended_from_, k = f.positions.size() - a;
                if (k) check(salva_hip_add_particles(w_, slot, k, f.positions[a].data(), f.velocities[a].data()));
                f.appended_from_ = Fluid::kNone;
            }
            if (any_deleted) {
                std::vector<uint8_t> mask(f.deleted_.size());
                for (size_t i = 0; i < mask.size(); ++i) mask[i] = f.deleted_[i] ? 1 : 0;
                const int64_t kept = salva_hip_delete_particles(w_, slot, mask.data());
                if (kept < 0) check((int)kept);
                size_t k = 0;
                for (size_t i = 0; i < f.positions.size(); ++i) {
                    if (f.deleted_[i]) continue;
                    f.positions[k] = f.positions[i]; f.velocities[k] = f.velocities[i];
                    f.accelerations[k] = f.accelerations[i]; f.volumes[k] = f.volumes[i];
                    ++k;
                }
                f.positions.resize(k); f.velocities.resize(k); f.accelerations.resize(k); f.volumes.resize(k);
                f.deleted_.assign(k, false);
                any_deleted = false;
            }
        } else if (any_deleted || f.appended_from_ != Fluid::kNone) {
            f.structural_ = true;  // mixed with other host edits: take the full re-upload path below
            f.appended_from_ = Fluid::kNone;
        }
        if (f.structural_ && on_device) {
            const uint64_t old_n = salva_hip_fluid_len(w_, slot);
            if (old_n) {
                dv.assign(old_n, Vec3{0, 0, 0});
                check(salva_hip_get_fluid_field(w_, slot, SALVA_HIP_FIELD_VELOCITY_CHANGE, dv[0].data()));
                dv.resize(f.num_particles(), Vec3{0, 0, 0});
                pr.assign(old_n, 0.0f);
                check(salva_hip_get_fluid_field(w_, slot, SALVA_HIP_FIELD_PRESSURE, pr.data()));
                pr.resize(f.num_particles(), 0.0f);
                have_dv = true;
            }
        }
        if (any_deleted) {  // Fluid::apply_particles_removal (fluid.rs:88-98) = stable compaction (helper.rs:4-12)
            size_t k = 0;
            for (size_t i = 0; i < f.positions.size(); ++i) {
                if (f.deleted_[i]) continue;
                f.positions[k] = f.positions[i]; f.velocities[k] = f.velocities[i];
                f.accelerations[k] = f.accelerations[i]; f.volumes[k] = f.volumes[i];
                if (have_dv) { dv[k] = dv[i]; pr[k] = pr[i]; }
                ++k;
            }
            f.positions.resize(k); f.velocities.resize(k); f.accelerations.resize(k); f.volumes.resize(k);
            if (have_dv) { dv.resize(k); pr.resize(k); }
            f.deleted_.assign(k, false);
        }
        if (f.structural_ || f.dirty_) {
            const size_t n = f.num_particles();
            const float* acc = nullptr;
            for (const Vec3& a : f.accelerations) if (a[0] != 0 || a[1] != 0 || a[2] != 0) { acc = f.accelerations[0].data(); break; }
            check(salva_hip_set_fluid(w_, slot, n, n ? f.positions[0].data() : nullptr, n ? f.velocities[0].data() : nullptr,
                                      n ? f.volumes.data() : nullptr, acc, (have_dv && n) ? dv[0].data() : nullptr, f.density0,
                                      f.interaction_groups.memberships, f.interaction_groups.filter,
                                      f.structural_ ? (uint32_t)SALVA_HIP_DIRTY_ALL : f.dirty_));
            if (have_dv && n) {
                bool any = false;
                for (Real x : pr) any |= x != 0.0f;
                if (any) check(salva_hip_set_fluid_field(w_, slot, SALVA_HIP_FIELD_PRESSURE, pr.data()));
            }
            f.structural_ = false;
            f.dirty_ = 0;
        }
        std::vector<SalvaHipForceDesc> descs;
        for (auto& np : f.nonpressure_forces) descs.push_back(np->desc());
        check(salva_hip_set_fluid_forces(w_, slot, descs.data(), (uint32_t)descs.size()));
    }
    void upload(Boundary& b, uint32_t slot) {
        if (!b.dirty_) return;
        const size_t n = b.num_particles();
        if (b.dynamic_shape.kind == SALVA_HIP_SHAPE_HOST) {
            check(salva_hip_set_boundary_dynamic_sampling_host(w_, slot, &b.dynamic_host, b.interaction_groups.memberships,
                                                               b.interaction_groups.filter));
            b.dirty_ = false;
            return;
        }
        if (b.dynamic_shape.kind) {
            check(salva_hip_set_boundary_dynamic_sampling(w_, slot, &b.dynamic_shape, b.interaction_groups.memberships,
                                                          b.interaction_groups.filter));
            b.dirty_ = false;
            return;
        }
        if (!b.sampling.empty()) {
            check(salva_hip_set_boundary_sampling(w_, slot, n, b.sampling[0].data(), b.interaction_groups.memberships,
                                                  b.interaction_groups.filter));
            b.dirty_ = false;
            return;
        }
        check(salva_hip_set_boundary(w_, slot, n, n ? b.positions[0].data() : nullptr, n ? b.velocities[0].data() : nullptr,
                                     b.interaction_groups.memberships, b.interaction_groups.filter, b.wants_forces ? 1 : 0));
        b.dirty_ = false;
    }

    SalvaHipWorld* w_ = nullptr;
    int cfl_mode_ = 0;  // set_cfl_substepping
    Real particle_radius_;
    std::vector<Fluid> fluids_;
    std::vector<Boundary> boundaries_;
    SalvaHipStepStats stats_{};
    bool decomposed_ = false;  // set_domain was called: particles are read with owned(), fluids are not re-uploaded
};

// ColliderCouplingSet / ColliderCouplingManager (integrations/rapier/fluids_pipeline.rs:64-288) without the rapier types: each
// entry reads the collider's pose through `pose()` and hands the step's impulse back through `apply(linear, angular)`
// (= body.apply_impulse(force * dt), body.apply_torque_impulse(torque * dt) about pose.world_com).
class ColliderCouplingSet : public CouplingManager {
  public:
    struct Entry {
        BoundaryHandle boundary;
        std::function<SalvaHipRigidPose()> pose;
        std::function<void(const Vec3& impulse, const Vec3& torque_impulse)> apply;  // may be empty (kinematic / fixed bodies)
    };
    // register_coupling(boundary, collider, sampling_method): StaticSampling(points) -> the points live in Boundary::sampling;
    // DynamicContactSampling -> the collider's shape lives in Boundary::dynamic_shape
    void register_coupling(BoundaryHandle boundary, std::function<SalvaHipRigidPose()> pose,
                           std::function<void(const Vec3&, const Vec3&)> apply = {}) {
        entries_.push_back(Entry{boundary, std::move(pose), std::move(apply)});
    }
    void unregister_coupling(BoundaryHandle boundary) {
        for (size_t k = 0; k < entries_.size(); ++k)
            if (entries_[k].boundary == boundary) { entries_.erase(entries_.begin() + (long)k); return; }
    }
    void update_boundaries(LiquidWorld& world) override {
        poses_.clear();
        for (Entry& e : entries_) {
            poses_.push_back(e.pose());
            world.update_boundary_pose(e.boundary, poses_.back());
        }
    }
    void transmit_forces(LiquidWorld& world, Real dt) override {
        for (size_t k = 0; k < entries_.size(); ++k) {
            const SalvaHipRigidPose& p = poses_[k];
            if (!entries_[k].apply || !p.has_body || !p.is_dynamic) continue;
            Vec3 f{0, 0, 0}, t{0, 0, 0};
            world.boundary_wrench(entries_[k].boundary, Vec3{p.world_com[0], p.world_com[1], p.world_com[2]}, f, t);
            entries_[k].apply(Vec3{f[0] * dt, f[1] * dt, f[2] * dt}, Vec3{t[0] * dt, t[1] * dt, t[2] * dt});
        }
    }

  private:
    std::vector<Entry> entries_;
    std::vector<SalvaHipRigidPose> poses_;
};

// FluidsPipeline (integrations/rapier/fluids_pipeline.rs:18-61): the liquid world (always DFSPH there, :35) and the coupling
// set in one object; step(gravity, dt) = liquid_world.step_with_coupling(dt, gravity, coupling manager) (:48-60).  The rapier
// ColliderSet / RigidBodySet arguments of the reference are the closures held by the coupling entries here.
struct FluidsPipeline {
    LiquidWorld liquid_world;
    ColliderCouplingSet coupling;
    FluidsPipeline(Real particle_radius, Real smoothing_factor) : liquid_world(DFSPHSolver(), particle_radius, smoothing_factor) {}
    void step(const Vec3& gravity, Real dt) { liquid_world.step_with_coupling(dt, gravity, coupling); }
};

}  // namespace salva
