// salva_oracle.cpp — CPU restatement of dimforge/salva's `LiquidWorld::step` hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
// `bench.py`'s `cpu_baseline` leg may load this library, and only as the checker / CPU baseline.
// The shipped path (`salva_amd/csrc/libsalva_hip.so`) never links, loads or calls it.
//
// PARITY UNPINNED: the reference (pure Rust; needs cargo + nalgebra 0.33, rayon 1.8, fnv 1.0 …) cannot
// be compiled in this environment, and its repository holds no golden vector, known-answer test or
// fixture for this path (its only 3 `#[test]`s never touch the solver).  This file therefore restates
// the algorithm from the source text alone; it is pinned only by the analytic self-checks in
// `tests/test_oracle.py` (lattice K = 33, rho ~= rho0, kernel normalisation, contact symmetry, momentum)
// and by the committed vectors under `tests/golden/` that were produced by *this* oracle.
//
// What is restated (paths relative to /root/reference):
//   src/liquid_world.rs:62-158                      step_with_coupling  (coupling = `()`, no-op)
//   src/geometry/hgrid.rs:41-63                     cell key = floor(x / h), hash grid of Vec<entry>
//   src/geometry/contacts.rs:133-400                grid insertion, 14-cell half stencil, directed contacts
//   src/solver/helper.rs:9-65                       weight / gradient evaluation for each stored contact
//   src/kernel/kernel.rs:13-24, cubic_spline_kernel.rs:12-79
//   src/solver/pressure/dfsph_solver.rs:54-708      DFSPH
//   src/solver/pressure/iisph_solver.rs:48-711      IISPH
//   src/solver/viscosity/xsph_viscosity.rs:31-95, artificial_viscosity.rs:29-124
//   src/solver/viscosity/dfsph_viscosity.rs:38-327                      DFSPHViscosity (with nalgebra's 6x6 LU restated)
//   src/solver/surface_tension/akinci2013_surface_tension.rs:43-192
//   src/solver/surface_tension/he2014_surface_tension.rs:40-181, wcsph_surface_tension.rs:32-87
//   src/solver/nonpressure_force.rs:10-30            user forces: a host callback at its place in the force list
//   src/timestep_manager.rs:23-94                   (CFL is bypassed upstream: one substep per step)
//   src/object/{fluid,boundary,interaction_groups}.rs   incl. add_particles / delete_particle_at_next_timestep /
//                                                   apply_particles_removal (fluid.rs:71-150), solver-side filter_from_mask
//   src/liquid_world.rs:171-178, object/contiguous_arena.rs:118-135    remove_fluid / remove_boundary (swap-remove; the
//                                                   solver's per-slot buffers stay positional)
//   src/integrations/rapier/fluids_pipeline.rs:160-193, 262-287        StaticSampling arm of the rigid-body coupling
//                                                   (rapier's velocity_at_point / apply_impulse_at_point restated)
//   src/integrations/rapier/fluids_pipeline.rs:193-259                 DynamicContactSampling arm for ball / cuboid colliders
//                                                   (parry3d 0.18's compute_aabb / project_point restated, see update_boundaries_dynamic)
//
// Third-party arithmetic that is NOT under /root/reference (nalgebra 0.33, semver range only, no lockfile):
//   dot / norm_squared of a 3-vector = ((x0*y0 + x1*y1) + x2*y2); Unit::try_new_and_get(v, eps) returns
//   None iff |v|^2 <= eps^2, else (v / |v| component-wise, |v|).  Rust never contracts a*b+c into an FMA,
//   so this file must be compiled with -ffp-contract=off.
//
// The one thing that cannot be restated is the reference's *summation order*: contacts are pushed in
// hashbrown-bucket (and, with `parallel`, thread-schedule) order.  Here cells are visited in first-insertion
// order (deterministic); `shuffle_seed != 0` permutes the cell visit order so that tests can measure how
// much a different—equally legitimate—order moves the result (the "self noise" of the reference).
//
// Build:  see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp -shared -fPIC).

#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <unordered_map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace so {

// ---------------------------------------------------------------------------------------------------
// Small vector type with nalgebra's evaluation order.
// ---------------------------------------------------------------------------------------------------
template <typename R>
struct V3 {
    R x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(R a, R b, R c) : x(a), y(b), z(c) {}
    V3 operator+(const V3& o) const { return V3(x + o.x, y + o.y, z + o.z); }
    V3 operator-(const V3& o) const { return V3(x - o.x, y - o.y, z - o.z); }
    V3 operator-() const { return V3(-x, -y, -z); }
    V3 operator*(R s) const { return V3(x * s, y * s, z * s); }
    V3 operator/(R s) const { return V3(x / s, y / s, z / s); }
    V3& operator+=(const V3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    V3& operator-=(const V3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    V3& operator*=(R s) { x *= s; y *= s; z *= s; return *this; }
    R dot(const V3& o) const { return (x * o.x + y * o.y) + z * o.z; }
    R norm_squared() const { return (x * x + y * y) + z * z; }
    V3 cross(const V3& o) const { return V3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
    R norm() const { return std::sqrt(norm_squared()); }
    void fill(R v) { x = y = z = v; }
};

template <typename R> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920929e-7f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };

// llvm.powi / compiler-rt __powisf2: square-and-multiply.
template <typename R>
static inline R powi(R a, int b) {
    const bool recip = b < 0;
    R r = 1;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}

// nalgebra Unit::try_new_and_get
template <typename R>
static inline bool try_new_and_get(const V3<R>& v, R min_norm, V3<R>& dir, R& norm) {
    R sq = v.norm_squared();
    if (sq > min_norm * min_norm) {
        norm = std::sqrt(sq);
        dir = v / norm;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------------
// kernel/cubic_spline_kernel.rs:12-33,55-79 ; kernel/kernel.rs:13-24
// ---------------------------------------------------------------------------------------------------
template <typename R>
struct CubicSpline {
    static constexpr R PI = (R)3.14159265358979323846264338327950288;
    static R scalar_apply(R r, R h) {
        R normalizer = (R)8.0 / (PI * h * h * h);
        R q = r / h;
        R rhs;
        if (q <= (R)0.5) {
            R q2 = q * q;
            rhs = (R)1 + (q2 * q - q2) * (R)6.0;
        } else if (q <= (R)1) {
            rhs = powi<R>((R)1 - q, 3) * (R)2;
        } else {
            rhs = 0;
        }
        return normalizer * rhs;
    }
    static R scalar_apply_diff(R r, R h) {
        R normalizer = (R)8.0 / (PI * h * h * h);
        R q = r / h;
        R rhs;
        if (q > (R)1 || q <= (R)1.0e-5) {
            rhs = 0;
        } else if (q <= (R)0.5) {
            rhs = (q * (R)3 - (R)2) * q * (R)6.0;
        } else {
            R one_q = (R)1 - q;
            rhs = -one_q * one_q * (R)6.0;
        }
        return normalizer * rhs / h;
    }
    static R apply(const V3<R>& v, R h) { return scalar_apply(v.norm(), h); }
    static V3<R> apply_diff(const V3<R>& v, R h) {
        V3<R> dir; R n;
        if (try_new_and_get<R>(v, Eps<R>::v, dir, n)) return dir * scalar_apply_diff(n, h);
        return V3<R>();
    }
};

// ---------------------------------------------------------------------------------------------------
// kernel/poly6_kernel.rs:8-40, kernel/spiky_kernel.rs:8-38, kernel/viscosity_kernel.rs:8-50 (dim3 normalizers)
// ---------------------------------------------------------------------------------------------------
template <typename R>
struct Poly6 {
    static R scalar_apply(R r, R h) {
        R normalizer = (R)(315.0 / 64.0) / (CubicSpline<R>::PI * powi<R>(h, 9));
        if (r <= h) return normalizer * powi<R>(h * h - r * r, 3);
        return 0;
    }
    static R scalar_apply_diff(R r, R h) {
        R normalizer = (R)(315.0 / 64.0) / (CubicSpline<R>::PI * powi<R>(h, 9));
        if (r <= h) return normalizer * powi<R>(h * h - r * r, 2) * r * (R)-6.0;
        return 0;
    }
};
template <typename R>
struct Spiky {
    static R scalar_apply(R r, R h) {
        R normalizer = (R)15.0 / (CubicSpline<R>::PI * powi<R>(h, 6));
        if (r <= h) return normalizer * powi<R>(h - r, 3);
        return 0;
    }
    static R scalar_apply_diff(R r, R h) {
        R normalizer = (R)15.0 / (CubicSpline<R>::PI * powi<R>(h, 6));
        if (r <= h) return -normalizer * powi<R>(h - r, 2) * (R)3.0;
        return 0;
    }
};
template <typename R>
struct ViscosityK {
    static R scalar_apply(R r, R h) {
        const R _2 = 2, normalizer = (R)15.0 / (_2 * CubicSpline<R>::PI * powi<R>(h, 3));
        if (r > (R)0 && r <= h) {
            R rr_hh = r * r / (h * h);
            return normalizer * (rr_hh * ((R)1 - r / (_2 * h)) + h / (_2 * r) - (R)1);
        }
        return 0;
    }
    static R scalar_apply_diff(R r, R h) {
        const R _2 = 2, _3 = 3, normalizer = (R)15.0 / (_2 * CubicSpline<R>::PI * powi<R>(h, 3));
        if (r > (R)0 && r <= h) {
            R rr = r * r, hh = h * h, hhh = hh * h;
            return normalizer * (-_3 * rr / (_2 * hhh) + _2 * r / hh - h / (_2 * rr));
        }
        return 0;
    }
};
// the KernelDensity / KernelGradient type parameters of the solvers as run-time kinds (0 cubic spline, 1 poly6, 2 spiky,
// 3 viscosity); Kernel::apply / apply_diff (kernel.rs:13-24)
template <typename R>
static R kernel_scalar(int kind, bool diff, R r, R h) {
    switch (kind) {
        case 1: return diff ? Poly6<R>::scalar_apply_diff(r, h) : Poly6<R>::scalar_apply(r, h);
        case 2: return diff ? Spiky<R>::scalar_apply_diff(r, h) : Spiky<R>::scalar_apply(r, h);
        case 3: return diff ? ViscosityK<R>::scalar_apply_diff(r, h) : ViscosityK<R>::scalar_apply(r, h);
        default: return diff ? CubicSpline<R>::scalar_apply_diff(r, h) : CubicSpline<R>::scalar_apply(r, h);
    }
}
template <typename R>
static R kernel_apply(int kind, const V3<R>& v, R h) { return kernel_scalar<R>(kind, false, v.norm(), h); }
template <typename R>
static V3<R> kernel_apply_diff(int kind, const V3<R>& v, R h) {
    V3<R> dir; R n;
    if (try_new_and_get<R>(v, Eps<R>::v, dir, n)) return dir * kernel_scalar<R>(kind, true, n, h);
    return V3<R>();
}

// ---------------------------------------------------------------------------------------------------
// object/interaction_groups.rs:64-69
// ---------------------------------------------------------------------------------------------------
struct Groups {
    uint32_t memberships = 1u, filter = 0xffffffffu;  // default: GROUP_1 / ALL (:72-79)
    bool test(const Groups& rhs) const {
        return (memberships & rhs.filter) != 0 && (rhs.memberships & filter) != 0;
    }
};

enum ForceKind { FORCE_XSPH = 1, FORCE_ARTIFICIAL = 2, FORCE_AKINCI2013 = 3, FORCE_DFSPH_VISCOSITY = 4, FORCE_HE2014 = 5, FORCE_WCSPH_TENSION = 6, FORCE_CUSTOM = 7 };

template <typename R>
struct Force {
    int kind = 0;
    // XSPH: p0 = fluid coeff, p1 = boundary coeff
    // Artificial: p0 = fluid coeff, p1 = boundary coeff, p2 = alpha, p3 = beta, p4 = speed_of_sound
    // Akinci2013: p0 = tension coeff, p1 = boundary adhesion coeff
    // DFSPHViscosity: p0 = viscosity_coefficient, p1 = min_viscosity_iter, p2 = max_viscosity_iter, p3 = max_viscosity_error
    R p[5] = {0, 0, 0, 0, 0};
    // He2014 / WCSPHSurfaceTension: p0 = fluid tension coeff, p1 = boundary tension coeff
    std::vector<V3<R>> normals;  // Akinci state (akinci2013_surface_tension.rs:22)
    std::vector<R> colors, gradcs;  // He2014 state (he2014_surface_tension.rs:15-16)
    // DFSPHViscosity state (dfsph_viscosity.rs:98-99): betas (6x6 row-major), strain-rate target / error (6 each)
    std::vector<R> betas, strain_target, strain_error;
    int last_visc_iters = 0;
    R last_visc_error = 0;
};

template <typename R>
struct Fluid {  // object/fluid.rs:12-34
    std::vector<V3<R>> positions, velocities, accelerations;
    std::vector<R> volumes;
    R density0 = 1000;
    Groups groups;
    std::vector<Force<R>> forces;
    std::vector<char> deleted;  // deleted_particles (fluid.rs:28-30); sized lazily
    size_t num_deleted = 0;
    size_t n() const { return positions.size(); }
    R particle_mass(size_t i) const { return volumes[i] * density0; }  // fluid.rs:183-185
    // delete_particle_at_next_timestep (fluid.rs:71-76)
    void delete_particle_at_next_timestep(size_t i) {
        deleted.resize(n(), 0);
        if (!deleted[i]) { deleted[i] = 1; ++num_deleted; }
    }
    // helper::filter_from_mask (helper.rs:4-12)
    template <typename T>
    void filter(std::vector<T>& v) const {
        size_t k = 0;
        for (size_t i = 0; i < v.size(); ++i)
            if (!(i < deleted.size() && deleted[i])) v[k++] = v[i];
        v.resize(k);
    }
    // apply_particles_removal (fluid.rs:88-98)
    void apply_particles_removal() {
        if (num_deleted == 0) return;
        filter(positions); filter(velocities); filter(accelerations); filter(volumes);
        deleted.assign(positions.size(), 0);
        num_deleted = 0;
    }
    // add_particles (fluid.rs:126-150); `default_volume` = default_particle_volume()
    void add_particles(size_t k, const float* pos, const float* vel, R default_volume) {
        const size_t n0 = n();
        for (size_t i = 0; i < k; ++i) {
            positions.push_back(V3<R>((R)pos[3 * i], (R)pos[3 * i + 1], (R)pos[3 * i + 2]));
            velocities.push_back(vel ? V3<R>((R)vel[3 * i], (R)vel[3 * i + 1], (R)vel[3 * i + 2]) : V3<R>());
        }
        accelerations.resize(n0 + k, V3<R>());
        volumes.resize(n0 + k, default_volume);
        deleted.resize(n0 + k, 0);
    }
};

template <typename R>
struct Boundary {  // object/boundary.rs:11-24
    std::vector<V3<R>> positions, velocities;
    std::vector<R> volumes;
    bool has_forces = false;
    std::vector<V3<R>> forces;
    Groups groups;
    std::vector<V3<R>> sampling;  // ColliderSampling::StaticSampling(points), integrations/rapier/fluids_pipeline.rs:36-41
    // ColliderSampling::DynamicContactSampling (:42-43): the collider's shape (1 = ball(radius), 2 = cuboid(half extents))
    // and its pose / body state as of the last so_update_boundary_pose; `source` = (fluid, particle) of each emitted point
    int dyn_kind = 0;
    R dyn_p[3] = {0, 0, 0};
    V3<R> pose_t, pose_qv, pose_linvel, pose_angvel, pose_com;
    R pose_qw = 1;
    bool pose_has_body = false;
    std::vector<std::pair<uint32_t, uint32_t>> dyn_source;
    size_t n() const { return positions.size(); }
};

template <typename R>
struct Contact {  // geometry/contacts.rs:40-55
    size_t i, i_model, j, j_model;
    R weight;
    V3<R> gradient;
    Contact flip() const { return Contact{j, j_model, i, i_model, weight, -gradient}; }
};

struct SpinLock {
    std::atomic_flag f = ATOMIC_FLAG_INIT;
    void lock() { while (f.test_and_set(std::memory_order_acquire)) {} }
    void unlock() { f.clear(std::memory_order_release); }
};

template <typename R>
struct ParticleContacts {  // contacts.rs:83-130 (RwLock<Vec<Contact>> per particle)
    std::vector<std::vector<Contact<R>>> contacts;
    std::unique_ptr<SpinLock[]> locks;
    size_t nlocks = 0;
    void reset(size_t n) {
        for (auto& c : contacts) c.clear();
        contacts.resize(n);
        if (nlocks < n) { locks.reset(new SpinLock[n]); nlocks = n; }
    }
    void push(size_t i, const Contact<R>& c, bool threaded) {
        if (threaded) { locks[i].lock(); contacts[i].push_back(c); locks[i].unlock(); }
        else contacts[i].push_back(c);
    }
    size_t len() const { size_t s = 0; for (auto& c : contacts) s += c.size(); return s; }
};

// ---------------------------------------------------------------------------------------------------
// geometry/hgrid.rs — hash grid.  FNV-1a with key 1820 over the 24 little-endian bytes of the cell.
// ---------------------------------------------------------------------------------------------------
struct Cell { int64_t x, y, z; bool operator==(const Cell& o) const { return x == o.x && y == o.y && z == o.z; } };
struct CellHash {
    size_t operator()(const Cell& c) const {
        uint64_t h = 1820ull;
        const int64_t v[3] = {c.x, c.y, c.z};
        const unsigned char* b = reinterpret_cast<const unsigned char*>(v);
        for (int i = 0; i < 24; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
        return (size_t)h;
    }
};
struct Entry { uint32_t model; uint32_t particle; bool is_boundary; };  // contacts.rs:14-19

struct HGrid {
    std::unordered_map<Cell, uint32_t, CellHash> index;  // cell -> slot in `cells`
    std::vector<Cell> keys;                               // first-insertion order
    std::vector<std::vector<Entry>> cells;
    size_t used = 0;
    void clear() {  // hgrid.rs:55-57
        index.clear(); keys.clear();
        for (size_t i = 0; i < used; ++i) cells[i].clear();
        used = 0;
    }
    template <typename R>
    static int64_t quantify(R value, R cell_width) {  // hgrid.rs:41-43
        return (int64_t)(double)std::floor(value / cell_width);
    }
    template <typename R>
    void insert(const V3<R>& p, R w, Entry e) {  // hgrid.rs:60-63
        Cell c{quantify<R>(p.x, w), quantify<R>(p.y, w), quantify<R>(p.z, w)};
        auto it = index.find(c);
        uint32_t slot;
        if (it == index.end()) {
            slot = (uint32_t)used++;
            index.emplace(c, slot);
            keys.push_back(c);
            if (cells.size() < used) cells.emplace_back();
        } else slot = it->second;
        cells[slot].push_back(e);
    }
    const std::vector<Entry>* cell(const Cell& c) const {
        auto it = index.find(c);
        return it == index.end() ? nullptr : &cells[it->second];
    }
};

struct StepStats {
    int n_div_iters;        // number of compute_velocity_changes_for_divergence applications
    int n_press_iters;      // number of compute_velocity_changes applications (DFSPH) / Jacobi iterations (IISPH)
    double div_error;       // last evaluated average divergence error
    double density_error;   // last evaluated average density error
    uint64_t ncontacts;     // liquid_world.rs:119
    double t_grid_ms, t_contacts_ms, t_kernels_ms, t_solver_ms, t_total_ms;
};

static inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------------
// The world.
// ---------------------------------------------------------------------------------------------------
template <typename R>
struct World {
    R particle_radius, h;
    int solver_kind;  // 0 = DFSPH, 1 = IISPH
    int nthreads = 1;
    int kernel_density = 0, kernel_gradient = 0;  // KernelDensity / KernelGradient (dfsph_solver.rs:17-20): kinds of kernel_scalar
    uint64_t shuffle_seed = 0;
    std::vector<Fluid<R>> fluids;
    std::vector<Boundary<R>> boundaries;
    HGrid grid;
    std::vector<ParticleContacts<R>> ff, fb, bb;  // contact_manager.rs:8-15
    // timestep_manager.rs:11-34
    R dt = 0, inv_dt = 0, total_step_size = 0, remaining_time = 0;
    // timestep_manager.rs:24-26: the CFL parameters of `new`.  cfl_mode 0 = the code as it runs today (compute_substep returns the
    // whole step, :88); 1 = the clamp the reference left commented out below its FIXME (:90-93), opt-in; 2 = the
    // same, additionally never stepping past the end of the step (the commented code overshoots: remaining_time goes negative).
    int cfl_mode = 0;
    R cfl_coeff = (R)0.4;
    int min_num_substeps = 1, max_num_substeps = 10;
    std::vector<double> substeps_of_last_step;
    // CouplingManager::{update_boundaries, transmit_forces} inside the substep loop (coupling_manager.rs:9-28), driven by the host
    void (*substep_cb)(void*, int, double) = nullptr;
    void* substep_user = nullptr;

    // DFSPH parameters (dfsph_solver.rs:54-70) / IISPH (iisph_solver.rs:48-64)
    int min_pressure_iter = 1, max_pressure_iter = 50;
    R max_density_error = (R)0.05;
    int min_divergence_iter = 1, max_divergence_iter = 50;
    R max_divergence_error = (R)0.1;
    size_t min_neighbors_for_divergence_solve = 20;
    R omega = (R)0.5;

    // solver scratch, one vector per fluid
    std::vector<std::vector<R>> alphas, densities, predicted_densities, divergences;
    std::vector<std::vector<V3<R>>> velocity_changes;
    std::vector<std::vector<R>> aii, pressures, next_pressures;
    std::vector<std::vector<V3<R>>> dii, dij_pjl;

    StepStats stats{};

    World(R radius, R smoothing, int kind) : particle_radius(radius), solver_kind(kind) {
        h = radius * smoothing * (R)2.0;  // liquid_world.rs:44
    }

    bool threaded() const { return nthreads > 1; }

    // ------------------------------------------------------------------ init_with_fluids (dfsph :526-561, iisph :479-537)
    void init_with_fluids() {
        size_t nf = fluids.size();
        alphas.resize(nf); densities.resize(nf); predicted_densities.resize(nf); divergences.resize(nf);
        velocity_changes.resize(nf); aii.resize(nf); pressures.resize(nf); next_pressures.resize(nf);
        dii.resize(nf); dij_pjl.resize(nf);
        for (size_t f = 0; f < nf; ++f) {
            size_t n = fluids[f].n();
            alphas[f].resize(n, 0); densities[f].resize(n, 0); predicted_densities[f].resize(n, 0);
            divergences[f].resize(n, 0); velocity_changes[f].resize(n, V3<R>());
            aii[f].resize(n, 0); pressures[f].resize(n, 0); next_pressures[f].resize(n, 0);
            dii[f].resize(n, V3<R>()); dij_pjl[f].resize(n, V3<R>());
            if (fluids[f].num_deleted != 0) {  // dfsph :550-560 / iisph :503-536 (each solver filters the buffers it owns)
                const Fluid<R>& fl = fluids[f];
                fl.filter(alphas[f]); fl.filter(densities[f]); fl.filter(predicted_densities[f]); fl.filter(divergences[f]);
                fl.filter(velocity_changes[f]); fl.filter(aii[f]); fl.filter(dii[f]); fl.filter(dij_pjl[f]);
                fl.filter(pressures[f]); fl.filter(next_pressures[f]);
            }
        }
    }
    // LiquidWorld::remove_fluid (liquid_world.rs:171-173) = ContiguousArena::remove (contiguous_arena.rs:118-135): a
    // swap-remove of the OBJECT only.  The solver's per-fluid buffers are positional and stay where they are, so the
    // fluid that moves into the freed slot inherits the removed fluid's velocity_changes / pressures (resized to its own
    // particle count by the next init_with_fluids) — restated as it is.
    void remove_fluid(size_t slot) {
        if (slot + 1 != fluids.size()) std::swap(fluids[slot], fluids.back());
        fluids.pop_back();
    }

    // ------------------------------------------------------------------ contacts.rs:133-151
    void insert_fluids_to_grid() {  // contacts.rs:133-140, liquid_world.rs:91
        for (size_t f = 0; f < fluids.size(); ++f)
            for (size_t p = 0; p < fluids[f].n(); ++p)
                grid.insert<R>(fluids[f].positions[p], h, Entry{(uint32_t)f, (uint32_t)p, false});
    }

    // ------------------------------------------------------------------ integrations/rapier/fluids_pipeline.rs:193-259
    // The DynamicContactSampling arm of ColliderCouplingManager::update_boundaries, run where the reference runs it: after
    // the fluids were inserted into the grid and before the boundaries are (liquid_world.rs:94-106).  The grid is NOT
    // rebuilt afterwards, so a fluid particle pushed out of the shape stays registered in the cell of its old position for
    // this substep's contact search — restated as it is.
    // parry3d 0.18 is an un-vendored dependency; restated from its published source:
    //   Ball::compute_aabb(pos)   = [t - r, t + r]                      (bounding_volume/aabb_ball.rs)
    //   Cuboid::compute_aabb(pos) = t -+ |R| * half_extents, R = UnitQuaternion::to_rotation_matrix (aabb_cuboid.rs,
    //                               nalgebra geometry/quaternion.rs), Aabb::loosened(m) = [mins - m, maxs + m]
    //   Aabb::contains_local_point: mins <= p <= maxs on every axis
    //   PointQuery::project_point_and_get_feature(m, pt) = project_local(m^-1 * pt) transformed back by m, with solid = false:
    //     Ball (query/point/point_ball.rs): inside = |p|^2 <= r^2; proj = p * (r / |p|)
    //     Cuboid = Aabb[-he, he] (query/point/point_aabb.rs do_project_local_point): shift = sup(mins - p, 0) - sup(p - maxs, 0);
    //       outside iff shift != 0 -> p + shift; inside -> the nearest face (largest of mins - p, p - maxs over the axes)
    void update_boundaries_dynamic() {
        const R prediction = h * (R)0.5;
        const R margin = particle_radius * (R)0.1;
        const R amount = h + prediction;
        for (size_t b = 0; b < boundaries.size(); ++b) {
            Boundary<R>& bd = boundaries[b];
            if (!bd.dyn_kind) continue;
            bd.positions.clear(); bd.velocities.clear(); bd.volumes.clear(); bd.dyn_source.clear();
            const V3<R> t = bd.pose_t, qv = bd.pose_qv;
            const R qw = bd.pose_qw;
            V3<R> ext;
            if (bd.dyn_kind == 1) {
                ext = V3<R>(bd.dyn_p[0], bd.dyn_p[0], bd.dyn_p[0]);
            } else if (bd.dyn_kind == 3) {
                // Capsule::aabb(pos) (shape/capsule.rs) = transform_by(pos).local_aabb(): the posed segment's ends A, B;
                // mins = inf(A, B) - radius, maxs = sup(A, B) + radius.  Capsule::new_y: a = (0, -hh, 0), b = (0, hh, 0).
                const V3<R> lb((R)0, bd.dyn_p[0], (R)0);
                const V3<R> tb = qv.cross(lb) * (R)2;
                const V3<R> u = tb * qw + qv.cross(tb) + lb;  // q * b ; q * a = -(q * b)
                ext = V3<R>(std::fabs(u.x) + bd.dyn_p[1], std::fabs(u.y) + bd.dyn_p[1], std::fabs(u.z) + bd.dyn_p[1]);
            } else {
                const R i = qv.x, j = qv.y, k = qv.z, w = qw;
                const R ww = w * w, ii = i * i, jj = j * j, kk = k * k;
                const R ij = i * j * (R)2, wk = w * k * (R)2, wj = w * j * (R)2, ik = i * k * (R)2, jk = j * k * (R)2, wi = w * i * (R)2;
                const R m[3][3] = {{ww + ii - jj - kk, ij - wk, wj + ik}, {wk + ij, ww - ii + jj - kk, jk - wi}, {ik - wj, wi + jk, ww - ii - jj + kk}};
                // cuboid: its half extents; cylinder (shape/cylinder.rs local_aabb): half extents (radius, half_height, radius)
                const R he[3] = {bd.dyn_kind == 4 ? bd.dyn_p[1] : bd.dyn_p[0], bd.dyn_kind == 4 ? bd.dyn_p[0] : bd.dyn_p[1],
                                 bd.dyn_kind == 4 ? bd.dyn_p[1] : bd.dyn_p[2]};
                R e[3];
                for (int a = 0; a < 3; ++a) e[a] = (std::fabs(m[a][0]) * he[0] + std::fabs(m[a][1]) * he[1]) + std::fabs(m[a][2]) * he[2];
                ext = V3<R>(e[0], e[1], e[2]);
            }
            const V3<R> lo = (t - ext) - V3<R>(amount, amount, amount), hi = (t + ext) + V3<R>(amount, amount, amount);
            const Cell start{HGrid::quantify<R>(lo.x, h), HGrid::quantify<R>(lo.y, h), HGrid::quantify<R>(lo.z, h)};
            const Cell end{HGrid::quantify<R>(hi.x, h), HGrid::quantify<R>(hi.y, h), HGrid::quantify<R>(hi.z, h)};
            for (size_t ci = 0; ci < grid.used; ++ci) {  // cells_intersecting_aabb (hgrid.rs:122-133): the existing cells of the range
                const Cell& c = grid.keys[ci];
                if (c.x < start.x || c.x > end.x || c.y < start.y || c.y > end.y || c.z < start.z || c.z > end.z) continue;
                for (const Entry& e : grid.cells[ci]) {
                    if (e.is_boundary) continue;  // "Not yet implemented." (:252-254)
                    Fluid<R>& fl = fluids[e.model];
                    const V3<R> pp = fl.positions[e.particle] + fl.velocities[e.particle] * dt;  // :206-207, dt of the last substep
                    if (pp.x < lo.x || pp.x > hi.x || pp.y < lo.y || pp.y > hi.y || pp.z < lo.z || pp.z > hi.z) continue;
                    // m^-1 * pt = q^-1 * (pt - t)
                    const V3<R> d = pp - t, nq = -qv;
                    const V3<R> t1 = nq.cross(d) * (R)2;
                    const V3<R> lp = t1 * qw + nq.cross(t1) + d;
                    V3<R> lproj;
                    bool inside;
                    if (bd.dyn_kind == 1) {
                        const R r = bd.dyn_p[0], d2 = lp.norm_squared();
                        inside = d2 <= r * r;
                        lproj = lp * (r / std::sqrt(d2));
                    } else if (bd.dyn_kind == 3) {
                        // Capsule (query/point/point_capsule.rs, solid = false) over Segment (point_segment.rs):
                        //   ab = b - a, ap = pt - a; ab.ap <= 0 -> a; >= |ab|^2 -> b; else a + ab * (ab.ap / |ab|^2)
                        //   dproj = pt - proj; Some((dir, dist)) = try_new_and_get(dproj, eps): inside = dist <= r, proj + dir * r
                        //   None (the point is on the segment): proj + basis * r with basis = orthonormal_basis(ab / |ab|)[0] = (1, 0, 0), inside
                        const R hh = bd.dyn_p[0], r = bd.dyn_p[1];
                        const V3<R> a((R)0, -hh, (R)0), ab((R)0, hh - (-hh), (R)0);
                        const V3<R> ap = lp - a;
                        const R ab_ap = ab.dot(ap), sqnab = ab.norm_squared();
                        V3<R> sp;
                        if (ab_ap <= (R)0) sp = a;
                        else if (ab_ap >= sqnab) sp = V3<R>((R)0, hh, (R)0);
                        else sp = a + ab * (ab_ap / sqnab);
                        const V3<R> dproj = lp - sp;
                        V3<R> dir;
                        R dist;
                        if (try_new_and_get(dproj, Eps<R>::v, dir, dist)) {
                            inside = dist <= r;
                            lproj = sp + dir * r;
                        } else {
                            inside = true;
                            lproj = sp + V3<R>((R)1, (R)0, (R)0) * r;
                        }
                    } else if (bd.dyn_kind == 4) {
                        // Cylinder (query/point/point_cylinder.rs, solid = false), axis = local y
                        const R hh = bd.dyn_p[0], r = bd.dyn_p[1];
                        R dx = lp.x, dz = lp.z;
                        const R planar = std::sqrt(dx * dx + dz * dz);  // normalize_mut returns the norm and divides by it
                        dx = dx / planar; dz = dz / planar;
                        if (planar <= Eps<R>::v) { dx = (R)1; dz = (R)0; }
                        const R px = dx * r, pz = dz * r;
                        if (lp.y >= -hh && lp.y <= hh && planar <= r) {
                            inside = true;
                            const R top = hh - lp.y, bottom = lp.y - (-hh), side = r - planar;
                            if (top < bottom && top < side) lproj = V3<R>(lp.x, hh, lp.z);
                            else if (bottom < top && bottom < side) lproj = V3<R>(lp.x, -hh, lp.z);
                            else lproj = V3<R>(px, lp.y, pz);
                        } else {
                            inside = false;
                            if (lp.y > hh) lproj = planar <= r ? V3<R>(lp.x, hh, lp.z) : V3<R>(px, hh, pz);
                            else if (lp.y < -hh) lproj = planar <= r ? V3<R>(lp.x, -hh, lp.z) : V3<R>(px, -hh, pz);
                            else lproj = V3<R>(px, lp.y, pz);
                        }
                    } else {
                        const R he[3] = {bd.dyn_p[0], bd.dyn_p[1], bd.dyn_p[2]}, p3[3] = {lp.x, lp.y, lp.z};
                        R mins_pt[3], pt_maxs[3], shift[3];
                        inside = true;
                        for (int a = 0; a < 3; ++a) {
                            mins_pt[a] = -he[a] - p3[a];
                            pt_maxs[a] = p3[a] - he[a];
                            shift[a] = std::max(mins_pt[a], (R)0) - std::max(pt_maxs[a], (R)0);
                            if (shift[a] != (R)0) inside = false;
                        }
                        if (inside) {
                            R best = -std::numeric_limits<R>::max();
                            bool is_mins = false;
                            int best_id = 0;
                            for (int a = 0; a < 3; ++a) {
                                if (mins_pt[a] < pt_maxs[a]) {
                                    if (pt_maxs[a] > best) { best_id = a; is_mins = false; best = pt_maxs[a]; }
                                } else if (mins_pt[a] > best) { best_id = a; is_mins = true; best = mins_pt[a]; }
                            }
                            shift[0] = shift[1] = shift[2] = 0;
                            shift[best_id] = is_mins ? best : -best;
                        }
                        lproj = V3<R>(p3[0] + shift[0], p3[1] + shift[1], p3[2] + shift[2]);
                    }
                    const V3<R> t2 = qv.cross(lproj) * (R)2;
                    const V3<R> proj = (t2 * qw + qv.cross(t2) + lproj) + t;
                    const V3<R> dpt = pp - proj;
                    V3<R> normal;
                    R depth;
                    if (try_new_and_get(dpt, Eps<R>::v, normal, depth)) {
                        if (inside) {
                            fl.positions[e.particle] -= normal * (depth + margin);
                            const R vel_err = normal.dot(fl.velocities[e.particle]);
                            if (vel_err > (R)0) fl.velocities[e.particle] -= normal * vel_err;
                        } else if (depth > h + prediction) {
                            continue;
                        }
                    }
                    bd.velocities.push_back(bd.pose_has_body ? bd.pose_linvel + bd.pose_angvel.cross(proj - bd.pose_com) : V3<R>());
                    bd.positions.push_back(proj);
                    bd.volumes.push_back((R)0);
                    bd.dyn_source.emplace_back(e.model, e.particle);
                }
            }
            if (bd.has_forces) bd.forces.assign(bd.n(), V3<R>());  // clear_forces(true) :262
        }
    }

    void insert_boundaries_to_grid() {  // contacts.rs:142-151, liquid_world.rs:106
        for (size_t b = 0; b < boundaries.size(); ++b)
            for (size_t p = 0; p < boundaries[b].n(); ++p)
                grid.insert<R>(boundaries[b].positions[p], h, Entry{(uint32_t)b, (uint32_t)p, true});
    }

    // ------------------------------------------------------------------ contacts.rs:254-400
    void contacts_for_pair_of_cells(const Cell& curr_cell, const std::vector<Entry>& curr,
                                    const Cell& nbr_cell, const std::vector<Entry>& nbr) {
        const bool thr = threaded();
        const bool same_cell = curr_cell == nbr_cell;
        const R h2 = h * h;
        for (const Entry& ei : curr) {
            if (ei.is_boundary) {
                for (const Entry& ej : nbr) {
                    if (ej.is_boundary) {
                        const Boundary<R>& bi = boundaries[ei.model];
                        const Boundary<R>& bj = boundaries[ej.model];
                        if (ei.model != ej.model && !bi.groups.test(bj.groups)) continue;
                        const V3<R>& pi = bi.positions[ei.particle];
                        const V3<R>& pj = bj.positions[ej.particle];
                        if ((pi - pj).norm_squared() <= h2) {
                            Contact<R> c{ei.particle, ei.model, ej.particle, ej.model, 0, V3<R>()};
                            bb[ei.model].push(ei.particle, c, thr);
                            if (!same_cell) bb[ej.model].push(ej.particle, c.flip(), thr);
                        }
                    } else {
                        if (same_cell) continue;  // handled when particle_i is the fluid particle
                        const Boundary<R>& bi = boundaries[ei.model];
                        const Fluid<R>& fj = fluids[ej.model];
                        if (!bi.groups.test(fj.groups)) continue;
                        const V3<R>& pi = bi.positions[ei.particle];
                        const V3<R>& pj = fj.positions[ej.particle];
                        if ((pi - pj).norm_squared() <= h2) {
                            Contact<R> c{ej.particle, ej.model, ei.particle, ei.model, 0, V3<R>()};
                            fb[ej.model].push(ej.particle, c, thr);
                        }
                    }
                }
            } else {
                for (const Entry& ej : nbr) {
                    const Fluid<R>& fi = fluids[ei.model];
                    const V3<R> pi = fi.positions[ei.particle];
                    V3<R> pj;
                    if (ej.is_boundary) {
                        const Boundary<R>& bj = boundaries[ej.model];
                        if (!fi.groups.test(bj.groups)) continue;
                        pj = bj.positions[ej.particle];
                    } else {
                        if (ei.model != ej.model) {
                            if (!fi.groups.test(fluids[ej.model].groups)) continue;
                        }
                        pj = fluids[ej.model].positions[ej.particle];
                    }
                    if ((pi - pj).norm_squared() <= h2) {
                        Contact<R> c{ei.particle, ei.model, ej.particle, ej.model, 0, V3<R>()};
                        if (ej.is_boundary) {
                            fb[ei.model].push(ei.particle, c, thr);
                        } else {
                            ff[ei.model].push(ei.particle, c, thr);
                            if (!same_cell) ff[ej.model].push(ej.particle, c.flip(), thr);
                        }
                    }
                }
            }
        }
    }

    // ------------------------------------------------------------------ contacts.rs:154-252
    void compute_contacts() {
        ff.resize(fluids.size()); fb.resize(fluids.size()); bb.resize(boundaries.size());
        for (size_t f = 0; f < fluids.size(); ++f) { ff[f].reset(fluids[f].n()); fb[f].reset(fluids[f].n()); }
        for (size_t b = 0; b < boundaries.size(); ++b) bb[b].reset(boundaries[b].n());

        static const int nb[14][3] = {{0, 0, 0},  {0, 0, 1},  {0, 1, -1}, {0, 1, 0},  {0, 1, 1},
                                      {1, -1, -1}, {1, -1, 0}, {1, -1, 1}, {1, 0, -1}, {1, 0, 0},
                                      {1, 0, 1},  {1, 1, -1}, {1, 1, 0},  {1, 1, 1}};
        std::vector<uint32_t> order(grid.used);
        for (size_t i = 0; i < grid.used; ++i) order[i] = (uint32_t)i;
        if (shuffle_seed) {
            std::mt19937_64 rng(shuffle_seed);
            std::shuffle(order.begin(), order.end(), rng);
        }
        const long ncells = (long)grid.used;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads) if (nthreads > 1)
        for (long ci = 0; ci < ncells; ++ci) {
            const Cell& cc = grid.keys[order[ci]];
            const std::vector<Entry>& cp = grid.cells[order[ci]];
            for (int s = 0; s < 14; ++s) {
                Cell nc{cc.x + nb[s][0], cc.y + nb[s][1], cc.z + nb[s][2]};
                const std::vector<Entry>* np = grid.cell(nc);
                if (np) contacts_for_pair_of_cells(cc, cp, nc, *np);
            }
        }
    }

    uint64_t ncontacts() const {  // contact_manager.rs:31-47
        uint64_t s = 0;
        for (auto& c : ff) s += c.len();
        for (auto& c : fb) s += c.len();
        for (auto& c : bb) s += c.len();
        return s;
    }

    // ------------------------------------------------------------------ helper.rs:9-65
    void evaluate_kernels() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const long n = (long)fluids[f].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                for (auto& c : ff[f].contacts[i]) {
                    const V3<R>& pi = fluids[c.i_model].positions[c.i];
                    const V3<R>& pj = fluids[c.j_model].positions[c.j];
                    c.weight = kernel_apply<R>(kernel_density, pi - pj, h);
                    c.gradient = kernel_apply_diff<R>(kernel_gradient, pi - pj, h);
                }
                for (auto& c : fb[f].contacts[i]) {
                    const V3<R>& pi = fluids[c.i_model].positions[c.i];
                    const V3<R>& pj = boundaries[c.j_model].positions[c.j];
                    c.weight = kernel_apply<R>(kernel_density, pi - pj, h);
                    c.gradient = kernel_apply_diff<R>(kernel_gradient, pi - pj, h);
                }
            }
        }
        for (size_t b = 0; b < boundaries.size(); ++b) {
            const long n = (long)boundaries[b].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                for (auto& c : bb[b].contacts[i]) {
                    const V3<R>& pi = boundaries[c.i_model].positions[c.i];
                    const V3<R>& pj = boundaries[c.j_model].positions[c.j];
                    c.weight = kernel_apply<R>(kernel_density, pi - pj, h);
                    c.gradient = kernel_apply_diff<R>(kernel_gradient, pi - pj, h);
                }
            }
        }
    }

    // ------------------------------------------------------------------ dfsph_solver.rs:72-96 / iisph :66-90
    void compute_boundary_volumes() {
        for (size_t b = 0; b < boundaries.size(); ++b) {
            const long n = (long)boundaries[b].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                R denominator = 0;
                for (auto& c : bb[b].contacts[i]) denominator += c.weight;
                assert(denominator != 0);
                boundaries[b].volumes[i] = (R)1 / denominator;
            }
        }
    }

    // ------------------------------------------------------------------ dfsph_solver.rs:628-665 / iisph :598-640
    void compute_densities() {
        compute_boundary_volumes();
        for (size_t f = 0; f < fluids.size(); ++f) {
            const long n = (long)fluids[f].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                R density = 0;
                for (auto& c : ff[f].contacts[i]) density += fluids[c.j_model].particle_mass(c.j) * c.weight;
                for (auto& c : fb[f].contacts[i])
                    density += boundaries[c.j_model].volumes[c.j] * fluids[c.i_model].density0 * c.weight;
                assert(density != 0);
                densities[f][i] = density;
            }
        }
    }

    void apply_force(Boundary<R>& b, size_t i, const V3<R>& f) {  // boundary.rs:62-67
        if (!b.has_forces) return;
        if (threaded()) {
#pragma omp critical(boundary_force)
            b.forces[i] += f;
        } else b.forces[i] += f;
    }

    // ================================================================== DFSPH
    // dfsph_solver.rs:165-216
    void compute_alphas() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                V3<R> grad_sum; R squared_grad_sum = 0;
                for (auto& c : ff[f].contacts[i]) {
                    V3<R> grad_i = c.gradient * fluids[c.j_model].particle_mass(c.j);
                    squared_grad_sum += grad_i.norm_squared();
                    grad_sum += grad_i;
                }
                for (auto& c : fb[f].contacts[i]) {
                    V3<R> grad_i = c.gradient * boundaries[c.j_model].volumes[c.j] * fluid_i.density0;
                    squared_grad_sum += grad_i.norm_squared();
                    grad_sum += grad_i;
                }
                R denominator = squared_grad_sum + grad_sum.norm_squared();
                alphas[f][i] = (denominator <= (R)1.0e-5) ? (R)0 : (R)1 / denominator;
            }
        }
    }

    // dfsph_solver.rs:279-356
    R compute_divergences() {
        R max_error = 0;
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
            std::vector<R> errs(n);
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                R div = 0;
                if (ff[f].contacts[i].size() + fb[f].contacts[i].size() < min_neighbors_for_divergence_solve) {
                    divergences[f][i] = 0; errs[i] = 0; continue;
                }
                for (auto& c : ff[f].contacts[i]) {
                    const Fluid<R>& fluid_j = fluids[c.j_model];
                    V3<R> v_i = fluid_i.velocities[c.i] + velocity_changes[c.i_model][c.i];
                    V3<R> v_j = fluid_j.velocities[c.j] + velocity_changes[c.j_model][c.j];
                    V3<R> dvel = v_i - v_j;
                    div += dvel.dot(c.gradient) * fluid_j.particle_mass(c.j);
                }
                for (auto& c : fb[f].contacts[i]) {
                    V3<R> v_i = fluid_i.velocities[c.i] + velocity_changes[c.i_model][c.i];
                    div += v_i.dot(c.gradient) * boundaries[c.j_model].volumes[c.j] * fluid_i.density0;
                }
                div = std::max(div, (R)0);
                divergences[f][i] = div;
                errs[i] = div / fluid_i.density0;
            }
            R err = 0;  // par_reduce_sum! (serial fold order)
            for (long i = 0; i < n; ++i) err = err + errs[i];
            if (n != 0) max_error = std::max(max_error, err / (R)(double)n);
        }
        return max_error;
    }

    // dfsph_solver.rs:358-409
    void compute_velocity_changes_for_divergence() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid1 = fluids[f];
            const long n = (long)fluid1.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                V3<R>& velocity_change = velocity_changes[f][i];
                R ki = divergences[f][i] * alphas[f][i];
                for (auto& c : ff[f].contacts[i]) {
                    const Fluid<R>& fluid2 = fluids[c.j_model];
                    R kj = divergences[c.j_model][c.j] * alphas[c.j_model][c.j];
                    R coeff = -(ki + kj) * fluid2.particle_mass(c.j);
                    velocity_change += c.gradient * coeff;
                }
                for (auto& c : fb[f].contacts[i]) {
                    Boundary<R>& boundary2 = boundaries[c.j_model];
                    R coeff = -ki * boundary2.volumes[c.j] * fluid1.density0;
                    V3<R> delta = c.gradient * coeff;
                    velocity_change += delta;
                    R particle_mass = fluid1.particle_mass(c.i);
                    apply_force(boundary2, c.j, delta * (-inv_dt * particle_mass));
                }
            }
        }
    }

    // dfsph_solver.rs:98-162
    R compute_predicted_densities_dfsph() {
        R max_error = 0;
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
            std::vector<R> errs(n);
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                R delta = 0;
                for (auto& c : ff[f].contacts[i]) {
                    const Fluid<R>& fluid_j = fluids[c.j_model];
                    V3<R> vi = fluid_i.velocities[c.i] + velocity_changes[c.i_model][c.i];
                    V3<R> vj = fluid_j.velocities[c.j] + velocity_changes[c.j_model][c.j];
                    delta += fluid_j.particle_mass(c.j) * (vi - vj).dot(c.gradient);
                }
                for (auto& c : fb[f].contacts[i]) {
                    V3<R> vi = fluid_i.velocities[c.i] + velocity_changes[c.i_model][c.i];
                    V3<R> vj = boundaries[c.j_model].velocities[c.j];
                    delta += boundaries[c.j_model].volumes[c.j] * fluid_i.density0 * (vi - vj).dot(c.gradient);
                }
                R pd = densities[f][i] + delta * dt;
                assert(pd != 0);
                predicted_densities[f][i] = pd;
                errs[i] = (pd < fluid_i.density0) ? (R)0 : pd / fluid_i.density0 - (R)1;
            }
            R err = 0;
            for (long i = 0; i < n; ++i) err = err + errs[i];
            if (n != 0) max_error = std::max(max_error, err / (R)(double)n);
        }
        return max_error;
    }

    // dfsph_solver.rs:218-277
    void compute_velocity_changes_dfsph() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid1 = fluids[f];
            const long n = (long)fluid1.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                V3<R>& velocity_change = velocity_changes[f][i];
                R ki = (predicted_densities[f][i] - fluid1.density0) * alphas[f][i];
                for (auto& c : ff[f].contacts[i]) {
                    const Fluid<R>& fluid2 = fluids[c.j_model];
                    R kj = (predicted_densities[c.j_model][c.j] - fluid2.density0) * alphas[c.j_model][c.j];
                    R kij = std::max(ki, (R)0) + std::max(kj, (R)0);
                    if (kij > (R)0) {
                        R coeff = kij * fluid2.particle_mass(c.j);
                        velocity_change -= c.gradient * (coeff * inv_dt);
                    }
                }
                if (ki > (R)0) {
                    for (auto& c : fb[f].contacts[i]) {
                        R coeff = ki * boundaries[c.j_model].volumes[c.j] * fluid1.density0;
                        V3<R> delta = c.gradient * (coeff * inv_dt);
                        velocity_change -= delta;
                        R particle_mass = fluid1.particle_mass(c.i);
                        apply_force(boundaries[c.j_model], c.j, delta * (inv_dt * particle_mass));
                    }
                }
            }
        }
    }

    // ================================================================== non-pressure forces
    // xsph_viscosity.rs:31-95
    void solve_xsph(size_t f, Force<R>& force) {
        Fluid<R>& fluid = fluids[f];
        const R fc = force.p[0], bc = force.p[1];
        const R density0 = fluid.density0;
        const std::vector<R>& dens = densities[f];
        const long n = (long)fluid.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {
            V3<R> added_fluid_vel, added_boundary_vel;
            V3<R> vi = fluid.velocities[i];
            if (fc != (R)0) {
                for (auto& c : ff[f].contacts[i]) {
                    if (c.i_model == c.j_model) {
                        added_fluid_vel += (fluid.velocities[c.j] - vi) *
                                           (fc * c.weight * fluid.volumes[c.j] * density0 / dens[c.j]);
                    }
                }
            }
            if (bc != (R)0) {
                for (auto& c : fb[f].contacts[i]) {
                    V3<R> delta = (boundaries[c.j_model].velocities[c.j] - vi) *
                                  (bc * c.weight * boundaries[c.j_model].volumes[c.j] * density0 / dens[c.i]);
                    added_boundary_vel += delta;
                    R mi = fluid.volumes[c.i] * density0;
                    apply_force(boundaries[c.j_model], c.j, delta * (-mi * inv_dt));
                }
            }
            fluid.accelerations[i] += added_fluid_vel * inv_dt + added_boundary_vel * inv_dt;
        }
    }

    // artificial_viscosity.rs:41-124
    void solve_artificial(size_t f, Force<R>& force) {
        Fluid<R>& fluid = fluids[f];
        const R fc = force.p[0], bc = force.p[1], alpha = force.p[2], beta = force.p[3], speed_of_sound = force.p[4];
        const R density0 = fluid.density0;
        const std::vector<R>& dens = densities[f];
        const R kernel_radius = h;
        const long n = (long)fluid.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {
            V3<R> fluid_acc, boundary_acc;
            if (fc != (R)0) {
                for (auto& c : ff[f].contacts[i]) {
                    if (c.i_model == c.j_model) {
                        V3<R> r_ij = fluid.positions[c.i] - fluid.positions[c.j];
                        V3<R> v_ij = fluid.velocities[c.i] - fluid.velocities[c.j];
                        R vr = r_ij.dot(v_ij);
                        if (vr < (R)0) {
                            R density_average = (dens[c.i] + dens[c.j]) * (R)0.5;
                            R eta2 = kernel_radius * kernel_radius * (R)0.01;
                            R mu_ij = kernel_radius * vr / (r_ij.norm_squared() + eta2);
                            fluid_acc += c.gradient * (fc * (speed_of_sound * alpha * mu_ij - beta * mu_ij * mu_ij) *
                                                       (fluid.volumes[c.j] * density0 / density_average));
                        }
                    }
                }
            }
            if (bc != (R)0) {
                for (auto& c : fb[f].contacts[i]) {
                    V3<R> r_ij = fluid.positions[c.i] - boundaries[c.j_model].positions[c.j];
                    V3<R> v_ij = fluid.velocities[c.i] - boundaries[c.j_model].velocities[c.j];
                    R vr = r_ij.dot(v_ij);
                    if (vr < (R)0) {
                        R density_average = dens[c.i];
                        R eta2 = kernel_radius * kernel_radius * (R)0.01;
                        R mu_ij = kernel_radius * vr / (r_ij.norm_squared() + eta2);
                        boundary_acc += c.gradient * (bc * (speed_of_sound * alpha * mu_ij - beta * mu_ij * mu_ij) *
                                                      (boundaries[c.j_model].volumes[c.j] * density0 / density_average));
                        R mi = fluid.volumes[c.i] * density0;
                        // NOTE: the reference applies the *running sum* here (artificial_viscosity.rs:117).
                        apply_force(boundaries[c.j_model], c.j, boundary_acc * -mi);
                    }
                }
            }
            fluid.accelerations[i] += fluid_acc + boundary_acc;
        }
    }

    // akinci2013_surface_tension.rs:71-88
    static R cohesion_kernel(R r, R hh) {
        R normalizer = (R)32.0 / (CubicSpline<R>::PI * powi<R>(hh, 9));
        R coeff;
        if (r <= hh / (R)2) coeff = (R)2 * powi<R>(hh - r, 3) * powi<R>(r, 3) - powi<R>(hh, 6) / (R)64.0;
        else if (r <= hh) coeff = powi<R>(hh - r, 3) * powi<R>(r, 3);
        else coeff = 0;
        return normalizer * coeff;
    }
    // akinci2013_surface_tension.rs:90-111
    static R adhesion_kernel(R r, R hh) {
        if (r > hh / (R)2 && r <= hh) {
            R normalizer = (R)0.007 / std::pow(hh, (R)3.25);
            R coeff = std::pow(std::max((R)-4 * r * r / hh + (R)6 * r - (R)2 * hh, (R)0), (R)0.25);
            return normalizer * coeff;
        }
        return 0;
    }

    // akinci2013_surface_tension.rs:43-68, 114-192
    void solve_akinci(size_t f, Force<R>& force) {
        Fluid<R>& fluid = fluids[f];
        const R tc = force.p[0], ac = force.p[1];
        const R density0 = fluid.density0;
        const std::vector<R>& dens = densities[f];
        const R kernel_radius = h;
        const long n = (long)fluid.n();
        if (force.normals.size() != (size_t)n) force.normals.resize(n, V3<R>());
        std::vector<V3<R>>& normals = force.normals;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {
            V3<R> normal;
            for (auto& c : ff[f].contacts[i])
                if (c.i_model == c.j_model) normal += c.gradient * (fluid.particle_mass(c.j) / dens[c.j]);
            normals[i] = normal * kernel_radius;
        }
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {
            V3<R>& acceleration_i = fluid.accelerations[i];
            if (tc != (R)0) {
                for (auto& c : ff[f].contacts[i]) {
                    if (c.i_model == c.j_model) {
                        V3<R> dpos = fluid.positions[c.i] - fluid.positions[c.j];
                        V3<R> dir; R dist; V3<R> cohesion_vec;
                        if (try_new_and_get<R>(dpos, Eps<R>::v, dir, dist)) cohesion_vec = dir * cohesion_kernel(dist, kernel_radius);
                        V3<R> cohesion_acc = cohesion_vec * (-tc * fluid.volumes[c.j] * density0);
                        V3<R> curvature_acc = (normals[c.i] - normals[c.j]) * -tc;
                        R kij = (R)2 * density0 / (dens[c.i] + dens[c.j]);
                        acceleration_i += (curvature_acc + cohesion_acc) * kij;
                    }
                }
            }
            if (ac != (R)0) {
                for (auto& c : fb[f].contacts[i]) {
                    V3<R> dpos = fluid.positions[c.i] - boundaries[c.j_model].positions[c.j];
                    V3<R> dir; R dist; V3<R> adhesion_vec;
                    if (try_new_and_get<R>(dpos, Eps<R>::v, dir, dist)) adhesion_vec = dir * adhesion_kernel(dist, kernel_radius);
                    R mi = fluid.volumes[c.i] * density0;
                    R mj = boundaries[c.j_model].volumes[c.j] * density0;
                    V3<R> adhesion_acc = adhesion_vec * (ac * mj);
                    acceleration_i -= adhesion_acc;
                    apply_force(boundaries[c.j_model], c.j, adhesion_acc * mi);
                }
            }
        }
    }

    // surface_tension/he2014_surface_tension.rs:40-181 (compute_colors, compute_gradc, solve)
    void solve_he2014(size_t f, Force<R>& force) {
        Fluid<R>& fluid = fluids[f];
        const R tc = force.p[0], bc = force.p[1];
        const R density0 = fluid.density0;
        const std::vector<R>& dens = densities[f];
        const long n = (long)fluid.n();
        if (force.gradcs.size() != (size_t)n) { force.gradcs.assign(n, (R)0); force.colors.assign(n, (R)0); }  // init :31-38
        std::vector<R>& colors = force.colors;
        std::vector<R>& gradcs = force.gradcs;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {  // compute_colors :40-76
            R color = 0;
            for (auto& c : ff[f].contacts[i])
                if (c.i_model == c.j_model) color += c.weight * fluid.particle_mass(c.j) / dens[c.j];
            for (auto& c : fb[f].contacts[i]) color += c.weight * boundaries[c.j_model].volumes[c.j];
            colors[i] = color;
        }
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {  // compute_gradc :78-106
            V3<R> gradc;
            for (auto& c : ff[f].contacts[i])
                if (c.i_model == c.j_model) gradc += c.gradient * colors[c.j] * fluid.particle_mass(c.j) / dens[c.j];
            gradcs[i] = (gradc / colors[i]).norm_squared();
        }
        const R _2 = 2;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {  // solve :132-178
            V3<R>& acceleration_i = fluid.accelerations[i];
            const R mi = fluid.volumes[i] * density0;
            if (tc != (R)0) {
                for (auto& c : ff[f].contacts[i]) {
                    if (c.i_model == c.j_model) {
                        const R mj = fluid.volumes[c.j] * density0;
                        const R gradsum = gradcs[c.i] + gradcs[c.j];
                        V3<R> fv = c.gradient * (mi / dens[c.i] * mj / dens[c.j] * gradsum / _2);
                        acceleration_i += fv * (tc / (_2 * mi));
                    }
                }
            }
            if (bc != (R)0) {
                for (auto& c : fb[f].contacts[i]) {
                    const R mj = boundaries[c.j_model].volumes[c.j] * density0;
                    const R gradsum = gradcs[c.i];
                    V3<R> fv = c.gradient * (mi / dens[c.i] * mj / density0 * gradsum * bc * (R)0.25);
                    acceleration_i += fv / mi;
                    apply_force(boundaries[c.j_model], c.j, fv * (R)-1);
                }
            }
        }
    }

    // surface_tension/wcsph_surface_tension.rs:32-87.  NOTE: the reference's boundary loop (:66-83) iterates the
    // *fluid-fluid* contacts and indexes `boundaries[c.j_model].positions[c.j]` with them — it panics (index out of
    // bounds) unless a boundary with that index and enough particles happens to exist.  Restated as written, with the
    // out-of-bounds case reported through `reference_would_panic` instead of undefined behaviour.
    bool reference_would_panic = false;
    void solve_wcsph_tension(size_t f, Force<R>& force) {
        Fluid<R>& fluid = fluids[f];
        const R tc = force.p[0], bc = force.p[1];
        const R density0 = fluid.density0;
        const long n = (long)fluid.n();
        bool oob = false;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1) reduction(|| : oob)
        for (long i = 0; i < n; ++i) {
            V3<R>& acceleration_i = fluid.accelerations[i];
            if (tc != (R)0) {
                for (auto& c : ff[f].contacts[i]) {
                    if (c.i_model == c.j_model) {
                        V3<R> dpos = fluid.positions[c.i] - fluid.positions[c.j];
                        V3<R> cohesion_acc = dpos * (-tc * c.weight * fluid.volumes[c.j] * density0 / (fluid.volumes[c.i] * density0));
                        acceleration_i += cohesion_acc;
                    }
                }
            }
            if (bc != (R)0) {
                for (auto& c : ff[f].contacts[i]) {
                    if (c.j_model >= boundaries.size() || c.j >= boundaries[c.j_model].positions.size()) { oob = true; continue; }
                    V3<R> dpos = fluid.positions[c.i] - boundaries[c.j_model].positions[c.j];
                    const R mi = fluid.volumes[c.i] * density0;
                    V3<R> cohesion_force = dpos * (bc * c.weight * boundaries[c.j_model].volumes[c.j] * density0);
                    acceleration_i -= cohesion_force / mi;
                    apply_force(boundaries[c.j_model], c.j, cohesion_force);
                }
            }
        }
        if (oob) reference_would_panic = true;
    }

    // ---- viscosity/dfsph_viscosity.rs (viscous DFSPH).  3D: strain rates are 6-vectors, betas 6x6 matrices.
    // compute_strain_rate (:38-58)
    static inline void strain_rate(const V3<R>& g, const V3<R>& v, R out[6]) {
        const R _2 = 2;
        out[0] = _2 * v.x * g.x; out[1] = _2 * v.y * g.y; out[2] = _2 * v.z * g.z;
        out[3] = v.x * g.y + v.y * g.x; out[4] = v.x * g.z + v.z * g.x; out[5] = v.y * g.z + v.z * g.y;
    }
    // compute_gradient_matrix (:60-83): 6x3, row-major
    static inline void gradient_matrix(const V3<R>& g, R m[6][3]) {
        const R _2 = 2;
        m[0][0] = g.x * _2; m[0][1] = 0; m[0][2] = 0;
        m[1][0] = 0; m[1][1] = g.y * _2; m[1][2] = 0;
        m[2][0] = 0; m[2][1] = 0; m[2][2] = g.z * _2;
        m[3][0] = g.y; m[3][1] = g.x; m[3][2] = 0;
        m[4][0] = g.z; m[4][1] = 0; m[4][2] = g.x;
        m[5][0] = 0; m[5][1] = g.z; m[5][2] = g.y;
    }
    // nalgebra 0.33 `Matrix6::lu()` (linalg/lu.rs: partial pivoting on the largest |.| of the column, multipliers = entry *
    // (1 / pivot), trailing update y = (-p_k) * l + y) followed by `determinant()` and `try_inverse()` (linalg/solve.rs:
    // column-oriented forward substitution with unit diagonal, then back substitution dividing by the diagonal).  nalgebra
    // is an un-vendored dependency: restated from its published algorithm, not checked against its source here.
    // Returns false when U has a zero on the diagonal (try_inverse -> None); det receives lu.determinant().
    static bool lu_inverse6(const R a_in[6][6], R inv[6][6], R& det) {
        R a[6][6];
        int perm[6];
        for (int i = 0; i < 6; ++i) { perm[i] = i; for (int j = 0; j < 6; ++j) a[i][j] = a_in[i][j]; }
        int nswaps = 0;
        for (int i = 0; i < 6; ++i) {
            int piv = i;
            R best = std::abs(a[i][i]);
            for (int r = i + 1; r < 6; ++r) { const R v = std::abs(a[r][i]); if (v > best) { best = v; piv = r; } }  // icamax: first maximum
            const R diag = a[piv][i];
            if (diag == (R)0) continue;
            if (piv != i) {
                for (int j = 0; j < 6; ++j) std::swap(a[i][j], a[piv][j]);
                std::swap(perm[i], perm[piv]);
                ++nswaps;
            }
            const R inv_diag = (R)1 / diag;
            for (int r = i + 1; r < 6; ++r) a[r][i] *= inv_diag;
            for (int k = i + 1; k < 6; ++k) {
                const R pk = a[i][k];
                for (int r = i + 1; r < 6; ++r) a[r][k] = (-pk) * a[r][i] + a[r][k];
            }
        }
        det = 1;
        for (int i = 0; i < 6; ++i) det *= a[i][i];
        if (nswaps & 1) det = -det;
        for (int c = 0; c < 6; ++c) {
            R b[6];
            for (int r = 0; r < 6; ++r) b[r] = (perm[r] == c) ? (R)1 : (R)0;  // P * e_c
            for (int i = 0; i < 5; ++i) {
                const R coeff = b[i];
                for (int r = i + 1; r < 6; ++r) b[r] = (-coeff) * a[r][i] + b[r];
            }
            for (int i = 5; i >= 0; --i) {
                const R d = a[i][i];
                if (d == (R)0) return false;
                const R coeff = b[i] / d;
                b[i] = coeff;
                for (int r = 0; r < i; ++r) b[r] = (-coeff) * a[r][i] + b[r];
            }
            for (int r = 0; r < 6; ++r) inv[r][c] = b[r];
        }
        return true;
    }
    // compute_betas (:130-194)
    void visc_compute_betas(size_t f, Force<R>& force) {
        Fluid<R>& fluid = fluids[f];
        const std::vector<R>& dens = densities[f];
        const long n = (long)fluid.n();
        const R _2 = 2;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {
            R grad_sum[6][3] = {}, sq[6][6] = {};
            for (auto& c : ff[f].contacts[i]) {
                if (c.i_model != c.j_model) continue;
                R mat[6][3];
                gradient_matrix(c.gradient, mat);
                const R s = fluid.volumes[c.j] * fluid.density0 / (_2 * dens[c.i]);  // particle_mass(j) / (2 rho_i)
                R gi[6][3];
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 3; ++b) gi[a][b] = mat[a][b] * s;
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) {
                        R v = 0;
                        for (int k = 0; k < 3; ++k) v += gi[a][k] * gi[b][k];
                        sq[a][b] += v / dens[c.i];
                    }
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 3; ++b) grad_sum[a][b] += gi[a][b];
            }
            R den[6][6];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    R v = 0;
                    for (int k = 0; k < 3; ++k) v += grad_sum[a][k] * grad_sum[b][k];
                    den[a][b] = sq[a][b] + v / dens[i];
                }
            // preconditioner (:163-175).  NOTE the reference loops over SPATIAL_DIM = 3 columns only, and scales the rows
            // of each of them component-wise by inv_diag (column_mut(i).component_mul_assign(&inv_diag)).
            R inv_diag[6];
            for (int a = 0; a < 6; ++a) { const R d = den[a][a]; inv_diag[a] = (std::abs(d) < (R)1.0e-6) ? (R)1 : (R)1 / d; }
            for (int col = 0; col < 3; ++col) for (int r = 0; r < 6; ++r) den[r][col] *= inv_diag[r];
            // (:177-193) the dim3 determinant()/try_inverse() block is overwritten by the lu() block that follows it
            R inv[6][6], det = 0;
            R* beta = &force.betas[(size_t)i * 36];
            const bool ok = lu_inverse6(den, inv, det);
            if (std::abs(det) < (R)1.0e-6 || !ok) { for (int k = 0; k < 36; ++k) beta[k] = 0; }
            else { for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) beta[a * 6 + b] = inv[a][b]; }
            for (int col = 0; col < 3; ++col) for (int r = 0; r < 6; ++r) beta[r * 6 + col] *= inv_diag[col];
        }
    }
    // compute_strain_rates (:196-247)
    R visc_compute_strain_rates(size_t f, Force<R>& force, bool compute_error) {
        Fluid<R>& fluid = fluids[f];
        const std::vector<R>& dens = densities[f];
        const long n = (long)fluid.n();
        const R coef = force.p[0], _2 = 2;
        const R tdt = dt;  // timestep.dt(): still the previous step's here (the dt lag)
        std::vector<R> errs((size_t)n, 0);
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {
            R rate[6] = {0, 0, 0, 0, 0, 0};
            for (auto& c : ff[f].contacts[i]) {
                if (c.i_model != c.j_model) continue;
                const V3<R> v_i = fluid.velocities[c.i] + fluid.accelerations[c.i] * tdt;
                const V3<R> v_j = fluid.velocities[c.j] + fluid.accelerations[c.j] * tdt;
                R r[6];
                strain_rate(c.gradient, v_j - v_i, r);
                const R s = fluid.volumes[c.j] * fluid.density0 / (_2 * dens[c.i]);
                for (int k = 0; k < 6; ++k) rate[k] += r[k] * s;
            }
            R* tgt = &force.strain_target[(size_t)i * 6];
            R* err = &force.strain_error[(size_t)i * 6];
            if (compute_error) {
                R l1 = 0;
                for (int k = 0; k < 6; ++k) { err[k] = rate[k] - tgt[k]; l1 += std::abs(err[k]); }
                errs[i] = l1 / (R)6;
            } else {
                for (int k = 0; k < 6; ++k) tgt[k] = rate[k] * ((R)1 - coef);
            }
        }
        R sum = 0;
        for (long i = 0; i < n; ++i) sum += errs[i];
        return n ? std::max((R)0, sum / (R)n) : (R)0;
    }
    // compute_accelerations (:249-287)
    void visc_compute_accelerations(size_t f, Force<R>& force) {
        Fluid<R>& fluid = fluids[f];
        const std::vector<R>& dens = densities[f];
        const long n = (long)fluid.n();
        const R density0 = fluid.density0, _2 = 2;
        std::vector<V3<R>> add((size_t)n);
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
        for (long i = 0; i < n; ++i) {
            auto u_of = [&](size_t k, R u[6]) {
                const R* b = &force.betas[k * 36];
                const R* e = &force.strain_error[k * 6];
                const R d2 = dens[k] * dens[k];
                for (int a = 0; a < 6; ++a) {
                    R v = 0;
                    for (int q = 0; q < 6; ++q) v += b[a * 6 + q] * e[q];
                    u[a] = v / d2;
                }
            };
            R ui[6];
            u_of((size_t)i, ui);
            V3<R> acc;
            for (auto& c : ff[f].contacts[i]) {
                if (c.i_model != c.j_model) continue;
                R uj[6], coeff[6];
                u_of(c.j, uj);
                const R s = fluid.volumes[c.j] * density0 / _2;
                for (int a = 0; a < 6; ++a) coeff[a] = (ui[a] + uj[a]) * s;
                const V3<R>& g = c.gradient;
                // gradient.tr_mul(&coeff): M^T coeff
                const V3<R> t((g.x * _2) * coeff[0] + g.y * coeff[3] + g.z * coeff[4],
                              (g.y * _2) * coeff[1] + g.x * coeff[3] + g.z * coeff[5],
                              (g.z * _2) * coeff[2] + g.x * coeff[4] + g.y * coeff[5]);
                acc += t * (fluid.volumes[c.i] * density0 * inv_dt);
            }
            add[i] = acc;
        }
        // accelerations feed the next strain-rate pass of every particle: apply after the whole pass (the reference
        // updates in place under rayon, i.e. a neighbour's acceleration may or may not be updated yet — order dependent;
        // the strain pass only starts after this one has finished, so applying at the end is one of the legal orders
        // only if nothing reads accelerations inside this pass — and nothing does)
        for (long i = 0; i < n; ++i) fluid.accelerations[i] += add[i];
    }
    // NonPressureForce::solve (:290-327)
    void solve_dfsph_viscosity(size_t f, Force<R>& force) {
        const size_t n = fluids[f].n();
        if (force.betas.size() != n * 36) { force.betas.assign(n * 36, 0); force.strain_target.assign(n * 6, 0); force.strain_error.assign(n * 6, 0); }
        visc_compute_betas(f, force);
        (void)visc_compute_strain_rates(f, force, false);
        const int min_it = (int)force.p[1], max_it = (int)force.p[2];
        const R max_err = force.p[3];
        int it = 0;
        R err = 0;
        for (int i = 0; i < max_it; ++i) {
            err = visc_compute_strain_rates(f, force, true);
            if (err <= max_err && i >= min_it) break;
            visc_compute_accelerations(f, force);
            it = i + 1;
        }
        force.last_visc_iters = it;
        force.last_visc_error = err;
    }

    // A user `NonPressureForce` (solver/nonpressure_force.rs:10-30, examples3d/custom_forces3.rs:67-90): the host callback
    // sees the fluid as `solve` does (positions, velocities, densities of this substep; the contact lists through the
    // contact accessors) and adds to `fluid.accelerations`.
    typedef void (*custom_cb_t)(void* user, int fluid, int force_index, uint64_t n, const double* pos, const double* vel,
                                const double* dens, double* acc);
    custom_cb_t custom_cb = nullptr;
    void* custom_user = nullptr;
    void solve_custom(size_t f, size_t index) {
        if (!custom_cb) return;
        Fluid<R>& fluid = fluids[f];
        const size_t n = fluid.n();
        std::vector<double> pos(3 * n), vel(3 * n), dens(n), acc(3 * n);
        for (size_t i = 0; i < n; ++i) {
            pos[3 * i] = fluid.positions[i].x; pos[3 * i + 1] = fluid.positions[i].y; pos[3 * i + 2] = fluid.positions[i].z;
            vel[3 * i] = fluid.velocities[i].x; vel[3 * i + 1] = fluid.velocities[i].y; vel[3 * i + 2] = fluid.velocities[i].z;
            acc[3 * i] = fluid.accelerations[i].x; acc[3 * i + 1] = fluid.accelerations[i].y; acc[3 * i + 2] = fluid.accelerations[i].z;
            dens[i] = densities[f][i];
        }
        custom_cb(custom_user, (int)f, (int)index, n, pos.data(), vel.data(), dens.data(), acc.data());
        for (size_t i = 0; i < n; ++i) fluid.accelerations[i] = V3<R>((R)acc[3 * i], (R)acc[3 * i + 1], (R)acc[3 * i + 2]);
    }

    // dfsph_solver.rs:565-604 / iisph_solver.rs:541-580
    void predict_advection(const V3<R>& gravity) {
        for (auto& fluid : fluids) {
            const long n = (long)fluid.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) fluid.accelerations[i] += gravity;
        }
        for (size_t f = 0; f < fluids.size(); ++f) {
            for (auto& force : fluids[f].forces) {
                switch (force.kind) {
                    case FORCE_XSPH: solve_xsph(f, force); break;
                    case FORCE_ARTIFICIAL: solve_artificial(f, force); break;
                    case FORCE_AKINCI2013: solve_akinci(f, force); break;
                    case FORCE_DFSPH_VISCOSITY: solve_dfsph_viscosity(f, force); break;
                    case FORCE_HE2014: solve_he2014(f, force); break;
                    case FORCE_WCSPH_TENSION: solve_wcsph_tension(f, force); break;
                    case FORCE_CUSTOM: solve_custom(f, (size_t)(&force - fluids[f].forces.data())); break;
                    default: break;
                }
            }
        }
    }

    // timestep_manager.rs:36-46
    R max_substep() const {
        R max_sq_vel = 0;
        for (const auto& f : fluids)
            for (size_t i = 0; i < f.n(); ++i) {
                const V3<R> u = f.velocities[i] + f.accelerations[i] * remaining_time;
                max_sq_vel = std::max(max_sq_vel, u.norm_squared());
            }
        return particle_radius * (R)2 / std::sqrt(max_sq_vel) * cfl_coeff;
    }
    void advance() {  // timestep_manager.rs:76-94
        R substep = total_step_size;
        if (cfl_mode) {  // the commented body of compute_substep (:90-93); na::clamp(v, lo, hi) = v > hi ? hi : (v < lo ? lo : v)
            const R min_substep = total_step_size / (R)max_num_substeps;
            const R max_sub = total_step_size / (R)min_num_substeps;
            const R computed = max_substep();
            substep = computed > max_sub ? max_sub : (computed < min_substep ? min_substep : computed);
            // mode 2: cut at the remaining time — and a remainder of float residue (below 1e-4 of the step) is taken along now instead
            // of becoming a last substep of a few ulps with inv_dt ~ 1e6
            if (cfl_mode == 2 && (substep > remaining_time || remaining_time - substep < total_step_size * (R)1e-4)) substep = remaining_time;
        }
        substeps_of_last_step.push_back((double)substep);
        dt = substep;
        inv_dt = (substep == (R)0) ? (R)0 : (R)1 / substep;
        remaining_time -= dt;
    }

    // dfsph_solver.rs:505-519 / iisph :458-471
    void integrate_and_clear_accelerations() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const long n = (long)fluids[f].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                velocity_changes[f][i] += fluids[f].accelerations[i] * dt;
                fluids[f].accelerations[i].fill(0);
            }
        }
    }

    // dfsph_solver.rs:667-708
    void dfsph_step(const V3<R>& gravity) {
        compute_alphas();
        // divergence_solve :466-503
        stats.n_div_iters = 0;
        for (int i = 0; i < max_divergence_iter; ++i) {
            R avg_err = compute_divergences();
            stats.div_error = (double)avg_err;
            R max_err = max_divergence_error * inv_dt * (R)0.01;
            if (avg_err <= max_err && i >= min_divergence_iter) break;
            compute_velocity_changes_for_divergence();
            stats.n_div_iters++;
        }
        // update_velocities :422-430 ; zero velocity changes :689-691
        for (size_t f = 0; f < fluids.size(); ++f) {
            const long n = (long)fluids[f].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                fluids[f].velocities[i] += velocity_changes[f][i];
                velocity_changes[f][i].fill(0);
            }
        }
        predict_advection(gravity);
        advance();
        integrate_and_clear_accelerations();
        // pressure_solve :432-464
        stats.n_press_iters = 0;
        for (int i = 0; i < max_pressure_iter; ++i) {
            R avg_err = compute_predicted_densities_dfsph();
            stats.density_error = (double)avg_err;
            if (avg_err <= max_density_error && i >= min_pressure_iter) break;
            compute_velocity_changes_dfsph();
            stats.n_press_iters++;
        }
        // update_positions :411-420
        for (size_t f = 0; f < fluids.size(); ++f) {
            const long n = (long)fluids[f].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i)
                fluids[f].positions[i] += (fluids[f].velocities[i] + velocity_changes[f][i]) * dt;
        }
    }

    // ================================================================== IISPH
    // iisph_solver.rs:92-142
    void compute_predicted_densities_iisph() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                R delta = 0;
                for (auto& c : ff[f].contacts[i]) {
                    const Fluid<R>& fluid_j = fluids[c.j_model];
                    V3<R> vi = fluid_i.velocities[c.i] + velocity_changes[c.i_model][c.i];
                    V3<R> vj = fluid_j.velocities[c.j] + velocity_changes[c.j_model][c.j];
                    delta += fluid_j.particle_mass(c.j) * (vi - vj).dot(c.gradient);
                }
                for (auto& c : fb[f].contacts[i]) {
                    V3<R> vi = fluid_i.velocities[c.i] + velocity_changes[c.i_model][c.i];
                    V3<R> vj = boundaries[c.j_model].velocities[c.j];
                    delta += boundaries[c.j_model].volumes[c.j] * fluid_i.density0 * (vi - vj).dot(c.gradient);
                }
                predicted_densities[f][i] = densities[f][i] + delta * dt;
                assert(predicted_densities[f][i] != 0);
            }
        }
    }
    // iisph_solver.rs:144-186
    void compute_dii() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                V3<R> d;
                R rhoi = densities[f][i];
                R factor = -dt * dt / (rhoi * rhoi);
                for (auto& c : ff[f].contacts[i]) {
                    R mj = fluids[c.j_model].particle_mass(c.j);
                    d += c.gradient * (mj * factor);
                }
                for (auto& c : fb[f].contacts[i]) {
                    R mj = boundaries[c.j_model].volumes[c.j] * fluid_i.density0;
                    d += c.gradient * (mj * factor);
                }
                dii[f][i] = d;
            }
        }
    }
    // iisph_solver.rs:188-233
    void compute_aii() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                R a = 0;
                R rhoi = densities[f][i];
                R mi = fluid_i.particle_mass(i);
                R factor = dt * dt * mi / (rhoi * rhoi);
                for (auto& c : ff[f].contacts[i]) {
                    R mj = fluids[c.j_model].particle_mass(c.j);
                    V3<R> dji = c.gradient * factor;
                    a += mj * (dii[f][c.i] - dji).dot(c.gradient);
                }
                for (auto& c : fb[f].contacts[i]) {
                    R mj = boundaries[c.j_model].volumes[c.j] * fluid_i.density0;
                    V3<R> dji = c.gradient * factor;
                    a += mj * (dii[f][c.i] - dji).dot(c.gradient);
                }
                aii[f][i] = a;
            }
        }
    }
    // iisph_solver.rs:235-268
    void compute_dij_pjl() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const long n = (long)fluids[f].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                V3<R> d;
                for (auto& c : ff[f].contacts[i]) {
                    R rhoj = densities[c.j_model][c.j];
                    R mj = fluids[c.j_model].particle_mass(c.j);
                    R p_jl = pressures[c.j_model][c.j];
                    d += c.gradient * (-mj * p_jl / (rhoj * rhoj));
                }
                d *= dt * dt;
                dij_pjl[f][i] = d;
            }
        }
    }
    // iisph_solver.rs:270-353
    R compute_next_pressures() {
        R max_error = 0;
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
            std::vector<R> errs(n);
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                R& next_pressure = next_pressures[f][i];
                if (std::abs(aii[f][i]) > (R)1.0e-9) {
                    R sum = 0;
                    R pi = pressures[f][i];
                    R mi = fluid_i.particle_mass(i);
                    R rhoi = densities[f][i];
                    R derr = fluid_i.density0 - predicted_densities[f][i];
                    for (auto& c : ff[f].contacts[i]) {
                        R mj = fluids[c.j_model].particle_mass(c.j);
                        V3<R> dji = c.gradient * (dt * dt * mi / (rhoi * rhoi));
                        V3<R> factor = dij_pjl[c.i_model][c.i] - dii[c.j_model][c.j] * pressures[c.j_model][c.j] -
                                       (dij_pjl[c.j_model][c.j] - dji * pi);
                        sum += mj * factor.dot(c.gradient);
                    }
                    for (auto& c : fb[f].contacts[i]) {
                        R mj = boundaries[c.j_model].volumes[c.j] * fluid_i.density0;
                        sum += mj * dij_pjl[c.i_model][c.i].dot(c.gradient);
                    }
                    next_pressure = ((R)1 - omega) * pi + omega * (derr - sum) / aii[f][i];
                    if (next_pressure > (R)0) {
                        errs[i] = (-sum - aii[f][i] * next_pressure) / fluid_i.density0;
                    } else {
                        next_pressure = 0; errs[i] = 0;
                    }
                } else {
                    next_pressure = 0; errs[i] = 0;
                }
            }
            R err = 0;
            for (long i = 0; i < n; ++i) err = err + errs[i];
            if (n != 0) max_error = std::max(max_error, err / (R)(double)n);
        }
        return max_error;
    }
    // iisph_solver.rs:355-404
    void compute_velocity_changes_iisph() {
        for (size_t f = 0; f < fluids.size(); ++f) {
            const Fluid<R>& fluid_i = fluids[f];
            const long n = (long)fluid_i.n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                V3<R>& velocity_change = velocity_changes[f][i];
                R pi = pressures[f][i];
                R rhoi = densities[f][i];
                for (auto& c : ff[f].contacts[i]) {
                    R mj = fluids[c.j_model].particle_mass(c.j);
                    R pj = pressures[c.j_model][c.j];
                    R rhoj = densities[c.j_model][c.j];
                    velocity_change -= c.gradient * (dt * mj * (pi / (rhoi * rhoi) + pj / (rhoj * rhoj)));
                }
                for (auto& c : fb[f].contacts[i]) {
                    R mj = boundaries[c.j_model].volumes[c.j] * fluid_i.density0;
                    V3<R> acc = c.gradient * (mj * pi / (rhoi * rhoi));
                    velocity_change -= acc * dt;
                    R mi = fluid_i.particle_mass(c.i);
                    apply_force(boundaries[c.j_model], c.j, acc * mi);
                }
            }
        }
    }
    // iisph_solver.rs:643-711
    void iisph_step(const V3<R>& gravity) {
        stats.n_div_iters = 0; stats.div_error = 0;
        predict_advection(gravity);
        advance();
        integrate_and_clear_accelerations();
        compute_dii();
        for (auto& v : pressures) for (auto& p : v) p *= (R)0.5;
        compute_predicted_densities_iisph();
        compute_aii();
        // pressure_solve :422-456
        stats.n_press_iters = 0;
        for (int i = 0; i < max_pressure_iter; ++i) {
            compute_dij_pjl();
            R avg_err = compute_next_pressures();
            stats.density_error = (double)avg_err;
            std::swap(pressures, next_pressures);
            stats.n_press_iters++;
            if (avg_err <= max_density_error && i >= min_pressure_iter) break;
        }
        compute_velocity_changes_iisph();
        // update_velocities_and_positions :406-420 ; zero velocity changes :707-709
        for (size_t f = 0; f < fluids.size(); ++f) {
            const long n = (long)fluids[f].n();
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (long i = 0; i < n; ++i) {
                fluids[f].velocities[i] += velocity_changes[f][i];
                fluids[f].positions[i] += fluids[f].velocities[i] * dt;
                velocity_changes[f][i].fill(0);
            }
        }
    }

    // ------------------------------------------------------------------ liquid_world.rs:67-158
    void step(R step_dt, const V3<R>& gravity) {
        double t0 = now_ms();
        total_step_size = step_dt; remaining_time = step_dt;  // timestep_manager.reset
        init_with_fluids();
        for (auto& fl : fluids) fl.apply_particles_removal();  // liquid_world.rs:80-82
        stats = StepStats{};
        substeps_of_last_step.clear();
        while (!(remaining_time <= Eps<R>::v)) {  // is_done, timestep_manager.rs:56-58
            double ta = now_ms();
            grid.clear();
            insert_fluids_to_grid();
            // coupling.update_boundaries(&timestep, ...) (liquid_world.rs:94-103): the host's manager, with timestep.dt() of the LAST substep
            if (substep_cb) substep_cb(substep_user, 0, (double)dt);
            update_boundaries_dynamic();
            insert_boundaries_to_grid();
            double tb = now_ms();
            compute_contacts();
            stats.ncontacts = ncontacts();
            double tc = now_ms();
            evaluate_kernels();
            double td = now_ms();
            compute_densities();
            if (solver_kind == 0) dfsph_step(gravity); else iisph_step(gravity);
            // coupling.transmit_forces(&timestep, boundaries) (liquid_world.rs:146): timestep.dt() is now THIS substep's
            if (substep_cb) substep_cb(substep_user, 1, (double)dt);
            double te = now_ms();
            stats.t_grid_ms += tb - ta; stats.t_contacts_ms += tc - tb;
            stats.t_kernels_ms += td - tc; stats.t_solver_ms += te - td;
        }
        stats.t_total_ms = now_ms() - t0;
    }
};

// ---------------------------------------------------------------------------------------------------
// Type-erased handle for the C ABI (f32 and f64 builds of the same restatement).
// ---------------------------------------------------------------------------------------------------
struct Handle {
    bool f64;
    World<float>* wf = nullptr;
    World<double>* wd = nullptr;
};

}  // namespace so

using namespace so;

#define DISPATCH(h, expr_f, expr_d) do { if ((h)->f64) { auto& w = *(h)->wd; expr_d; } else { auto& w = *(h)->wf; expr_f; } } while (0)

template <typename R>
static int add_fluid_t(World<R>& w, uint64_t n, const float* pos, const float* vel, float density0, uint32_t mem, uint32_t filt) {
    Fluid<R> f;
    f.positions.resize(n); f.velocities.resize(n); f.accelerations.resize(n);
    R pr = w.particle_radius;
    R vol = pr * pr * pr * (R)(8.0 * 0.8);  // fluid.rs:110-120
    f.volumes.assign(n, vol);
    for (uint64_t i = 0; i < n; ++i) {
        f.positions[i] = V3<R>((R)pos[3 * i], (R)pos[3 * i + 1], (R)pos[3 * i + 2]);
        if (vel) f.velocities[i] = V3<R>((R)vel[3 * i], (R)vel[3 * i + 1], (R)vel[3 * i + 2]);
    }
    f.density0 = (R)density0;
    f.groups.memberships = mem; f.groups.filter = filt;
    w.fluids.push_back(std::move(f));
    return (int)w.fluids.size() - 1;
}

// ---- integrations/rapier/fluids_pipeline.rs (rapier itself is an un-vendored dependency: the two rigid-body formulas used
// here are restated from rapier3d 0.23's published source, `RigidBody::velocity_at_point` = linvel + angvel x (pt - world_com)
// and `apply_impulse_at_point` = apply_impulse(J) + apply_torque_impulse((pt - world_com) x J); nalgebra's
// `Isometry3 * Point3` = UnitQuaternion * pt + translation with q * v = v + w t + q.vec x t, t = 2 q.vec x v).
template <typename R>
static void set_boundary_sampling_t(World<R>& w, int b, uint64_t n, const float* pts) {
    Boundary<R>& bd = w.boundaries[b];
    bd.sampling.resize(n);
    for (uint64_t i = 0; i < n; ++i) bd.sampling[i] = V3<R>((R)pts[3 * i], (R)pts[3 * i + 1], (R)pts[3 * i + 2]);
}
// The StaticSampling arm of ColliderCouplingManager::update_boundaries (:160-193, :262):
//   boundary.forces = Some/None by body.is_dynamic() (:163-171); positions[i] = collider.position() * pt (:182);
//   velocities[i] = body.velocity_at_point(pt) — with the LOCAL point, as written (:183) — or zero without a body;
//   volumes reset to 0 (:190); clear_forces(true) (:262).
// pose = translation[3], rotation (i, j, k, w), linvel[3], angvel[3], world_com[3]
template <typename R>
static void update_boundary_pose_t(World<R>& w, int b, const double* pose, int has_body, int is_dynamic) {
    Boundary<R>& bd = w.boundaries[b];
    const V3<R> t((R)pose[0], (R)pose[1], (R)pose[2]);
    const V3<R> qv((R)pose[3], (R)pose[4], (R)pose[5]);
    const R qw = (R)pose[6];
    const V3<R> linvel((R)pose[7], (R)pose[8], (R)pose[9]), angvel((R)pose[10], (R)pose[11], (R)pose[12]);
    const V3<R> com((R)pose[13], (R)pose[14], (R)pose[15]);
    if (has_body) {
        bd.has_forces = is_dynamic != 0;
        if (!bd.has_forces) bd.forces.clear();
    }
    if (bd.dyn_kind) {  // DynamicContactSampling: the projection itself runs inside step() (update_boundaries_dynamic)
        bd.pose_t = t; bd.pose_qv = qv; bd.pose_qw = qw; bd.pose_linvel = linvel; bd.pose_angvel = angvel; bd.pose_com = com;
        bd.pose_has_body = has_body != 0;
        return;
    }
    const size_t n = bd.sampling.size();
    bd.positions.resize(n); bd.velocities.resize(n); bd.volumes.assign(n, (R)0);
    for (size_t i = 0; i < n; ++i) {
        const V3<R> pt = bd.sampling[i];
        const V3<R> tt = qv.cross(pt) * (R)2;
        const V3<R> cr = qv.cross(tt);
        bd.positions[i] = (tt * qw + cr + pt) + t;
        bd.velocities[i] = has_body ? linvel + angvel.cross(pt - com) : V3<R>();
    }
    if (bd.has_forces) bd.forces.assign(n, V3<R>());
}
// transmit_forces (:266-287): body.apply_impulse_at_point(force * dt, pos) for every boundary particle, i.e. the sums
// returned here times dt.  (rapier accumulates them one by one in f32 into linvel / angvel; sums are taken in R here.)
template <typename R>
static void boundary_wrench_t(World<R>& w, int b, const double* com, double* force, double* torque) {
    Boundary<R>& bd = w.boundaries[b];
    V3<R> F, T;
    const V3<R> c((R)com[0], (R)com[1], (R)com[2]);
    if (bd.has_forces)
        for (size_t i = 0; i < bd.n(); ++i) { F += bd.forces[i]; T += (bd.positions[i] - c).cross(bd.forces[i]); }
    force[0] = F.x; force[1] = F.y; force[2] = F.z; torque[0] = T.x; torque[1] = T.y; torque[2] = T.z;
}

// host edit of boundary.positions / boundary.velocities (pub fields, boundary.rs:13-15); same particle count
template <typename R>
static void set_boundary_particles_t(World<R>& w, int b, uint64_t n, const float* pos, const float* vel) {
    Boundary<R>& bd = w.boundaries[b];
    bd.positions.resize(n); bd.velocities.resize(n); bd.volumes.assign(n, (R)0);
    if (bd.has_forces) bd.forces.resize(n, V3<R>());
    for (uint64_t i = 0; i < n; ++i) {
        bd.positions[i] = V3<R>((R)pos[3 * i], (R)pos[3 * i + 1], (R)pos[3 * i + 2]);
        bd.velocities[i] = vel ? V3<R>((R)vel[3 * i], (R)vel[3 * i + 1], (R)vel[3 * i + 2]) : V3<R>();
    }
}
template <typename R>
static int add_boundary_t(World<R>& w, uint64_t n, const float* pos, const float* vel, uint32_t mem, uint32_t filt, int wants_forces) {
    Boundary<R> b;
    b.positions.resize(n); b.velocities.resize(n); b.volumes.assign(n, 0);
    for (uint64_t i = 0; i < n; ++i) {
        b.positions[i] = V3<R>((R)pos[3 * i], (R)pos[3 * i + 1], (R)pos[3 * i + 2]);
        if (vel) b.velocities[i] = V3<R>((R)vel[3 * i], (R)vel[3 * i + 1], (R)vel[3 * i + 2]);
    }
    b.groups.memberships = mem; b.groups.filter = filt;
    b.has_forces = wants_forces != 0;
    if (b.has_forces) b.forces.assign(n, V3<R>());
    w.boundaries.push_back(std::move(b));
    return (int)w.boundaries.size() - 1;
}

template <typename R>
static void copy_v3(const std::vector<V3<R>>& v, double* out) {
    for (size_t i = 0; i < v.size(); ++i) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].z; }
}
template <typename R>
static void copy_s(const std::vector<R>& v, double* out) { for (size_t i = 0; i < v.size(); ++i) out[i] = v[i]; }

extern "C" {

struct so_stats {
    int32_t n_div_iters, n_press_iters;
    double div_error, density_error;
    uint64_t ncontacts;
    double t_grid_ms, t_contacts_ms, t_kernels_ms, t_solver_ms, t_total_ms;
};

void* so_create(int use_f64, float particle_radius, float smoothing_factor, int solver_kind, int nthreads) {
    Handle* h = new Handle();
    h->f64 = use_f64 != 0;
    if (h->f64) { h->wd = new World<double>((double)particle_radius, (double)smoothing_factor, solver_kind); h->wd->nthreads = nthreads; }
    else { h->wf = new World<float>(particle_radius, smoothing_factor, solver_kind); h->wf->nthreads = nthreads; }
    return h;
}
void so_destroy(void* p) { Handle* h = (Handle*)p; delete h->wf; delete h->wd; delete h; }

void so_set_shuffle_seed(void* p, uint64_t seed) { Handle* h = (Handle*)p; DISPATCH(h, w.shuffle_seed = seed, w.shuffle_seed = seed); }
void so_set_kernels(void* p, int density, int gradient) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { w.kernel_density = density; w.kernel_gradient = gradient; }, { w.kernel_density = density; w.kernel_gradient = gradient; });
}
void so_set_threads(void* p, int n) { Handle* h = (Handle*)p; DISPATCH(h, w.nthreads = n, w.nthreads = n); }
void so_set_solver_params(void* p, int min_p, int max_p, float max_derr, int min_d, int max_d, float max_diverr) {
    Handle* h = (Handle*)p;
    DISPATCH(h,
        (w.min_pressure_iter = min_p, w.max_pressure_iter = max_p, w.max_density_error = max_derr, w.min_divergence_iter = min_d, w.max_divergence_iter = max_d, w.max_divergence_error = max_diverr),
        (w.min_pressure_iter = min_p, w.max_pressure_iter = max_p, w.max_density_error = max_derr, w.min_divergence_iter = min_d, w.max_divergence_iter = max_d, w.max_divergence_error = max_diverr));
}
// TimestepManager::{dt, inv_dt} (timestep_manager.rs:14-15) as a previous run left them: what the next step reads before it advances
void so_set_timestep(void* p, float dt, float inv_dt) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { w.dt = dt; w.inv_dt = inv_dt; }, { w.dt = dt; w.inv_dt = inv_dt; });
}
// opt-in CFL sub-stepping (timestep_manager.rs:36-46 + the commented clamp :90-93); mode 0 restores the reference's running behaviour
void so_set_cfl(void* p, int mode, float cfl_coeff, int min_substeps, int max_substeps) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { w.cfl_mode = mode; w.cfl_coeff = cfl_coeff; w.min_num_substeps = min_substeps; w.max_num_substeps = max_substeps; },
                { w.cfl_mode = mode; w.cfl_coeff = (double)cfl_coeff; w.min_num_substeps = min_substeps; w.max_num_substeps = max_substeps; });
}
// the substep lengths of the last step (counters.nsubsteps of them); returns their number
int so_get_substeps(void* p, double* out, int cap) {
    Handle* h = (Handle*)p; int n = 0;
    DISPATCH(h, { n = (int)w.substeps_of_last_step.size(); for (int i = 0; i < n && i < cap; ++i) out[i] = w.substeps_of_last_step[i]; },
                { n = (int)w.substeps_of_last_step.size(); for (int i = 0; i < n && i < cap; ++i) out[i] = w.substeps_of_last_step[i]; });
    return n;
}
double so_h(void* p) { Handle* h = (Handle*)p; double r = 0; DISPATCH(h, r = w.h, r = w.h); return r; }

int so_add_fluid(void* p, uint64_t n, const float* pos, const float* vel, float density0, uint32_t mem, uint32_t filt) {
    Handle* h = (Handle*)p; int r = -1;
    DISPATCH(h, r = add_fluid_t(w, n, pos, vel, density0, mem, filt), r = add_fluid_t(w, n, pos, vel, density0, mem, filt));
    return r;
}
int so_add_boundary(void* p, uint64_t n, const float* pos, const float* vel, uint32_t mem, uint32_t filt, int wants_forces) {
    Handle* h = (Handle*)p; int r = -1;
    DISPATCH(h, r = add_boundary_t(w, n, pos, vel, mem, filt, wants_forces), r = add_boundary_t(w, n, pos, vel, mem, filt, wants_forces));
    return r;
}
// kind: 1 XSPH(p0 fluid coeff, p1 boundary coeff); 2 Artificial(p0, p1, alpha, beta, speed_of_sound); 3 Akinci2013(p0 tension, p1 adhesion);
// 4 DFSPHViscosity(p0 coefficient, p1 min iter, p2 max iter, p3 max error); 5 He2014(p0 fluid tension, p1 boundary tension);
// 6 WCSPHSurfaceTension(p0 fluid tension, p1 boundary tension)
int so_add_force(void* p, int fluid, int kind, const float* params, int nparams) {
    Handle* h = (Handle*)p;
    DISPATCH(h,
        { Force<float> f; f.kind = kind; for (int i = 0; i < nparams && i < 5; ++i) f.p[i] = params[i]; w.fluids[fluid].forces.push_back(f); },
        { Force<double> f; f.kind = kind; for (int i = 0; i < nparams && i < 5; ++i) f.p[i] = (double)params[i]; w.fluids[fluid].forces.push_back(f); });
    return 0;
}
void so_set_fluid_velocities(void* p, int fluid, const float* vel) {
    Handle* h = (Handle*)p;
    DISPATCH(h,
        { auto& f = w.fluids[fluid]; for (size_t i = 0; i < f.n(); ++i) f.velocities[i] = V3<float>(vel[3*i], vel[3*i+1], vel[3*i+2]); },
        { auto& f = w.fluids[fluid]; for (size_t i = 0; i < f.n(); ++i) f.velocities[i] = V3<double>(vel[3*i], vel[3*i+1], vel[3*i+2]); });
}
void so_set_fluid_volumes(void* p, int fluid, const float* vol) {
    Handle* h = (Handle*)p;
    DISPATCH(h,
        { auto& f = w.fluids[fluid]; for (size_t i = 0; i < f.n(); ++i) f.volumes[i] = vol[i]; },
        { auto& f = w.fluids[fluid]; for (size_t i = 0; i < f.n(); ++i) f.volumes[i] = (double)vol[i]; });
}

void so_step(void* p, float dt, float gx, float gy, float gz, so_stats* out) {
    Handle* h = (Handle*)p;
    StepStats s{};
    DISPATCH(h, (w.step(dt, V3<float>(gx, gy, gz)), s = w.stats), (w.step((double)dt, V3<double>(gx, gy, gz)), s = w.stats));
    if (out) {
        out->n_div_iters = s.n_div_iters; out->n_press_iters = s.n_press_iters;
        out->div_error = s.div_error; out->density_error = s.density_error; out->ncontacts = s.ncontacts;
        out->t_grid_ms = s.t_grid_ms; out->t_contacts_ms = s.t_contacts_ms; out->t_kernels_ms = s.t_kernels_ms;
        out->t_solver_ms = s.t_solver_ms; out->t_total_ms = s.t_total_ms;
    }
}

uint64_t so_fluid_len(void* p, int fluid) { Handle* h = (Handle*)p; uint64_t r = 0; DISPATCH(h, r = w.fluids[fluid].n(), r = w.fluids[fluid].n()); return r; }
uint64_t so_boundary_len(void* p, int b) { Handle* h = (Handle*)p; uint64_t r = 0; DISPATCH(h, r = w.boundaries[b].n(), r = w.boundaries[b].n()); return r; }

// field: 0 positions, 1 velocities, 2 velocity_changes, 3 accelerations, 4 dii, 5 dij_pjl, 6 akinci normals (force index 0..)
void so_get_fluid_vec(void* p, int fluid, int field, double* out) {
    Handle* h = (Handle*)p;
#define GETV(w) do { switch (field) { \
        case 0: copy_v3(w.fluids[fluid].positions, out); break; \
        case 1: copy_v3(w.fluids[fluid].velocities, out); break; \
        case 2: copy_v3(w.velocity_changes[fluid], out); break; \
        case 3: copy_v3(w.fluids[fluid].accelerations, out); break; \
        case 4: copy_v3(w.dii[fluid], out); break; \
        case 5: copy_v3(w.dij_pjl[fluid], out); break; \
        case 6: for (auto& f : w.fluids[fluid].forces) if (f.kind == FORCE_AKINCI2013) { copy_v3(f.normals, out); break; } break; \
        default: break; } } while (0)
    DISPATCH(h, GETV(w), GETV(w));
#undef GETV
}
// 1 if a step hit a code path on which the reference would panic (WCSPHSurfaceTension's boundary loop, see solve_wcsph_tension)
// the coupling manager's two calls per substep: phase 0 = update_boundaries (dt of the last substep), 1 = transmit_forces (this substep's)
void so_set_substep_callback(void* p, void (*cb)(void*, int, double), void* user) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { w.substep_cb = cb; w.substep_user = user; }, { w.substep_cb = cb; w.substep_user = user; });
}
// user force callback for FORCE_CUSTOM entries (kind 7 of so_add_force)
void so_set_force_callback(void* p, void (*cb)(void*, int, int, uint64_t, const double*, const double*, const double*, double*), void* user) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { w.custom_cb = cb; w.custom_user = user; }, { w.custom_cb = cb; w.custom_user = user; });
}
void so_add_particles(void* p, int fluid, uint64_t n, const float* pos, const float* vel) {
    Handle* h = (Handle*)p;
#define ADDP(w, R) do { R pr = w.particle_radius; w.fluids[fluid].add_particles(n, pos, vel, pr * pr * pr * (R)(8.0 * 0.8)); } while (0)
    DISPATCH(h, ADDP(w, float), ADDP(w, double));
#undef ADDP
}
void so_delete_particle(void* p, int fluid, uint64_t i) {
    Handle* h = (Handle*)p;
    DISPATCH(h, w.fluids[fluid].delete_particle_at_next_timestep(i), w.fluids[fluid].delete_particle_at_next_timestep(i));
}
// LiquidWorld::remove_boundary (liquid_world.rs:176-178): swap-remove; boundaries carry no solver state between steps
void so_remove_boundary(void* p, int b) {
    Handle* h = (Handle*)p;
#define RMB(w) do { if ((size_t)b + 1 != w.boundaries.size()) std::swap(w.boundaries[b], w.boundaries.back()); w.boundaries.pop_back(); } while (0)
    DISPATCH(h, RMB(w), RMB(w));
#undef RMB
}
void so_set_boundary_particles(void* p, int b, uint64_t n, const float* pos, const float* vel) {
    Handle* h = (Handle*)p;
    DISPATCH(h, set_boundary_particles_t(w, b, n, pos, vel), set_boundary_particles_t(w, b, n, pos, vel));
}
void so_remove_fluid(void* p, int fluid) {
    Handle* h = (Handle*)p;
    DISPATCH(h, w.remove_fluid(fluid), w.remove_fluid(fluid));
}
int so_num_forces(void* p, int fluid) { Handle* h = (Handle*)p; int r = 0; DISPATCH(h, r = (int)w.fluids[fluid].forces.size(), r = (int)w.fluids[fluid].forces.size()); return r; }
int so_reference_would_panic(void* p) { Handle* h = (Handle*)p; int r = 0; DISPATCH(h, r = w.reference_would_panic, r = w.reference_would_panic); return r; }
// field: 0 densities, 1 alphas, 2 divergences, 3 predicted_densities, 4 volumes, 5 aii, 6 pressures, 7 He2014 colors, 8 He2014 gradcs
void so_get_fluid_scalar(void* p, int fluid, int field, double* out) {
    Handle* h = (Handle*)p;
#define GETS(w) do { switch (field) { \
        case 0: copy_s(w.densities[fluid], out); break; \
        case 1: copy_s(w.alphas[fluid], out); break; \
        case 2: copy_s(w.divergences[fluid], out); break; \
        case 3: copy_s(w.predicted_densities[fluid], out); break; \
        case 4: copy_s(w.fluids[fluid].volumes, out); break; \
        case 5: copy_s(w.aii[fluid], out); break; \
        case 6: copy_s(w.pressures[fluid], out); break; \
        case 7: for (auto& f : w.fluids[fluid].forces) if (f.kind == FORCE_HE2014) { copy_s(f.colors, out); break; } break; \
        case 8: for (auto& f : w.fluids[fluid].forces) if (f.kind == FORCE_HE2014) { copy_s(f.gradcs, out); break; } break; \
        default: break; } } while (0)
    DISPATCH(h, GETS(w), GETS(w));
#undef GETS
}
// number of fluid-fluid (which = 0) / fluid-boundary (which = 1) contacts of each particle of `fluid`
void so_get_contact_counts(void* p, int fluid, int which, uint32_t* out) {
    Handle* h = (Handle*)p;
    DISPATCH(h,
        { auto& pc = which ? w.fb[fluid] : w.ff[fluid]; for (size_t i = 0; i < pc.contacts.size(); ++i) out[i] = (uint32_t)pc.contacts[i].size(); },
        { auto& pc = which ? w.fb[fluid] : w.ff[fluid]; for (size_t i = 0; i < pc.contacts.size(); ++i) out[i] = (uint32_t)pc.contacts[i].size(); });
}
// Sorted (j_model << 32 | j) keys of the contacts of one particle; returns count (writes at most cap entries).
uint64_t so_get_contacts_of(void* p, int fluid, int which, uint64_t i, uint64_t* out, uint64_t cap) {
    Handle* h = (Handle*)p; uint64_t n = 0;
#define GETC(w) do { auto& v = (which ? w.fb[fluid] : w.ff[fluid]).contacts[i]; n = v.size(); std::vector<uint64_t> k; \
        for (auto& c : v) k.push_back(((uint64_t)c.j_model << 32) | (uint64_t)c.j); std::sort(k.begin(), k.end()); \
        for (uint64_t q = 0; q < n && q < cap; ++q) out[q] = k[q]; } while (0)
    DISPATCH(h, GETC(w), GETC(w));
#undef GETC
    return n;
}
// DFSPHViscosity: iterations / last error of the force at index `force` of `fluid` after the last step; betas (36 per
// particle, row-major) of the same force.
void so_get_viscosity_stats(void* p, int fluid, int force, int* iters, double* err) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { auto& f = w.fluids[fluid].forces[force]; *iters = f.last_visc_iters; *err = (double)f.last_visc_error; },
                { auto& f = w.fluids[fluid].forces[force]; *iters = f.last_visc_iters; *err = (double)f.last_visc_error; });
}
void so_get_viscosity_betas(void* p, int fluid, int force, double* out) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { auto& f = w.fluids[fluid].forces[force]; for (size_t k = 0; k < f.betas.size(); ++k) out[k] = (double)f.betas[k]; },
                { auto& f = w.fluids[fluid].forces[force]; for (size_t k = 0; k < f.betas.size(); ++k) out[k] = (double)f.betas[k]; });
}
// self-check hook: inverse of a 6x6 matrix with the oracle's LU (row-major in / out); returns 1 if invertible, det in *det
int so_test_lu6(const double* a_in, double* inv_out, double* det) {
    double a[6][6], inv[6][6], d = 0;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) a[i][j] = a_in[i * 6 + j];
    const bool ok = World<double>::lu_inverse6(a, inv, d);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) inv_out[i * 6 + j] = inv[i][j];
    *det = d;
    return ok ? 1 : 0;
}
// field: 0 positions, 1 velocities, 2 forces
void so_get_boundary_vec(void* p, int b, int field, double* out) {
    Handle* h = (Handle*)p;
#define GETB(w) do { switch (field) { \
        case 0: copy_v3(w.boundaries[b].positions, out); break; \
        case 1: copy_v3(w.boundaries[b].velocities, out); break; \
        case 2: copy_v3(w.boundaries[b].forces, out); break; default: break; } } while (0)
    DISPATCH(h, GETB(w), GETB(w));
#undef GETB
}

void so_set_boundary_sampling(void* p, int b, uint64_t n, const float* pts) {
    Handle* h = (Handle*)p;
    DISPATCH(h, set_boundary_sampling_t(w, b, n, pts), set_boundary_sampling_t(w, b, n, pts));
}
// ColliderSampling::DynamicContactSampling for boundary `b`: kind 1 = ball (params[0] = radius), 2 = cuboid (half extents)
void so_set_boundary_dynamic_sampling(void* p, int b, int kind, const float* params) {
    Handle* h = (Handle*)p;
#define SETD(w) do { auto& bd = w.boundaries[b]; bd.dyn_kind = kind; for (int a = 0; a < 3; ++a) bd.dyn_p[a] = params[a]; } while (0)
    DISPATCH(h, SETD(w), SETD(w));
#undef SETD
}
// (fluid, particle) of the fluid particle each point of a dynamically sampled boundary was projected from
void so_get_boundary_sources(void* p, int b, uint32_t* fluid, uint32_t* particle) {
    Handle* h = (Handle*)p;
#define GETS(w) do { auto& v = w.boundaries[b].dyn_source; for (size_t i = 0; i < v.size(); ++i) { fluid[i] = v[i].first; particle[i] = v[i].second; } } while (0)
    DISPATCH(h, GETS(w), GETS(w));
#undef GETS
}
void so_update_boundary_pose(void* p, int b, const double* pose16, int has_body, int is_dynamic) {
    Handle* h = (Handle*)p;
    DISPATCH(h, update_boundary_pose_t(w, b, pose16, has_body, is_dynamic), update_boundary_pose_t(w, b, pose16, has_body, is_dynamic));
}
void so_get_boundary_wrench(void* p, int b, const double* com, double* force, double* torque) {
    Handle* h = (Handle*)p;
    DISPATCH(h, boundary_wrench_t(w, b, com, force, torque), boundary_wrench_t(w, b, com, force, torque));
}
void so_get_boundary_volumes(void* p, int b, double* out) {
    Handle* h = (Handle*)p;
    DISPATCH(h, copy_s(w.boundaries[b].volumes, out), copy_s(w.boundaries[b].volumes, out));
}
void so_clear_boundary_forces(void* p, int b) {
    Handle* h = (Handle*)p;
    DISPATCH(h, { for (auto& f : w.boundaries[b].forces) f.fill(0); }, { for (auto& f : w.boundaries[b].forces) f.fill(0); });
}

// Scalar kernel evaluations, for the analytic self-checks (f32 build).
float so_kernel_w(float r, float h) { return CubicSpline<float>::scalar_apply(r, h); }
float so_kernel_dw(float r, float h) { return CubicSpline<float>::scalar_apply_diff(r, h); }
double so_kernel_w_f64(double r, double h) { return CubicSpline<double>::scalar_apply(r, h); }
double so_kernel_dw_f64(double r, double h) { return CubicSpline<double>::scalar_apply_diff(r, h); }
double so_kernel_scalar_f64(int kind, int diff, double r, double h) { return kernel_scalar<double>(kind, diff != 0, r, h); }
float so_kernel_scalar(int kind, int diff, float r, float h) { return kernel_scalar<float>(kind, diff != 0, r, h); }
float so_cohesion_kernel(float r, float h) { return World<float>::cohesion_kernel(r, h); }
float so_adhesion_kernel(float r, float h) { return World<float>::adhesion_kernel(r, h); }
int so_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
