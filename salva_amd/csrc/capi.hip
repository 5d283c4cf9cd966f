// capi.hip — the extern "C" boundary of libsalva_hip.so (declared in include/salva_hip.h).
// Exceptions never cross the ABI: every entry point maps them to an error code + thread-local message.
#include <cstring>
#include <string>

#include <memory>
#include <mutex>

#include "world.h"

using salva::World;

// Every entry point that takes a world holds the world's lock for its duration: the `&self` methods of the salva3d API
// (`particles_intersecting_aabb`, `fluids()`, `counters` ...) may be called from several threads at once — `LiquidWorld: Send + Sync` is
// pinned by the reference's own test (/root/reference/src/liquid_world.rs:283-287), and a bevy `Res<FluidsPipeline>` is shared between
// systems — while the C side keeps scratch buffers, a stream and lazily refreshed staging arrays per world.  Recursive: a force or
// shape callback re-enters the library from inside salva_hip_step on the thread that holds the lock.
struct SalvaHipWorld {
    World* w;
    mutable std::recursive_mutex mu;
};
struct WorldLock {
    std::unique_lock<std::recursive_mutex> l;
    explicit WorldLock(const SalvaHipWorld* world) { if (world) l = std::unique_lock<std::recursive_mutex>(world->mu); }
};

static thread_local std::string g_last_error;

template <typename F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const salva::HipError& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return SALVA_HIP_E_HIP;
    } catch (...) {
        g_last_error = "unknown error";
        return SALVA_HIP_E_HIP;
    }
}

static void not_in_force_callback(const SalvaHipWorld* world) {
    if (world->w->in_force_callback())
        throw salva::HipError(SALVA_HIP_E_INVALID, "this entry point is not available inside a force callback");
}

extern "C" {

void salva_hip_default_params(SalvaHipParams* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->particle_radius = 0.05f;
    p->smoothing_factor = 2.0f;
    p->solver = SALVA_HIP_SOLVER_DFSPH;
    p->min_pressure_iter = 1;      // dfsph_solver.rs:56 / iisph_solver.rs:50
    p->max_pressure_iter = 50;     // :57 / :51
    p->max_density_error = 0.05f;  // :58 / :52
    p->min_divergence_iter = 1;    // :59
    p->max_divergence_iter = 50;   // :60
    p->max_divergence_error = 0.1f;  // :61
    p->device = 0;
    p->enable_timers = 0;
}

int salva_hip_create(const SalvaHipParams* params, SalvaHipWorld** out) {
    return guarded([&]() -> int {
        if (!params || !out) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        *out = nullptr;
        World* w = new World(*params);
        *out = new SalvaHipWorld{w, {}};
        return SALVA_HIP_OK;
    });
}

void salva_hip_destroy(SalvaHipWorld* world) {
    if (!world) return;
    {   // (whoever still runs an entry point on another thread finishes first; using the world after this call is the caller's bug)
        WorldLock _lk(world);
        try { delete world->w; } catch (...) {}
        world->w = nullptr;
    }
    delete world;
}

float salva_hip_h(const SalvaHipWorld* world) { WorldLock _lk(world); return world ? world->w->sc.h : 0.0f; }

int salva_hip_set_fluid(SalvaHipWorld* world, uint32_t slot, uint64_t n, const float* positions_xyz,
                        const float* velocities_xyz, const float* volumes, const float* accelerations_xyz,
                        const float* velocity_changes_xyz, float density0, uint32_t memberships, uint32_t filter,
                        uint32_t dirty_mask) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->set_fluid(slot, n, positions_xyz, velocities_xyz, volumes, accelerations_xyz, velocity_changes_xyz,
                            density0, memberships, filter, dirty_mask);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_fluid_forces(SalvaHipWorld* world, uint32_t slot, const SalvaHipForceDesc* forces, uint32_t nforces) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world || (nforces && !forces)) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        not_in_force_callback(world);
        world->w->set_fluid_forces(slot, forces, nforces);
        return SALVA_HIP_OK;
    });
}

int salva_hip_remove_fluid(SalvaHipWorld* world, uint32_t slot) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->remove_fluid(slot);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_boundary(SalvaHipWorld* world, uint32_t slot, uint64_t n, const float* positions_xyz,
                           const float* velocities_xyz, uint32_t memberships, uint32_t filter, int32_t wants_forces) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->set_boundary(slot, n, positions_xyz, velocities_xyz, memberships, filter, wants_forces != 0);
        return SALVA_HIP_OK;
    });
}

int salva_hip_remove_boundary(SalvaHipWorld* world, uint32_t slot) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->remove_boundary(slot);
        return SALVA_HIP_OK;
    });
}

uint32_t salva_hip_num_fluids(const SalvaHipWorld* world) { WorldLock _lk(world); return world ? (uint32_t)world->w->fluids.size() : 0; }
uint32_t salva_hip_num_boundaries(const SalvaHipWorld* world) { WorldLock _lk(world); return world ? (uint32_t)world->w->bounds.size() : 0; }
uint64_t salva_hip_fluid_len(const SalvaHipWorld* world, uint32_t slot) { WorldLock _lk(world);
    return (world && slot < world->w->fluids.size()) ? world->w->fluids[slot].n : 0;
}
uint64_t salva_hip_boundary_len(const SalvaHipWorld* world, uint32_t slot) { WorldLock _lk(world);
    return (world && slot < world->w->bounds.size()) ? world->w->bounds[slot].n : 0;
}

int salva_hip_step(SalvaHipWorld* world, float dt, const float gravity[3], SalvaHipStepStats* stats) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world || !gravity) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        not_in_force_callback(world);
        return world->w->step(dt, gravity, stats);
    });
}

int salva_hip_get_fluid(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->get_fluid(slot, positions_xyz, velocities_xyz);
        return SALVA_HIP_OK;
    });
}

int salva_hip_get_fluid_field(SalvaHipWorld* world, uint32_t slot, int32_t field, float* out) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->get_fluid_field(slot, field, out);
        return SALVA_HIP_OK;
    });
}

int salva_hip_get_boundary(SalvaHipWorld* world, uint32_t slot, float* volumes, float* forces_xyz) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->get_boundary(slot, volumes, forces_xyz);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_boundary_sampling(SalvaHipWorld* world, uint32_t slot, uint64_t n, const float* local_points_xyz,
                                    uint32_t memberships, uint32_t filter) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->set_boundary_sampling(slot, n, local_points_xyz, memberships, filter);
        return SALVA_HIP_OK;
    });
}

int salva_hip_update_boundary_pose(SalvaHipWorld* world, uint32_t slot, const SalvaHipRigidPose* pose) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        if (!pose) throw salva::HipError(SALVA_HIP_E_INVALID, "null pose");
        not_in_force_callback(world);
        world->w->update_boundary_pose(slot, *pose);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_boundary_dynamic_sampling(SalvaHipWorld* world, uint32_t slot, const SalvaHipShape* collider_shape,
                                            uint32_t memberships, uint32_t filter) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        if (!collider_shape) throw salva::HipError(SALVA_HIP_E_INVALID, "null shape");
        not_in_force_callback(world);
        world->w->set_boundary_dynamic_sampling(slot, *collider_shape, memberships, filter);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_boundary_dynamic_sampling_host(SalvaHipWorld* world, uint32_t slot, const SalvaHipHostShape* collider_shape,
                                                 uint32_t memberships, uint32_t filter) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        if (!collider_shape) throw salva::HipError(SALVA_HIP_E_INVALID, "null shape");
        not_in_force_callback(world);
        world->w->set_boundary_dynamic_sampling_host(slot, *collider_shape, memberships, filter);
        return SALVA_HIP_OK;
    });
}

int salva_hip_get_dist_timing(const SalvaHipWorld* world, double out4[4]) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world || !out4) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        world->w->get_dist_timing(out4);
        return SALVA_HIP_OK;
    });
}
uint64_t salva_hip_local_len(const SalvaHipWorld* world) { WorldLock _lk(world); return world ? world->w->local_len() : 0; }
int salva_hip_get_local(SalvaHipWorld* world, uint32_t* ids, uint32_t* fluid_slots, uint8_t* is_ghost, float* positions_xyz,
                        float* velocities_xyz, float* densities, float* volumes) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->get_local(ids, fluid_slots, is_ghost, positions_xyz, velocities_xyz, densities, volumes);
        return SALVA_HIP_OK;
    });
}
int64_t salva_hip_get_local_contacts(SalvaHipWorld* world, int32_t boundary_contacts, uint64_t* offsets, uint32_t* j_model, uint32_t* j,
                                     uint64_t capacity) { WorldLock _lk(world);
    int64_t total = 0;
    const int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        total = (int64_t)world->w->get_local_contacts(boundary_contacts, offsets, j_model, j, capacity);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? total : (int64_t)rc;
}
int salva_hip_force_add_local_accelerations(SalvaHipWorld* world, const float* accelerations_xyz) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->force_add_local_accelerations(accelerations_xyz);
        return SALVA_HIP_OK;
    });
}
int salva_hip_get_fluid_async(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->get_fluid_async(slot, positions_xyz, velocities_xyz);
        return SALVA_HIP_OK;
    });
}
int salva_hip_wait_download(SalvaHipWorld* world) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->wait_download();
        return SALVA_HIP_OK;
    });
}
void* salva_hip_host_alloc(SalvaHipWorld* world, uint64_t bytes) { WorldLock _lk(world);
    void* p = nullptr;
    const int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        SALVA_HIP_CHECK(hipSetDevice(world->w->prm.device));
        SALVA_HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? p : nullptr;
}
int salva_hip_host_free(void* p) {
    return guarded([&]() -> int {
        if (p) SALVA_HIP_CHECK(hipHostFree(p));
        return SALVA_HIP_OK;
    });
}
int salva_hip_host_register(SalvaHipWorld* world, void* p, uint64_t bytes) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world || !p || !bytes) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        SALVA_HIP_CHECK(hipSetDevice(world->w->prm.device));
        SALVA_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
        return SALVA_HIP_OK;
    });
}
int salva_hip_host_unregister(void* p) {
    return guarded([&]() -> int {
        if (p) SALVA_HIP_CHECK(hipHostUnregister(p));
        return SALVA_HIP_OK;
    });
}

int salva_hip_clear_boundary_sampling(SalvaHipWorld* world, uint32_t slot) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->clear_boundary_sampling(slot);
        return SALVA_HIP_OK;
    });
}

int salva_hip_get_boundary_sources(SalvaHipWorld* world, uint32_t slot, uint32_t* fluid_slots, uint32_t* indices) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->get_boundary_sources(slot, fluid_slots, indices);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_force_callback(SalvaHipWorld* world, SalvaHipForceCallback cb, void* user) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->set_force_callback(cb, user, world);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_coupling_callback(SalvaHipWorld* world, SalvaHipCouplingCallback cb, void* user) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->set_coupling_callback(cb, user, world);
        return SALVA_HIP_OK;
    });
}

int salva_hip_force_get_state(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz, float* densities) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->force_get_state(slot, positions_xyz, velocities_xyz, densities);
        return SALVA_HIP_OK;
    });
}

int salva_hip_force_add_accelerations(SalvaHipWorld* world, uint32_t slot, const float* accelerations_xyz) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->force_add_accelerations(slot, accelerations_xyz);
        return SALVA_HIP_OK;
    });
}

int salva_hip_set_fluid_field(SalvaHipWorld* world, uint32_t slot, int32_t field, const float* data) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->set_fluid_field(slot, field, data);
        return SALVA_HIP_OK;
    });
}

int salva_hip_get_timestep(const SalvaHipWorld* world, float* dt, float* inv_dt) { WorldLock _lk(world);
    if (!world) return SALVA_HIP_E_INVALID;
    world->w->get_timestep(dt, inv_dt);
    return SALVA_HIP_OK;
}

int salva_hip_set_timestep(SalvaHipWorld* world, float dt, float inv_dt) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->set_timestep(dt, inv_dt);
        return SALVA_HIP_OK;
    });
}

int salva_hip_get_boundary_particles(SalvaHipWorld* world, uint32_t slot, float* positions_xyz, float* velocities_xyz) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->get_boundary_particles(slot, positions_xyz, velocities_xyz);
        return SALVA_HIP_OK;
    });
}

int salva_hip_get_boundary_wrench(SalvaHipWorld* world, uint32_t slot, const float point[3], float force[3], float torque[3]) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        if (!point || !force || !torque) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        world->w->get_boundary_wrench(slot, point, force, torque);
        return SALVA_HIP_OK;
    });
}

int salva_hip_clear_boundary_forces(SalvaHipWorld* world, uint32_t slot) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->clear_boundary_forces(slot);
        return SALVA_HIP_OK;
    });
}

uint64_t salva_hip_device_bytes(const SalvaHipWorld* world) { WorldLock _lk(world); return world ? world->w->device_bytes() : 0; }

float salva_hip_time_pred_density(SalvaHipWorld* world, int32_t reps) { WorldLock _lk(world);
    float us = -1.0f;
    int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        us = world->w->time_pred_density(reps);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? us : (float)rc;
}

float salva_hip_time_kernel(SalvaHipWorld* world, int32_t kernel, int32_t reps) { WorldLock _lk(world);
    float us = -1.0f;
    int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        us = world->w->time_kernel(kernel, reps);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? us : (float)rc;
}

int salva_hip_enable_counters(SalvaHipWorld* world, int32_t enabled) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->set_timers(enabled != 0);
        return SALVA_HIP_OK;
    });
}
int salva_hip_set_cfl(SalvaHipWorld* world, int32_t mode, float cfl_coeff, int32_t min_num_substeps, int32_t max_num_substeps) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->set_cfl(mode, cfl_coeff, min_num_substeps, max_num_substeps);
        return SALVA_HIP_OK;
    });
}
int64_t salva_hip_get_substeps(const SalvaHipWorld* world, float* out, uint64_t capacity) { WorldLock _lk(world);
    int64_t n = 0;
    int rc = guarded([&]() -> int {
        if (!world || (!out && capacity)) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        const auto& v = world->w->last_substeps();
        for (size_t k = 0; k < v.size() && k < capacity; ++k) out[k] = v[k];
        n = (int64_t)v.size();
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? n : (int64_t)rc;
}
int salva_hip_get_counters(const SalvaHipWorld* world, SalvaHipCounters* out) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world || !out) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        *out = world->w->counters;
        return SALVA_HIP_OK;
    });
}

#ifdef SALVA_HIP_DIAG
// kernel experiments (declared in diag/salva_hip_diag.h; `make VARIANT=diag` only — not an entry point of libsalva_hip.so)
float salva_hip_time_variant(SalvaHipWorld* world, int32_t variant, uint32_t param, int32_t reps, uint64_t* checksum) { WorldLock _lk(world);
    float us = -1.0f;
    int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        us = world->w->time_variant(variant, param, reps, checksum);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? us : (float)rc;
}
#endif

// ---- multi-GPU (x-slab decomposition) -------------------------------------------------------------------------
struct SalvaHipComm {
    std::shared_ptr<salva::LoopbackShared> group;  // loopback only
    salva::Transport* t = nullptr;
};

int salva_hip_comm_rccl_unique_id(unsigned char* out128) {
    return guarded([&]() -> int {
        if (!out128) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        salva::rccl_unique_id(out128);
        return SALVA_HIP_OK;
    });
}
int salva_hip_comm_rccl_create(int32_t rank, int32_t size, const unsigned char* id128, int32_t device, SalvaHipComm** out) {
    return guarded([&]() -> int {
        if (!id128 || !out) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        auto* c = new SalvaHipComm();
        c->t = salva::rccl_transport(rank, size, id128, device);
        *out = c;
        return SALVA_HIP_OK;
    });
}
struct SalvaHipPeerSetup {
    salva::PeerSetup* s = nullptr;
};
int salva_hip_comm_peer_begin(int32_t rank, int32_t size, int32_t device, uint64_t slot_bytes, unsigned char* handle64,
                              SalvaHipPeerSetup** out) {
    return guarded([&]() -> int {
        if (!handle64 || !out) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        static_assert(SALVA_HIP_PEER_HANDLE_BYTES == salva::PEER_HANDLE_BYTES, "handle size");
        auto* p = new SalvaHipPeerSetup();
        try {
            p->s = salva::peer_begin(rank, size, device, (size_t)slot_bytes, handle64);
        } catch (...) {
            delete p;
            throw;
        }
        *out = p;
        return SALVA_HIP_OK;
    });
}
int salva_hip_comm_peer_connect(SalvaHipPeerSetup* setup, const unsigned char* handles, SalvaHipComm** out) {
    return guarded([&]() -> int {
        if (!setup || !handles || !out) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        salva::PeerSetup* s = setup->s;
        delete setup;  // consumed either way: peer_transport owns `s` from its first line on
        auto* c = new SalvaHipComm();
        try {
            c->t = salva::peer_transport(s, handles);
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
        return SALVA_HIP_OK;
    });
}
void salva_hip_comm_peer_abort(SalvaHipPeerSetup* setup) {
    if (!setup) return;
    try { salva::peer_abort(setup->s); } catch (...) {}
    delete setup;
}
int salva_hip_comm_loopback_create(int32_t size, SalvaHipComm** out_ranks) {
    return guarded([&]() -> int {
        if (size < 1 || !out_ranks) throw salva::HipError(SALVA_HIP_E_INVALID, "bad argument");
        auto g = salva::loopback_create(size);
        for (int r = 0; r < size; ++r) {
            auto* c = new SalvaHipComm();
            c->group = g;
            c->t = salva::loopback_transport(g, r);
            out_ranks[r] = c;
        }
        return SALVA_HIP_OK;
    });
}
void salva_hip_comm_destroy(SalvaHipComm* comm) {
    if (!comm) return;
    try { delete comm->t; } catch (...) {}
    delete comm;
}
int salva_hip_comm_selftest(SalvaHipComm* comm, uint64_t max_bytes, int32_t rounds) {
    return guarded([&]() -> int {
        if (!comm || !comm->t) throw salva::HipError(SALVA_HIP_E_INVALID, "null communicator");
        if (comm->t->device() >= 0) SALVA_HIP_CHECK(hipSetDevice(comm->t->device()));  // (a stream belongs to the current device)
        hipStream_t s = nullptr;
        SALVA_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        try {
            salva::transport_selftest(*comm->t, (size_t)max_bytes, rounds, s);
        } catch (...) {
            (void)hipStreamDestroy(s);
            throw;
        }
        SALVA_HIP_CHECK(hipStreamDestroy(s));
        return SALVA_HIP_OK;
    });
}
int salva_hip_comm_time(SalvaHipComm* comm, uint64_t bytes, int32_t iters, float* us_exchange, float* us_allreduce) {
    return guarded([&]() -> int {
        if (!comm || !comm->t) throw salva::HipError(SALVA_HIP_E_INVALID, "null communicator");
        if (comm->t->device() >= 0) SALVA_HIP_CHECK(hipSetDevice(comm->t->device()));  // (a stream belongs to the current device)
        hipStream_t s = nullptr;
        SALVA_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        try {
            salva::transport_time(*comm->t, (size_t)bytes, iters, us_exchange, us_allreduce, s);
        } catch (...) {
            (void)hipStreamDestroy(s);
            throw;
        }
        SALVA_HIP_CHECK(hipStreamDestroy(s));
        return SALVA_HIP_OK;
    });
}
int salva_hip_set_domain(SalvaHipWorld* world, SalvaHipComm* comm, int32_t cell_lo, int32_t cell_hi, uint32_t gid_offset) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world || !comm) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        not_in_force_callback(world);
        world->w->set_domain(comm->t, cell_lo, cell_hi, gid_offset);
        return SALVA_HIP_OK;
    });
}
int salva_hip_rebalance(SalvaHipWorld* world, int32_t* cell_lo, int32_t* cell_hi) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->rebalance(cell_lo, cell_hi);
        return SALVA_HIP_OK;
    });
}
int64_t salva_hip_get_owned(SalvaHipWorld* world, uint32_t capacity, uint32_t* gids, float* positions_xyz, float* velocities_xyz,
                            uint32_t* fluid_slots) { WorldLock _lk(world);
    int64_t count = 0;
    const int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        count = (int64_t)world->w->get_owned(capacity, gids, positions_xyz, velocities_xyz, fluid_slots);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? count : (int64_t)rc;
}

int64_t salva_hip_delete_owned(SalvaHipWorld* world, uint32_t n, const uint32_t* gids) { WorldLock _lk(world);
    int64_t count = 0;
    const int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        count = (int64_t)world->w->delete_owned(n, gids);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? count : (int64_t)rc;
}

int64_t salva_hip_particles_intersecting_aabb(SalvaHipWorld* world, const float mins[3], const float maxs[3], uint64_t capacity,
                                             uint32_t* kinds, uint32_t* slots, uint32_t* indices) { WorldLock _lk(world);
    int64_t total = 0;
    const int rc = guarded([&]() -> int {
        if (!world || !mins || !maxs) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        total = (int64_t)world->w->particles_in_aabb(mins, maxs, capacity, kinds, slots, indices);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? total : (int64_t)rc;
}
int64_t salva_hip_particles_intersecting_shape(SalvaHipWorld* world, const float translation[3], const float rotation_ijkw[4],
                                              const SalvaHipShape* shape, uint64_t capacity, uint32_t* kinds, uint32_t* slots,
                                              uint32_t* indices) { WorldLock _lk(world);
    int64_t total = 0;
    const int rc = guarded([&]() -> int {
        if (!world || !translation || !rotation_ijkw || !shape) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        total = (int64_t)world->w->particles_in_shape(translation, rotation_ijkw, *shape, capacity, kinds, slots, indices);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? total : (int64_t)rc;
}
int64_t salva_hip_particles_intersecting_host_shape(SalvaHipWorld* world, const SalvaHipHostQueryShape* shape, uint64_t capacity,
                                                    uint32_t* kinds, uint32_t* slots, uint32_t* indices) { WorldLock _lk(world);
    int64_t total = 0;
    const int rc = guarded([&]() -> int {
        if (!world || !shape) throw salva::HipError(SALVA_HIP_E_INVALID, "null argument");
        not_in_force_callback(world);
        total = (int64_t)world->w->particles_in_host_shape(*shape, capacity, kinds, slots, indices);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? total : (int64_t)rc;
}
int salva_hip_add_particles(SalvaHipWorld* world, uint32_t slot, uint64_t n_add, const float* positions_xyz, const float* velocities_xyz) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        world->w->add_particles(slot, n_add, positions_xyz, velocities_xyz);
        return SALVA_HIP_OK;
    });
}
int64_t salva_hip_delete_particles(SalvaHipWorld* world, uint32_t slot, const uint8_t* deleted_mask) { WorldLock _lk(world);
    int64_t kept = 0;
    const int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        not_in_force_callback(world);
        kept = (int64_t)world->w->delete_particles(slot, deleted_mask);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? kept : (int64_t)rc;
}

int64_t salva_hip_get_fluid_contacts(SalvaHipWorld* world, uint32_t slot, int32_t boundary_contacts, uint64_t* offsets,
                                     uint32_t* j_model, uint32_t* j, uint64_t capacity) { WorldLock _lk(world);
    int64_t total = 0;
    const int rc = guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        total = (int64_t)world->w->get_fluid_contacts(slot, boundary_contacts, offsets, j_model, j, capacity);
        return SALVA_HIP_OK;
    });
    return rc == SALVA_HIP_OK ? total : (int64_t)rc;
}

int salva_hip_get_force_stats(SalvaHipWorld* world, uint32_t slot, uint32_t force, int32_t* iters, float* error) { WorldLock _lk(world);
    return guarded([&]() -> int {
        if (!world) throw salva::HipError(SALVA_HIP_E_INVALID, "null world");
        world->w->get_force_stats(slot, force, iters, error);
        return SALVA_HIP_OK;
    });
}

const char* salva_hip_last_error(void) { return g_last_error.c_str(); }
const char* salva_hip_version(void) { return "salva_hip 0.1 (gfx950)"; }

}  // extern "C"
