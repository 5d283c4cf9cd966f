// world.hip — host side of libsalva_hip: device memory, the per-step launch sequence, up/download.
//
// Restates the control flow of /root/reference/src/liquid_world.rs:67-158 (step_with_coupling with the no-op
// `()` coupling manager), src/solver/pressure/dfsph_solver.rs:667-708 (DFSPH step, iteration protocol :432-503)
// and src/solver/pressure/iisph_solver.rs:643-711 on top of the kernels in grid/dfsph/iisph/forces.hip.
// Host <-> device traffic happens only in set_* / get_*; `step` works on HBM-resident state and reads back a
// few scalars (convergence errors, list sizes, the next cell bounding box).
#include "world.h"
#include "dcs.h"
#include "bbox.h"

#include <algorithm>
#include <cfloat>
#include <cstddef>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

#include <mutex>
#include <unordered_map>

namespace salva {

void raise_tile_lds_limit(const void* kernel, uint32_t bytes) {
    if (bytes > 160u * 1024u)
        throw HipError(SALVA_HIP_E_CAPACITY, "a tile's halo does not fit the 160 KiB LDS (particles are compressed far beyond rest density)");
    static std::mutex mu;
    static std::unordered_map<const void*, uint32_t> granted;
    std::lock_guard<std::mutex> lock(mu);
    uint32_t& g = granted[kernel];
    if (bytes <= g) return;
    const uint32_t want = std::min<uint32_t>(160u * 1024u, std::max<uint32_t>(bytes + bytes / 4, 64u * 1024u));
    SALVA_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
    g = want;
}

// ------------------------------------------------------------------------------------------------ small helpers
__global__ void k_pack_xyz(uint32_t n, const float* __restrict__ src, float4* __restrict__ dst, int keep_w, float wval) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 d = dst[i];
    d.x = src[3 * i]; d.y = src[3 * i + 1]; d.z = src[3 * i + 2];
    if (!keep_w) d.w = wval;
    dst[i] = d;
}
__global__ void k_pack_w(uint32_t n, const float* __restrict__ src, float4* __restrict__ dst) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) reinterpret_cast<float*>(&dst[i])[3] = src[i];
}
__global__ void k_fill_f4(uint32_t n, float4* __restrict__ dst, float4 v, int keep_w) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (keep_w) v.w = dst[i].w;
    dst[i] = v;
}
__global__ void k_fill_w(uint32_t n, float4* __restrict__ dst, float v) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) reinterpret_cast<float*>(&dst[i])[3] = v;
}
__global__ void k_fill_u32(uint32_t n, uint32_t* __restrict__ dst, uint32_t v) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) dst[i] = v;
}
__global__ void k_unpack_xyz(uint32_t n, const float4* __restrict__ src, float* __restrict__ dst) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 s = src[i];
    dst[3 * i] = s.x; dst[3 * i + 1] = s.y; dst[3 * i + 2] = s.z;
}
__global__ void k_unpack_w(uint32_t n, const float4* __restrict__ src, float* __restrict__ dst) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i].w;
}
__global__ void k_set_bmodel(uint32_t n, float4* __restrict__ bvel, uint32_t m) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) reinterpret_cast<float*>(&bvel[i])[3] = __uint_as_float(m);
}

static inline unsigned nblk(uint64_t n) { return div_up(n ? n : 1, BLOCK); }

struct Piece { uint64_t src_off, len; bool from_old; };

template <typename T>
static void rebuild(DevBuf<T>& buf, const std::vector<Piece>& pieces, uint64_t new_total, hipStream_t s) {
    DevBuf<T> nb;
    nb.ensure(new_total ? new_total : 1);
    uint64_t off = 0;
    for (const Piece& p : pieces) {
        if (p.from_old && p.len)
            SALVA_HIP_CHECK(hipMemcpyAsync(nb.p + off, buf.p + p.src_off, p.len * sizeof(T), hipMemcpyDeviceToDevice, s));
        off += p.len;
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(s));
    std::swap(buf.p, nb.p);
    std::swap(buf.cap, nb.cap);
}

static int bits_for(uint64_t ncells) {
    int b = 1;
    while (b < 32 && ((uint64_t)1 << b) < ncells) ++b;
    return b;
}

// ------------------------------------------------------------------------------------------------ ctor / dtor
World::World(const SalvaHipParams& p) : prm(p) {
    if (!(p.particle_radius > 0.0f) || !(p.smoothing_factor > 0.0f))
        throw HipError(SALVA_HIP_E_INVALID, "particle_radius and smoothing_factor must be positive");
    if (p.solver != SALVA_HIP_SOLVER_DFSPH && p.solver != SALVA_HIP_SOLVER_IISPH)
        throw HipError(SALVA_HIP_E_INVALID, "unknown solver kind");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        throw HipError(SALVA_HIP_E_HIP, "no HIP device available: libsalva_hip has no CPU fallback");
    if (p.device < 0 || p.device >= ndev) throw HipError(SALVA_HIP_E_INVALID, "device ordinal out of range");
    use_device();
    // h = particle_radius * smoothing_factor * 2 (liquid_world.rs:44)
    const float h = p.particle_radius * p.smoothing_factor * 2.0f;
    sc = make_sph_consts(h);
    if (p.kernel_density < 0 || p.kernel_density > SALVA_HIP_KERNEL_VISCOSITY || p.kernel_gradient < 0 || p.kernel_gradient > SALVA_HIP_KERNEL_VISCOSITY)
        throw HipError(SALVA_HIP_E_INVALID, "unknown kernel kind");
    sc.kd = p.kernel_density;
    sc.kg = p.kernel_gradient;
    SALVA_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    SALVA_HIP_CHECK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
    SALVA_HIP_CHECK(hipEventCreateWithFlags(&ev_pre_refresh, hipEventDisableTiming));
    SALVA_HIP_CHECK(hipEventCreateWithFlags(&ev_interior, hipEventDisableTiming));
    SALVA_HIP_CHECK(hipEventCreateWithFlags(&ev_spec_eval, hipEventDisableTiming));
    SALVA_HIP_CHECK(hipEventCreateWithFlags(&ev_spec_apply, hipEventDisableTiming));
    spec_dist_off = getenv("SALVA_HIP_NO_SPEC_DIST") != nullptr;
    overlap_exchange = getenv("SALVA_HIP_NO_OVERLAP") == nullptr;
    // Speculative sizing is OFF unless asked for (SALVA_HIP_SPECULATE=1).  Measured on the bench scene (10^6 particles): it
    // removes two ~20 us host round trips from a ~0.9 ms free-fall step (-2 %), but a failed prediction costs a whole extra
    // step, and at the impact — where the halo and the lists grow for a dozen steps in a row — two passes in twenty were
    // discarded, 2.25 ms per step against 1.88 ms without speculation.  Kept as an option for steady flows.
    spec_off = getenv("SALVA_HIP_SPECULATE") == nullptr || getenv("SALVA_HIP_NO_SPECULATION") != nullptr;
    spec_tight = getenv("SALVA_HIP_SPEC_TIGHT") != nullptr;
    defer_off = getenv("SALVA_HIP_NO_DEFER_LISTS") != nullptr;
    spec_apply_off = getenv("SALVA_HIP_NO_SPEC_APPLY") != nullptr;
    chain_off = getenv("SALVA_HIP_NO_CHAIN") != nullptr;
    pre_off = getenv("SALVA_HIP_NO_PREGRID") != nullptr;
    split_off = getenv("SALVA_HIP_NO_SPLIT") != nullptr;
    classes_off = getenv("SALVA_HIP_NO_CLASSES") != nullptr;
    classes_forced = getenv("SALVA_HIP_CLASSES") != nullptr;
    light_on = getenv("SALVA_HIP_LIGHT") != nullptr;
    if (const char* e = getenv("SALVA_HIP_SPLIT_S")) split_forced = (uint32_t)std::max(atoi(e), 1);
    no_planes = getenv("SALVA_HIP_NO_PLANES") != nullptr;
    two_mass_off = getenv("SALVA_HIP_NO_TWO_MASS") != nullptr;
    if (const char* e = getenv("SALVA_HIP_MAX_MASSES")) max_masses = (uint32_t)std::min(std::max(atoi(e), 2), 4);  // (2 = round 5's rule, the default)
    fold_off = getenv("SALVA_HIP_NO_FOLD") != nullptr;
    if (const char* e = getenv("SALVA_HIP_FOLD_CELLS")) {
        const long v = atol(e);
        if (v >= 8 && v <= (1 << 20) && (v & (v - 1)) == 0) fold_forced = (uint32_t)v;
    }
    tile_trace = getenv("SALVA_HIP_TILE_TRACE") != nullptr;
    no_fused_div = getenv("SALVA_HIP_NO_FUSED_DIV") != nullptr;
    if (const char* e = getenv("SALVA_HIP_RADIX_SORT")) sort_mode = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("SALVA_HIP_DS_LEVEL")) lds.ds_level = (uint32_t)std::max(0, atoi(e));  // (tests: pairs.h pick_ds*)
#ifdef SALVA_HIP_DIAG
    if (const char* e = getenv("SALVA_HIP_SCHED")) sched_mode = atoi(e);
#endif
    // (tests: force an overflow — the given capacity also counts as checked, so that the very first step takes the deferred path)
    if (const char* e = getenv("SALVA_HIP_LIST_CAP0")) { cap_ff = std::max<uint32_t>(LIST_REGS, ((uint32_t)atoi(e) + 3u) & ~3u); trust_cap0 = true; }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p.device) == hipSuccess && cus > 0) num_cus = cus;
    }
    SALVA_HIP_CHECK(hipHostMalloc((void**)&h_rb, sizeof(Readback), hipHostMallocDefault));
    memset(h_rb, 0, sizeof(Readback));
    SALVA_HIP_CHECK(hipHostMalloc((void**)&h_ctl, 2 * NUM_SOLVES * sizeof(SolveCtl), hipHostMallocDefault));
    memset(h_ctl, 0, 2 * NUM_SOLVES * sizeof(SolveCtl));
    SALVA_HIP_CHECK(hipHostMalloc((void**)&h_pub, NUM_SOLVES * sizeof(SolveCtl), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h_pub, 0, NUM_SOLVES * sizeof(SolveCtl));
    SALVA_HIP_CHECK(hipHostMalloc((void**)&h_hostpub, sizeof(HostPub), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h_hostpub, 0, sizeof(HostPub));
    d_ctl.ensure(NUM_SOLVES);
    d_rb.ensure(1);
    d_flags.p = &d_rb.p->flags;
    SALVA_HIP_CHECK(hipMemset(d_rb.p, 0, sizeof(Readback)));
    {
        mass_slots.ensure(2 * MASS_SLOTS);  // (k_cell_keys' "a mass differs" flags + the reference bits, grid.hip)
        SALVA_HIP_CHECK(hipMemset(mass_slots.p, 0, 2 * MASS_SLOTS * sizeof(uint32_t)));
    }
    d_counters.ensure(4);
    for (auto& e2 : ev) SALVA_HIP_CHECK(hipEventCreate(&e2));
    for (auto& e2 : evc) SALVA_HIP_CHECK(hipEventCreate(&e2));
    SALVA_HIP_CHECK(hipEventCreateWithFlags(&ev_sync, hipEventDisableTiming));
}

World::~World() {
    (void)hipSetDevice(prm.device);
    if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    if (stream2) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
    if (dl_stream) { (void)hipStreamSynchronize(dl_stream); (void)hipStreamDestroy(dl_stream); }
    if (ev_dl_ready) (void)hipEventDestroy(ev_dl_ready);
    if (ev_dl_done) (void)hipEventDestroy(ev_dl_done);
    for (float* p : h_dl) if (p) (void)hipHostFree(p);
    for (hipEvent_t e : dist_ev) if (e) (void)hipEventDestroy(e);
    if (ev_pre_refresh) (void)hipEventDestroy(ev_pre_refresh);
    if (ev_interior) (void)hipEventDestroy(ev_interior);
    if (ev_spec_eval) (void)hipEventDestroy(ev_spec_eval);
    if (ev_spec_apply) (void)hipEventDestroy(ev_spec_apply);
    if (h_rb) (void)hipHostFree(h_rb);
    if (h_ctl) (void)hipHostFree(h_ctl);
    if (h_pub) (void)hipHostFree(h_pub);
    if (h_hostpub) (void)hipHostFree(h_hostpub);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : evc) if (e) (void)hipEventDestroy(e);
    if (ev_sync) (void)hipEventDestroy(ev_sync);
}

void World::use_device() const { SALVA_HIP_CHECK(hipSetDevice(prm.device)); }

uint64_t World::fluid_offset(uint32_t slot) const {
    uint64_t o = 0;
    for (uint32_t s = 0; s < slot; ++s) o += fluids[s].n;
    return o;
}
uint64_t World::boundary_offset(uint32_t slot) const {
    uint64_t o = 0;
    for (uint32_t s = 0; s < slot; ++s) o += bounds[s].n;
    return o;
}

FluidArrays World::arrays(int which) {
    FluidArrays a;
    a.posm = posm[which].p; a.vel = vel[which].p; a.dv = dv[which].p; a.model = model[which].p; a.perm = perm[which].p;
    a.gtag = comm ? gtag[which].p : nullptr;
    return a;
}

void World::ensure_cub_temp(size_t bytes) { cub_temp.ensure(bytes ? bytes : 1, stream, false, 1.25f); }

// Bring the canonical (host-order) staging arrays up to date with the sorted working set.
void World::ensure_staging_current() {
    if (staging_current) return;
    if (comm && dist_started)
        throw HipError(SALVA_HIP_E_INVALID, "host-order fluid arrays are not maintained in a multi-GPU run: use salva_hip_get_owned");
    if (sorted_valid && n) {
        launch_sorted_to_stage(n, arrays(cur), st_pos.p, st_vel.p, st_dv.p, stream);
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    staging_current = true;
}

// ------------------------------------------------------------------------------------------------ fluids
void World::set_fluid(uint32_t slot, uint64_t nn, const float* pos, const float* vel_h, const float* vol,
                      const float* acc_h, const float* dvs, float density0, uint32_t memberships, uint32_t filter,
                      uint32_t dirty) {
    use_device();
    if (slot > fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range (slots are dense)");
    if (slot >= (uint32_t)MAX_MODELS) throw HipError(SALVA_HIP_E_CAPACITY, "at most 32 fluids per world");
    const bool is_new = slot == fluids.size();
    if ((uint64_t)n - (is_new ? 0 : fluids[slot].n) + nn >= 0xfffffff0ull)
        throw HipError(SALVA_HIP_E_CAPACITY, "more than 2^32 fluid particles on one device");
    ensure_staging_current();
    const uint64_t old_n = is_new ? 0 : fluids[slot].n;
    const bool resized = is_new || old_n != nn;
    if (resized) {
        if (nn && !pos) throw HipError(SALVA_HIP_E_INVALID, "positions are required when a fluid is created or resized");
        dirty = SALVA_HIP_DIRTY_ALL;
    }
    if (is_new) fluids.emplace_back();
    const uint64_t off = fluid_offset(slot);
    const uint64_t old_total = n;
    const uint64_t new_total = old_total - old_n + nn;
    if (resized) {
        std::vector<Piece> pieces = {{0, off, true}, {0, nn, false}, {off + old_n, old_total - off - old_n, true}};
        rebuild(st_pos, pieces, new_total, stream);
        rebuild(st_vel, pieces, new_total, stream);
        rebuild(st_dv, pieces, new_total, stream);
        rebuild(st_acc, pieces, new_total, stream);
        rebuild(st_model, pieces, new_total, stream);
        n = (uint32_t)new_total;
        if (nn) {
            // defaults: zero velocity / velocity change / acceleration, Fluid::particle_volume (fluid.rs:110-120)
            const float r = prm.particle_radius;
            const float default_vol = r * r * r * 6.4f;  // 8 * 0.8
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_pos.p + off, make_float4(0, 0, 0, default_vol), 0);
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_vel.p + off, make_float4(0, 0, 0, 0), 0);
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_dv.p + off, make_float4(0, 0, 0, 0), 0);
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_acc.p + off, make_float4(0, 0, 0, 0), 0);
        }
        if (is_new && !dvs) {
            auto it = sticky.find(slot);
            if (it != sticky.end()) {  // buffer[slot] of a fluid removed since the last step: inherited (world.h `sticky`)
                const uint64_t inherit = std::min<uint64_t>(it->second.len, nn);
                if (inherit) SALVA_HIP_CHECK(hipMemcpyAsync(st_dv.p + off, it->second.data->p, inherit * sizeof(float4), hipMemcpyDeviceToDevice, stream));
                if (it->second.len <= nn) sticky.erase(it);
            }
        }
        // model ids of every slot at/after this one may have moved
        fluids[slot].n = nn;
        uint64_t o = 0;
        for (uint32_t s = 0; s < fluids.size(); ++s) {
            if (fluids[s].n) k_fill_u32<<<nblk(fluids[s].n), BLOCK, 0, stream>>>((uint32_t)fluids[s].n, st_model.p + o, s);
            o += fluids[s].n;
        }
    }
    FluidSlot& f = fluids[slot];
    f.density0 = density0; f.memberships = memberships; f.filter = filter;
    {   // the fluid's one volume, if it has one (FluidSlot::vol_uniform): a resize filled the default in, an upload may change it
        const float r = prm.particle_radius;
        if (resized) f.vol_uniform = r * r * r * 6.4f;
        if (nn && (dirty & SALVA_HIP_DIRTY_VOLUMES) && vol) {
            bool same = true;
            for (uint64_t k = 1; k < nn && same; ++k) same = vol[k] == vol[0];
            f.vol_uniform = same ? vol[0] : std::numeric_limits<float>::quiet_NaN();
        }
    }
    auto upload3 = [&](const float* src, float4* dst, int keep_w) {
        scratch_f.ensure(3 * nn, stream, false, 1.1f);
        SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, src, 3 * nn * sizeof(float), hipMemcpyHostToDevice, stream));
        k_pack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f.p, dst, keep_w, 0.0f);
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    };
    if (nn) {
        if ((dirty & SALVA_HIP_DIRTY_POSITIONS) && pos) upload3(pos, st_pos.p + off, 1);
        if ((dirty & SALVA_HIP_DIRTY_VELOCITIES) && vel_h) upload3(vel_h, st_vel.p + off, 1);
        if ((dirty & SALVA_HIP_DIRTY_VOLUMES) && vol) {
            scratch_f.ensure(nn, stream, false, 1.1f);
            SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, vol, nn * sizeof(float), hipMemcpyHostToDevice, stream));
            k_pack_w<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f.p, st_pos.p + off);
            SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        }
        if ((dirty & SALVA_HIP_DIRTY_ACCELERATIONS) && acc_h) {
            // accelerations are zero between steps; the first user upload must not resurrect stale staging values
            if (!acc_user) SALVA_HIP_CHECK(hipMemsetAsync(st_acc.p, 0, (size_t)n * sizeof(float4), stream));
            upload3(acc_h, st_acc.p + off, 1);
            acc_user = true;
        }
        if (dvs) upload3(dvs, st_dv.p + off, 1);
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    sorted_valid = false;
    bbox_known = false;
    tables_dirty = true;
    have_last_ctx = false;
}

// Fluid::add_particles (fluid.rs:126-150): append to the fluid's arrays — default volume, zero acceleration, zero
// velocity change (the solver's buffers grow with zeros, dfsph_solver.rs:526-549) — without re-uploading the rest.
void World::add_particles(uint32_t slot, uint64_t n_add, const float* pos, const float* vel_h) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    if (comm && dist_started) { dist_add_particles(slot, n_add, pos, vel_h); return; }  // (collective, world_dist.hip)
    if (n_add == 0) return;
    if (!pos) throw HipError(SALVA_HIP_E_INVALID, "positions are required");
    if ((uint64_t)n + n_add >= 0xfffffff0ull) throw HipError(SALVA_HIP_E_CAPACITY, "more than 2^32 fluid particles on one device");
    ensure_staging_current();
    const uint64_t old_n = fluids[slot].n, off = fluid_offset(slot), at = off + old_n, old_total = n, new_total = old_total + n_add;
    std::vector<Piece> pieces = {{0, at, true}, {0, n_add, false}, {at, old_total - at, true}};
    rebuild(st_pos, pieces, new_total, stream);
    rebuild(st_vel, pieces, new_total, stream);
    rebuild(st_dv, pieces, new_total, stream);
    rebuild(st_acc, pieces, new_total, stream);
    rebuild(st_model, pieces, new_total, stream);
    n = (uint32_t)new_total;
    fluids[slot].n = old_n + n_add;
    {   // (the new particles have the default volume: the fluid keeps ONE volume only if that is the one it had)
        const float r = prm.particle_radius, default_vol = r * r * r * 6.4f;
        FluidSlot& f = fluids[slot];
        if (old_n == 0) f.vol_uniform = default_vol;
        else if (!(f.vol_uniform == default_vol)) f.vol_uniform = std::numeric_limits<float>::quiet_NaN();
    }
    const float r = prm.particle_radius;
    k_fill_f4<<<nblk(n_add), BLOCK, 0, stream>>>((uint32_t)n_add, st_pos.p + at, make_float4(0, 0, 0, r * r * r * 6.4f), 0);
    k_fill_f4<<<nblk(n_add), BLOCK, 0, stream>>>((uint32_t)n_add, st_vel.p + at, make_float4(0, 0, 0, 0), 0);
    k_fill_f4<<<nblk(n_add), BLOCK, 0, stream>>>((uint32_t)n_add, st_dv.p + at, make_float4(0, 0, 0, 0), 0);
    k_fill_f4<<<nblk(n_add), BLOCK, 0, stream>>>((uint32_t)n_add, st_acc.p + at, make_float4(0, 0, 0, 0), 0);
    {
        auto it = sticky.find(slot);
        if (it != sticky.end() && it->second.len > old_n) {  // `velocity_changes[slot].resize(n)` exposes the inherited tail
            const uint64_t take = std::min<uint64_t>(it->second.len - old_n, n_add);
            SALVA_HIP_CHECK(hipMemcpyAsync(st_dv.p + at, it->second.data->p + old_n, take * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        }
    }
    uint64_t o = 0;
    for (uint32_t s = 0; s < fluids.size(); ++s) {
        if (fluids[s].n) k_fill_u32<<<nblk(fluids[s].n), BLOCK, 0, stream>>>((uint32_t)fluids[s].n, st_model.p + o, s);
        o += fluids[s].n;
    }
    scratch_f.ensure(3 * n_add, stream, false, 1.1f);
    SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, pos, 3 * n_add * sizeof(float), hipMemcpyHostToDevice, stream));
    k_pack_xyz<<<nblk(n_add), BLOCK, 0, stream>>>((uint32_t)n_add, scratch_f.p, st_pos.p + at, 1, 0.0f);
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    if (vel_h) {
        SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, vel_h, 3 * n_add * sizeof(float), hipMemcpyHostToDevice, stream));
        k_pack_xyz<<<nblk(n_add), BLOCK, 0, stream>>>((uint32_t)n_add, scratch_f.p, st_vel.p + at, 1, 0.0f);
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    sorted_valid = false; bbox_known = false; tables_dirty = true; have_last_ctx = false;
}

// Fluid::apply_particles_removal (fluid.rs:88-98) together with the compaction of the solver's velocity_changes
// (dfsph_solver.rs:550-560 / iisph_solver.rs:505-537): a stable compaction (`filter_from_mask`, helper.rs:4-12) of every
// per-particle array of the fluid, on the device.  `mask[i] != 0` deletes particle i.  Returns the remaining count.
uint64_t World::delete_particles(uint32_t slot, const uint8_t* mask) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    if (comm && dist_started) throw HipError(SALVA_HIP_E_INVALID, "host indices do not exist in a running multi-GPU world: delete by global id (salva_hip_delete_owned)");
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    if (nn == 0 || !mask) return nn;
    ensure_staging_current();
    // keep flags over the whole staging range (other fluids are kept)
    std::vector<uint8_t> keep((size_t)n, 1);
    uint64_t kept = 0;
    for (uint64_t k = 0; k < nn; ++k) { keep[off + k] = mask[k] ? 0 : 1; kept += keep[off + k]; }
    if (kept == nn) return nn;
    DevBuf<uint8_t> d_keep;
    DevBuf<uint32_t> d_num;
    d_keep.ensure(n); d_num.ensure(1);
    SALVA_HIP_CHECK(hipMemcpyAsync(d_keep.p, keep.data(), (size_t)n, hipMemcpyHostToDevice, stream));
    const uint64_t new_total = (uint64_t)n - (nn - kept);
    auto compact = [&](DevBuf<float4>& buf) {
        DevBuf<float4> out;
        out.ensure(std::max<uint64_t>(new_total, 1));
        const size_t tb = select_flagged_temp_bytes(n);
        ensure_cub_temp(tb);
        select_flagged_f4(cub_temp.p, tb, buf.p, d_keep.p, out.p, d_num.p, n, stream);
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        std::swap(buf.p, out.p);
        std::swap(buf.cap, out.cap);
    };
    compact(st_pos); compact(st_vel); compact(st_dv); compact(st_acc);
    {
        // an inherited solver buffer (world.h `sticky`) is filtered with the same mask by the reference — after its resize, so
        // the entries beyond the fluid's particles (handed to particles added before the next step) just move down
        auto it = sticky.find(slot);
        if (it != sticky.end() && it->second.len) {
            const uint64_t L = it->second.len, head = std::min<uint64_t>(L, nn);
            std::vector<uint8_t> k2((size_t)L, 1);
            uint64_t kept2 = 0;
            for (uint64_t k = 0; k < L; ++k) { if (k < head) k2[k] = mask[k] ? 0 : 1; kept2 += k2[k]; }
            DevBuf<uint8_t> d_k2;
            d_k2.ensure(L);
            SALVA_HIP_CHECK(hipMemcpyAsync(d_k2.p, k2.data(), (size_t)L, hipMemcpyHostToDevice, stream));
            auto out = std::make_shared<DevBuf<float4>>();
            out->ensure(std::max<uint64_t>(kept2, 1));
            const size_t tb = select_flagged_temp_bytes((uint32_t)L);
            ensure_cub_temp(tb);
            select_flagged_f4(cub_temp.p, tb, it->second.data->p, d_k2.p, out->p, d_num.p, (uint32_t)L, stream);
            SALVA_HIP_CHECK(hipStreamSynchronize(stream));
            it->second.data = out;
            it->second.len = kept2;
        }
    }
    n = (uint32_t)new_total;
    fluids[slot].n = kept;
    uint64_t o = 0;
    for (uint32_t s = 0; s < fluids.size(); ++s) {
        if (fluids[s].n) k_fill_u32<<<nblk(fluids[s].n), BLOCK, 0, stream>>>((uint32_t)fluids[s].n, st_model.p + o, s);
        o += fluids[s].n;
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    sorted_valid = false; bbox_known = false; tables_dirty = true; have_last_ctx = false;
    return kept;
}

void World::set_fluid_forces(uint32_t slot, const SalvaHipForceDesc* f, uint32_t nf) {
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    for (uint32_t k = 0; k < nf; ++k)
        if (f[k].kind < SALVA_HIP_FORCE_XSPH || f[k].kind > SALVA_HIP_FORCE_CUSTOM)
            throw HipError(SALVA_HIP_E_INVALID, "unknown non-pressure force kind (only built-ins run on the device)");
    for (uint32_t k = 0; k < nf; ++k)
        if (f[k].kind == SALVA_HIP_FORCE_DFSPH_VISCOSITY && !(f[k].p[0] >= 0.0f && f[k].p[0] <= 1.0f))
            throw HipError(SALVA_HIP_E_INVALID, "The viscosity coefficient must be between 0.0 and 1.0.");  // dfsph_viscosity.rs:104-108
    for (uint32_t k = 0; k < nf; ++k)
        if (f[k].kind == SALVA_HIP_FORCE_WCSPH_TENSION && f[k].p[1] != 0.0f)
            // wcsph_surface_tension.rs:66-83 walks the fluid-fluid contacts while indexing the boundaries: it panics
            throw HipError(SALVA_HIP_E_INVALID, "WCSPHSurfaceTension: a non-zero boundary coefficient panics in the reference (index out of bounds)");
    fluids[slot].forces.assign(f, f + nf);
}

void World::remove_fluid(uint32_t slot) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    ensure_staging_current();
    // ContiguousArena::remove is a swap-remove (contiguous_arena.rs): the last fluid takes the freed slot.
    const uint32_t last = (uint32_t)fluids.size() - 1;
    const uint64_t off = fluid_offset(slot), len = fluids[slot].n;
    const uint64_t loff = fluid_offset(last), llen = fluids[last].n;
    std::vector<Piece> pieces;
    pieces.push_back({0, off, true});
    if (slot != last) {
        pieces.push_back({loff, llen, true});
        pieces.push_back({off + len, loff - off - len, true});
    }
    const uint64_t new_total = n - len;
    rebuild(st_pos, pieces, new_total, stream);
    rebuild(st_vel, pieces, new_total, stream);
    {
        // The solver's per-fluid buffers are positional in the reference and are NOT swapped with the object
        // (liquid_world.rs:171-173 removes from the arena only; init_with_fluids then resizes `velocity_changes[slot]` /
        // `pressures[slot]` to the new occupant's particle count, dfsph_solver.rs:526-549 / iisph_solver.rs:479-501): the
        // fluid that moves into the freed slot inherits the removed fluid's velocity changes and pressures, truncated
        // or zero-extended, and the buffer of the last slot lives on until the next step (a fluid added before then
        // inherits it).  Reproduced: results must match the reference's, quirks included.  See `sticky` in world.h.
        auto content_of = [&](uint32_t sl, uint64_t o, uint64_t l) {
            auto it = sticky.find(sl);
            if (it != sticky.end()) return it->second;
            StickyBuf b;
            b.data = std::make_shared<DevBuf<float4>>();
            b.len = l;
            b.data->ensure(std::max<uint64_t>(l, 1));
            if (l) SALVA_HIP_CHECK(hipMemcpyAsync(b.data->p, st_dv.p + o, l * sizeof(float4), hipMemcpyDeviceToDevice, stream));
            return b;
        };
        const StickyBuf A = content_of(slot, off, len);
        const StickyBuf B = slot != last ? content_of(last, loff, llen) : StickyBuf{};
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<Piece> dvp;
        dvp.push_back({0, off, true});
        if (slot != last) {
            dvp.push_back({0, llen, false});
            dvp.push_back({off + len, loff - off - len, true});
        }
        rebuild(st_dv, dvp, new_total, stream);
        sticky.erase(slot);
        sticky.erase(last);
        if (slot != last) {
            const uint64_t inherit = std::min(A.len, llen);
            if (inherit) SALVA_HIP_CHECK(hipMemcpyAsync(st_dv.p + off, A.data->p, inherit * sizeof(float4), hipMemcpyDeviceToDevice, stream));
            if (llen > inherit) SALVA_HIP_CHECK(hipMemsetAsync(st_dv.p + off + inherit, 0, (llen - inherit) * sizeof(float4), stream));
            if (A.len > llen) sticky[slot] = A;  // its tail goes to particles added before the next step
            sticky[last] = B;
        } else {
            sticky[last] = A;
        }
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    rebuild(st_acc, pieces, new_total, stream);
    rebuild(st_model, pieces, new_total, stream);
    if (slot != last) fluids[slot] = fluids[last];
    fluids.pop_back();
    n = (uint32_t)new_total;
    uint64_t o = 0;
    for (uint32_t s = 0; s < fluids.size(); ++s) {
        if (fluids[s].n) k_fill_u32<<<nblk(fluids[s].n), BLOCK, 0, stream>>>((uint32_t)fluids[s].n, st_model.p + o, s);
        o += fluids[s].n;
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    sorted_valid = false; bbox_known = false; tables_dirty = true; have_last_ctx = false;
}

// ------------------------------------------------------------------------------------------------ boundaries
void World::set_boundary(uint32_t slot, uint64_t nn, const float* pos, const float* vel_h, uint32_t memberships,
                         uint32_t filter, bool wants_forces) {
    use_device();
    if (slot > bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range (slots are dense)");
    const bool is_new = slot == bounds.size();
    const uint64_t old_n = is_new ? 0 : bounds[slot].n;
    if ((uint64_t)nb - old_n + nn >= 0xfffffff0ull) throw HipError(SALVA_HIP_E_CAPACITY, "too many boundary particles");
    if (nn && !pos) throw HipError(SALVA_HIP_E_INVALID, "boundary positions are required");
    if (is_new) bounds.emplace_back();
    const uint64_t off = boundary_offset(slot);
    const uint64_t old_total = nb, new_total = old_total - old_n + nn;
    const bool resized = is_new || old_n != nn;
    if (resized) {
        std::vector<Piece> pieces = {{0, off, true}, {0, nn, false}, {off + old_n, old_total - off - old_n, true}};
        rebuild(bst_pos, pieces, new_total, stream);
        rebuild(bst_vel, pieces, new_total, stream);
        rebuild(bforce, pieces, new_total, stream);
        nb = (uint32_t)new_total;
        if (nn) {
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, bst_pos.p + off, make_float4(0, 0, 0, 0), 0);
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, bst_vel.p + off, make_float4(0, 0, 0, 0), 0);
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, bforce.p + off, make_float4(0, 0, 0, 0), 0);
        }
    }
    BoundarySlot& b = bounds[slot];
    b.n = nn; b.memberships = memberships; b.filter = filter; b.wants_forces = wants_forces;
    b.dyn_kind = 0;  // (re)uploading particles makes it a plain boundary; the sampling setters mark it again
    b.vel_zero = true;
    if (nn) {
        scratch_f.ensure(3 * nn, stream, false, 1.1f);
        SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, pos, 3 * nn * sizeof(float), hipMemcpyHostToDevice, stream));
        k_pack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f.p, bst_pos.p + off, 0, 0.0f);
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        b.vel_zero = true;
        if (vel_h)
            for (uint64_t k = 0; k < 3 * nn && b.vel_zero; ++k) b.vel_zero = vel_h[k] == 0.0f;
        if (vel_h) {
            SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, vel_h, 3 * nn * sizeof(float), hipMemcpyHostToDevice, stream));
            k_pack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f.p, bst_vel.p + off, 0, 0.0f);
        } else {
            k_fill_f4<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, bst_vel.p + off, make_float4(0, 0, 0, 0), 0);
        }
    }
    // boundary model ids ride in bst_vel.w
    uint64_t o = 0;
    for (uint32_t s = 0; s < bounds.size(); ++s) {
        if (bounds[s].n) k_set_bmodel<<<nblk(bounds[s].n), BLOCK, 0, stream>>>((uint32_t)bounds[s].n, bst_vel.p + o, s);
        o += bounds[s].n;
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    b_dirty = true; tables_dirty = true; have_last_ctx = false;
}

void World::remove_boundary(uint32_t slot) {
    use_device();
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    const uint32_t last = (uint32_t)bounds.size() - 1;
    const uint64_t off = boundary_offset(slot), len = bounds[slot].n;
    const uint64_t loff = boundary_offset(last), llen = bounds[last].n;
    std::vector<Piece> pieces;
    pieces.push_back({0, off, true});
    if (slot != last) {
        pieces.push_back({loff, llen, true});
        pieces.push_back({off + len, loff - off - len, true});
    }
    const uint64_t new_total = nb - len;
    rebuild(bst_pos, pieces, new_total, stream);
    rebuild(bst_vel, pieces, new_total, stream);
    rebuild(bforce, pieces, new_total, stream);
    if (slot != last) bounds[slot] = bounds[last];
    bounds.pop_back();
    nb = (uint32_t)new_total;
    uint64_t o = 0;
    for (uint32_t s = 0; s < bounds.size(); ++s) {
        if (bounds[s].n) k_set_bmodel<<<nblk(bounds[s].n), BLOCK, 0, stream>>>((uint32_t)bounds[s].n, bst_vel.p + o, s);
        o += bounds[s].n;
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    b_dirty = true; tables_dirty = true; have_last_ctx = false;
}

// ------------------------------------------------------------------------------------------------ tables
// InteractionGroups::test (interaction_groups.rs:64-69)
static inline bool groups_test(uint32_t m1, uint32_t f1, uint32_t m2, uint32_t f2) {
    return (m1 & f2) != 0 && (m2 & f1) != 0;
}

void World::upload_tables() {
    if (!tables_dirty) return;
    lists_checked = trust_cap0;  // the objects changed: the next step checks the list capacity before it solves (World::step)
    const uint32_t nm = (uint32_t)std::max<size_t>(fluids.size(), 1), nbm = (uint32_t)std::max<size_t>(bounds.size(), 1);
    std::vector<float> r0(nm, 1000.0f);
    std::vector<uint32_t> counts(nm, 0);
    std::vector<uint8_t> ff(nm * nm, 1), fb(nm * nbm, 1), bb(nbm * nbm, 1), bw(nbm, 0);
    any_wants_forces = false;
    for (uint32_t a = 0; a < fluids.size(); ++a) {
        r0[a] = fluids[a].density0;
        counts[a] = (uint32_t)fluids[a].n;
        for (uint32_t b = 0; b < fluids.size(); ++b)  // a fluid always interacts with itself (contacts.rs:350-358)
            ff[a * nm + b] = (a == b) || groups_test(fluids[a].memberships, fluids[a].filter, fluids[b].memberships, fluids[b].filter);
        for (uint32_t b = 0; b < bounds.size(); ++b)
            fb[a * nbm + b] = groups_test(fluids[a].memberships, fluids[a].filter, bounds[b].memberships, bounds[b].filter);
    }
    for (uint32_t a = 0; a < bounds.size(); ++a) {
        bw[a] = bounds[a].wants_forces ? 1 : 0;
        any_wants_forces |= bounds[a].wants_forces;
        for (uint32_t b = 0; b < bounds.size(); ++b)
            bb[a * nbm + b] = (a == b) || groups_test(bounds[a].memberships, bounds[a].filter, bounds[b].memberships, bounds[b].filter);
    }
    rho0_tab.ensure(nm); model_counts.ensure(nm); ff_ok.ensure(nm * nm); fb_ok.ensure(nm * nbm); bb_ok.ensure(nbm * nbm); bwants.ensure(nbm);
    SALVA_HIP_CHECK(hipMemcpyAsync(rho0_tab.p, r0.data(), nm * sizeof(float), hipMemcpyHostToDevice, stream));
    // (decomposed runs: the error averages divide by the GLOBAL particle count of each fluid, all-reduced once by
    // dist_prepare — a later table upload, e.g. after the caller re-uploaded boundaries for a re-cut slab, must not put the
    // local counts back: the ranks would then disagree on convergence and fall out of step)
    if (comm && dist_started && global_counts.size() == nm) counts = global_counts;
    SALVA_HIP_CHECK(hipMemcpyAsync(model_counts.p, counts.data(), nm * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(ff_ok.p, ff.data(), ff.size(), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(fb_ok.p, fb.data(), fb.size(), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(bb_ok.p, bb.data(), bb.size(), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(bwants.p, bw.data(), bw.size(), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));  // host vectors go out of scope
    tables_dirty = false;
    b_dirty = true;  // group changes alter the boundary-boundary sums
}

StepCtx World::make_ctx() {
    StepCtx c{};
    c.sc = sc;
    {   // groups of 64 slots (what one XCD's 32 CUs hold at a time), fewer for launches of less than 1024 tiles
        uint32_t lg = 1u;
        while (lg < 7u && (16u << lg) <= nlaunch) ++lg;
        c.xcd = lg;
    }
    c.n = n;
    c.posm = posm[cur].p; c.vel = vel[cur].p; c.dv = dv[cur].p; c.acc = acc.p; c.w = w.p; c.normal = normal.p;
    c.model = model[cur].p; c.perm = perm[cur].p; c.gtag = comm ? gtag[cur].p : nullptr;
    c.rho = rho.p; c.alpha = alpha.p; c.kappa = kappa.p; c.kappa2 = kappa2.p; c.rho_star = rho_star.p; c.aii = aii.p;
    c.dii = dii.p; c.dijpj = dijpj.p; c.iisph_q = iisph_q.p; c.iisph_pr = iisph_pr.p; c.posmr = posmr.p;
    c.nff = nff.p; c.nfb = nfb.p;
    c.nbr_ff = nbr_ff.p; c.nbr_fb = nbr_fb.p; c.cap_ff = cap_ff; c.cap_fb = cap_fb;
    c.slice_near = slice_near.p;
    c.tile_off = G().tile_off.p; c.halo_src = halo_src.p; c.bhalo_src = bhalo_src.p;
    c.halo_stride = halo_stride; c.bhalo_stride = bhalo_stride;
    c.ntiles = (uint32_t)gf.ntiles();
    c.split_s = split_s_cur;
    c.slot_order = (class_ntiny || class_nlight) ? slot_order.p : nullptr; c.slot_base = 0u; c.ntiny = class_ntiny; c.nlight = class_nlight;
    c.tile_ids = G().tile_ids.p; c.tile_rank = G().tile_rank.p; c.nlaunch = nlaunch; c.slot_desc = G().slot_desc.p; c.slot_info = slot_info.p;
    c.spec = spec_mode ? 1u : 0u; c.halo_cap = halo_cap; c.bhalo_cap = bhalo_cap; c.nslices_cap = nslices_cap;
    c.halo_len = halo_len; c.bhalo_len = bhalo_len;
    c.gf = gf.device(G().cell_start_f.p);
    c.stale_keys = has_dynamic_sampling() ? G().keys[1].p : nullptr;
    c.nb = nb;
    c.bposv = bposv.p; c.bvel = bvel.p; c.bperm = bperm.p;
    c.bforce = any_wants_forces ? bforce.p : nullptr;
    c.bwants = bwants.p;
    c.gb = gb.device(cell_start_b.p);
    c.nmodels = (uint32_t)std::max<size_t>(fluids.size(), 1);
    c.nbmodels = (uint32_t)std::max<size_t>(bounds.size(), 1);
    c.mass_uniform = mass_uniform;
    c.two_mass = two_mass ? 1u : 0u;
    c.nmass = two_mass ? nmass : 0u; c.cmask = mass_cmask;
    for (int k = 0; k < 4; ++k) c.class_mass[k] = mass_classes[k];
    c.tile_mass_bits = two_mass ? tile_mass_bits.p : nullptr;
    c.tile_massb_bits = two_mass ? tile_massb_bits.p : nullptr;
    c.tile_masscd_bits = (two_mass && nmass > 2u) ? tile_masscd_bits.p : nullptr;
    c.nffb = two_mass ? nffb.p : nullptr;
    c.nffc = (two_mass && nmass > 2u) ? nffc.p : nullptr;
    c.bvel_zero = 1u;  // no boundary particle moves: the passes that subtract a boundary velocity need not stage it
    for (const BoundarySlot& b : bounds) if (b.n && (!b.vel_zero || b.sampling || b.dyn_kind)) c.bvel_zero = 0u;
    c.rho0_tab = rho0_tab.p; c.rho0_single = fluids.empty() ? 1000.0f : fluids[0].density0; c.ff_ok = ff_ok.p; c.fb_ok = fb_ok.p; c.bb_ok = bb_ok.p;
    c.partials = partials.p;
    c.spec_k = -1; c.spec_ring = spec_ring.p; c.spec_pub = nullptr; c.model_counts = model_counts.p; c.w2 = w2.p;
    c.flags = d_flags.p;
    c.min_neighbors_for_divergence = 20;  // dfsph_solver.rs:62 (DIM == 3)
    c.phase = 0;
    c.ghost_lo_cx = (comm && comm->has_lo()) ? slab_lo - 1 : INT32_MIN;
    c.ghost_hi_cx = (comm && comm->has_hi()) ? slab_hi + 1 : INT32_MAX;
    return c;
}

// Tile grid over a cell bounding box.  The origin is the box's own minimum corner, so a block of fluid is cut into
// the same number of tiles wherever it sits (an absolute tile lattice would add a partially filled layer of tiles per
// axis whenever the block is not aligned with it).  Fluid and boundary grids have independent origins: a fluid
// tile looks boundary cells up by absolute cell coordinates.
// `fold` (fluid grid only): when the box holds more than fold->budget cells, fold its longest axes to power-of-two periods — never
// below fold->min_period[a] — until it does (device_types.h TileGrid: the table becomes a torus, the lists stay what they were).
struct FoldRule { double budget, target; uint32_t min_period[3]; bool axis[3]; };  // fold when the box exceeds `budget` cells, then down to `target`; axis[a]: may fold
static void dims_from_bbox(const int32_t* bb, GridDims& g, const FoldRule* fold = nullptr) {
    static const int T[3] = {TX, TY, TZ};
    int64_t cells[3];
    for (int a = 0; a < 3; ++a) {
        if (bb[a] > bb[3 + a]) throw HipError(SALVA_HIP_E_CAPACITY, "empty cell bounding box");
        const int64_t extent = (int64_t)bb[3 + a] - (int64_t)bb[a] + 1;
        if (extent > (1 << 30)) throw HipError(SALVA_HIP_E_CAPACITY, "cell bounding box too large");
        g.o[a] = bb[a];
        g.nt[a] = (int)((extent + T[a] - 1) / T[a]);
        g.mask[a] = 0xffffffffu;
        cells[a] = (int64_t)g.nt[a] * T[a];
    }
    if (fold && (double)cells[0] * (double)cells[1] * (double)cells[2] > fold->budget) {
        for (;;) {
            if ((double)cells[0] * (double)cells[1] * (double)cells[2] <= fold->target) break;
            // the longest axis that can still be folded: to the largest power of two below its present length
            int best = -1;
            int64_t best_p = 0;
            for (int a = 0; a < 3; ++a) {
                int64_t p = 1;
                while (p * 2 < cells[a]) p *= 2;
                if (!fold->axis[a] || p < (int64_t)fold->min_period[a] || p >= cells[a]) continue;
                if (best < 0 || cells[a] > cells[best]) { best = a; best_p = p; }
            }
            if (best < 0) break;  // (nothing left to fold: the budget check below decides)
            cells[best] = best_p;
            g.nt[best] = (int)(best_p / T[best]);
            g.mask[best] = (uint32_t)best_p - 1u;
        }
    }
    const double nc = (double)cells[0] * (double)cells[1] * (double)cells[2];
    if (nc >= 4.0e9) throw HipError(SALVA_HIP_E_CAPACITY, "dense cell table would exceed 2^32 cells; particles are too spread out");
    // The reference's hash grid costs memory per OCCUPIED cell; this table costs 4 bytes per cell of the (folded) bounding box, plus
    // 8 bytes per tile of 64 cells.  Where folding is not possible (decomposed runs, dynamic contact sampling, a boundary set as wide
    // as the box) a few stray particles far from the rest cost memory here that they do not cost there: refuse beyond a budget, with
    // a message that says what to do, rather than exhaust HBM.
    static const double budget_gib = [] {
        const char* e = getenv("SALVA_HIP_CELL_TABLE_GIB");
        const double v = e ? atof(e) : 8.0;
        return v > 0.0 ? v : 8.0;
    }();
    const double gib = (nc * 4.0 + nc / TCELLS * 8.0) / (1024.0 * 1024.0 * 1024.0);
    if (gib > budget_gib) {
        char msg[512];
        snprintf(msg, sizeof msg,
                 "the cell table over the particles' bounding box (%d x %d x %d tiles of %dx%dx%d cells) would take %.1f GiB, over the "
                 "%.1f GiB budget: particles are too spread out for the dense grid — delete strays (salva_hip_particles_intersecting_aabb "
                 "+ salva_hip_delete_particles) or raise SALVA_HIP_CELL_TABLE_GIB",
                 g.nt[0], g.nt[1], g.nt[2], TX, TY, TZ, gib, budget_gib);
        throw HipError(SALVA_HIP_E_CAPACITY, msg);
    }
}

// Sort the boundary particles by cell, build their cell table and volumes.  Runs when the boundary set changed
// (the reference redoes this every substep, contacts.rs:142-151 + dfsph_solver.rs:72-96; for unchanged boundary
// positions the result is identical).
void World::build_boundary_grid() {
    if (!b_dirty) return;
    ncontacts_bb = 0;
    if (nb == 0) { b_dirty = false; return; }
    bbox_partials.ensure(6 * std::max<size_t>(bbox_blocks(nb), 1024));
    launch_bbox(bst_pos.p, nb, sc.h, bbox_partials.p, d_rb.p->bbbox, d_flags.p, stream);
    SALVA_HIP_CHECK(hipMemcpyAsync(h_rb->bbbox, d_rb.p->bbbox, sizeof(int32_t) * 6, hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    dims_from_bbox(h_rb->bbbox, gb);
    const size_t nc = gb.ncells();
    bkeys[0].ensure(nb); bkeys[1].ensure(nb); bidx[0].ensure(nb); bidx[1].ensure(nb);
    bposv.ensure(nb); bvel.ensure(nb); bperm.ensure(nb); cell_start_b.ensure(nc + 1);
    TileGrid gv = gb.device(nullptr);
    launch_cell_keys(bst_pos.p, nb, sc.h, gv, bkeys[0].p, bidx[0].p, d_flags.p, nullptr, nullptr, nullptr, stream);
    const int end_bit = bits_for(nc);
    const size_t tb = sort_pairs_temp_bytes(nb, end_bit);
    ensure_cub_temp(tb);
    sort_pairs(cub_temp.p, tb, bkeys[0].p, bkeys[1].p, bidx[0].p, bidx[1].p, nb, end_bit, stream);
    launch_reorder_boundary(nb, bidx[1].p, bst_pos.p, bst_vel.p, bposv.p, bvel.p, bperm.p, stream);
    launch_cell_start(bkeys[1].p, nb, (uint32_t)nc, cell_start_b.p, stream);
    StepCtx c = make_ctx();
    SALVA_HIP_CHECK(hipMemsetAsync(d_counters.p + 2, 0, sizeof(unsigned long long), stream));
    launch_boundary_volumes(c, d_counters.p + 2, stream);
    SALVA_HIP_CHECK(hipMemcpyAsync(&h_rb->ncontacts_bb, d_counters.p + 2, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    ncontacts_bb = h_rb->ncontacts_bb;
    b_dirty = false;
}

// The convergence loops read one float back per iteration; an interrupt-driven hipStreamSynchronize costs tens of
// microseconds per wake-up, polling an event a few.
// The per-step read-backs without a copy engine: one wave copies the few words the host is waiting for into host-mapped memory,
// fences to system scope, then bumps the sequence word the host polls.  (A hipMemcpyAsync + event costs ~20 us of idle GPU each
// time the host has to wait for it: tools/gap_tsv_report.py.)
// What an end-of-step publication folds before it publishes (round 6: one launch where k_list_stats, k_bbox_final and the publication
// were three, each a few microseconds long with a launch gap on either side).
struct Epilogue {
    const TileListStats* ts; uint32_t nts; int own;   // k_nbr_tile's per-tile list statistics -> ncontacts_*, max_cnt_* (nullptr: k_list_stats ran)
    const int32_t* bbox_partials; uint32_t nbb;        // the position update's per-block cell boxes -> bbox (nullptr: nothing to fold)
    const uint32_t* chain_gate;                        // ... which was gated by this word (a chained step; nullptr: it ran)
};
__global__ __launch_bounds__(BLOCK) void k_publish_readback(Readback* __restrict__ src, const TileAcc* __restrict__ totals, int lists, int end_of_step,
                                                            uint32_t* mass_slots, Readback* pub_rb, volatile uint32_t* pub_seq, uint32_t seq,
                                                            const SolveCtl* __restrict__ ctl, PrePub pre, const uint32_t* gate, Epilogue ep) {
    if (gate && *gate == 0u) return;  // (the totals of a pre-enqueued grid that did not come true: nobody waits for them)
    __shared__ unsigned long long sred[4][BLOCK / WAVE];
    __shared__ int ired[6 * (BLOCK / WAVE)];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    if (ep.ts) {  // (block-uniform) the fold of k_list_stats, grid.hip
        unsigned long long a = 0, b = 0, oa = 0, ob = 0;
        uint32_t ma = 0, mb = 0;
        for (uint32_t k = threadIdx.x; k < ep.nts; k += blockDim.x) {
            const TileListStats t = ep.ts[k];
            a += t.sum_ff; b += t.sum_fb; oa += t.own_ff; ob += t.own_fb; ma = max(ma, t.max_ff); mb = max(mb, t.max_fb);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, WAVE); b += __shfl_xor(b, o, WAVE);
            oa += __shfl_xor(oa, o, WAVE); ob += __shfl_xor(ob, o, WAVE);
        }
        ma = wave_max_u32(ma); mb = wave_max_u32(mb);
        if (lane == 0) { sred[0][wv] = a; sred[1][wv] = b; sred[2][wv] = oa; sred[3][wv] = ob; ired[wv] = (int)ma; ired[nw + wv] = (int)mb; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long ta = 0, tb = 0, toa = 0, tob = 0; uint32_t xa = 0, xb = 0;
            for (int k = 0; k < nw; ++k) {
                ta += sred[0][k]; tb += sred[1][k]; toa += sred[2][k]; tob += sred[3][k];
                xa = max(xa, (uint32_t)ired[k]); xb = max(xb, (uint32_t)ired[nw + k]);
            }
            src->ncontacts_ff = ta; src->ncontacts_fb = tb; src->max_cnt_ff = xa; src->max_cnt_fb = xb;
            if (ep.own) { src->ncontacts_own_ff = toa; src->ncontacts_own_fb = tob; }
        }
        __syncthreads();
    }
    if (ep.bbox_partials && !(ep.chain_gate && gate_words_closed(ep.chain_gate[0], ep.chain_gate[1], 0u))) {  // the fold of k_bbox_final
        int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
        for (uint32_t k = threadIdx.x; k < ep.nbb; k += blockDim.x) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], ep.bbox_partials[6 * k + a]); mx[a] = max(mx[a], ep.bbox_partials[6 * k + 3 + a]); }
        }
        block_bbox_store(mn, mx, ired, src->bbox);  // (ends with a barrier: thread 0 below reads what threads 0..5 stored)
        __threadfence_block();
    }
    uint32_t mlo = 0u, mhi = 0u;
    if (totals && wv == 0) {  // did k_cell_keys see a mass other than particle 0's since the last such publication?  start the next one
        static_assert(MASS_SLOTS == WAVE, "one flag per lane of the publishing wave");
        const uint32_t differs = wave_max_u32(mass_slots[lane]);
        mass_slots[lane] = 0u;
        mlo = mass_slots[MASS_SLOTS];
        mhi = differs ? ~mlo : mlo;
    }
    if (threadIdx.x == 0) {
        if (totals) {
            pub_rb->tile_total = *totals;
            pub_rb->mass_mm[0] = mlo; pub_rb->mass_mm[1] = mhi;
        }
        if (lists) {
            pub_rb->ncontacts_ff = src->ncontacts_ff; pub_rb->ncontacts_fb = src->ncontacts_fb;
            pub_rb->max_cnt_ff = src->max_cnt_ff; pub_rb->max_cnt_fb = src->max_cnt_fb;
            pub_rb->ncontacts_own_ff = src->ncontacts_own_ff; pub_rb->ncontacts_own_fb = src->ncontacts_own_fb;
        }
        if (end_of_step) {
            pub_rb->flags = src->flags;
            src->flags = 0u;  // (the next step starts from clear flags without a memset of its own: World::flags_clean)
            const volatile int32_t* bb = src->bbox;  // (stored by threads 0..5 of this block a moment ago: not through a stale register)
            int32_t box[6];
            for (int a = 0; a < 6; ++a) { box[a] = bb[a]; pub_rb->bbox[a] = box[a]; }
            // chained steps (device_types.h StepCtx::gate): did every solve converge within its batch, and what they found
            pub_rb->chain_ok = src->chain_ok; pub_rb->chain_stage = src->chain_stage;
            uint32_t pok = (pre.on && (!pre.chained || src->chain_ok)) ? 1u : 0u;
            for (int a = 0; a < 6; ++a) pok &= (box[a] == pre.bbox[a]) ? 1u : 0u;
            src->pre_ok = pok; pub_rb->pre_ok = pok;
            for (int k = 0; k < 2; ++k) {
                pub_rb->solve[k][0] = ctl[k].done; pub_rb->solve[k][1] = ctl[k].iters;
                pub_rb->solve[k][2] = __float_as_uint(ctl[k].err); pub_rb->solve[k][3] = ctl[k].seq;
            }
        }
        __threadfence_system();
        *pub_seq = seq;
    }
}
// enqueue the publication on the world's stream ...
uint32_t World::publish_enqueue(const TileAcc* totals, bool lists, bool end_of_step, const PrePub* pre, const uint32_t* gate) {
    const uint32_t seq = ++hostpub_seq;
    Epilogue ep{nullptr, 0u, 0, nullptr, 0u, nullptr};
    if (end_of_step) {  // what this step left for the publication to fold (World::substep sets them, one publication consumes them)
        if (fold_stats) { ep.ts = reinterpret_cast<const TileListStats*>(tile_list_stats.p); ep.nts = nlaunch; ep.own = comm ? 1 : 0; }
        if (fold_bbox_blocks) { ep.bbox_partials = bbox_partials.p; ep.nbb = fold_bbox_blocks; ep.chain_gate = fold_bbox_gate; }
        fold_stats = false; fold_bbox_blocks = 0u; fold_bbox_gate = nullptr;
    }
    const unsigned threads = (ep.ts || ep.bbox_partials) ? BLOCK : WAVE;
    k_publish_readback<<<1, threads, 0, stream>>>(d_rb.p, totals, lists ? 1 : 0, end_of_step ? 1 : 0, mass_slots.p, &h_hostpub->rb, &h_hostpub->seq, seq,
                                                  d_ctl.p, pre ? *pre : PrePub{0, 0, {0, 0, 0, 0, 0, 0}}, gate, ep);
    SALVA_HIP_CHECK(hipGetLastError());
    return seq;
}
// ... and wait for it (kernels enqueued in between keep the GPU busy meanwhile); the published words are folded into h_rb
void World::publish_wait(uint32_t seq, bool totals, bool lists, bool end_of_step) {
    auto last_query = std::chrono::steady_clock::now();
    // (sequence numbers only grow; the totals of a pre-enqueued grid may follow the end-of-step publication before the host has
    // looked — the two write different fields)
    auto arrived = [&] { return (int32_t)(__atomic_load_n(&h_hostpub->seq, __ATOMIC_ACQUIRE) - seq) >= 0; };
    for (uint32_t spins = 0; !arrived(); ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xffu) != 0xffu) continue;
        const auto now = std::chrono::steady_clock::now();
        if (now - last_query < std::chrono::microseconds(100)) continue;
        last_query = now;
        const hipError_t e = hipStreamQuery(stream);  // a fault on the stream would otherwise spin forever
        if (e != hipSuccess && e != hipErrorNotReady) SALVA_HIP_CHECK(e);
        if (e == hipSuccess && !arrived())
            throw HipError(SALVA_HIP_E_HIP, "internal error: the stream drained without publishing its read-back");
    }
    const Readback& p = h_hostpub->rb;
    if (totals) { h_rb->tile_total = p.tile_total; h_rb->mass_mm[0] = p.mass_mm[0]; h_rb->mass_mm[1] = p.mass_mm[1]; }
    if (lists) {
        h_rb->ncontacts_ff = p.ncontacts_ff; h_rb->ncontacts_fb = p.ncontacts_fb; h_rb->max_cnt_ff = p.max_cnt_ff; h_rb->max_cnt_fb = p.max_cnt_fb;
        h_rb->ncontacts_own_ff = p.ncontacts_own_ff; h_rb->ncontacts_own_fb = p.ncontacts_own_fb;
    }
    if (end_of_step) {
        h_rb->flags = p.flags; memcpy(h_rb->bbox, p.bbox, sizeof(p.bbox));
        h_rb->chain_ok = p.chain_ok; h_rb->chain_stage = p.chain_stage; memcpy(h_rb->solve, p.solve, sizeof(p.solve));
        h_rb->pre_ok = p.pre_ok;
    }
}
void World::publish_and_wait(const TileAcc* totals, bool lists, bool end_of_step) {
    publish_wait(publish_enqueue(totals, lists, end_of_step), totals != nullptr, lists, end_of_step);
}

void World::wait_stream() {
    SALVA_HIP_CHECK(hipEventRecord(ev_sync, stream));
    for (;;) {
        const hipError_t e = hipEventQuery(ev_sync);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) SALVA_HIP_CHECK(e);
    }
}

// One iterative solve with the reference's protocol (dfsph_solver.rs:439-463 / :474-502 / iisph_solver.rs:422-456):
//   for i in 0..max { err = evaluate(); if err <= tol && i >= min { break }; apply(); }
// The break decision is taken on the device (k_finalize_error -> SolveCtl); iterations are enqueued in growing batches
// and the control block is read back once per batch.  Kernels enqueued after convergence return immediately.
__global__ void k_ghost_posmr(uint32_t n, const uint32_t* __restrict__ gtag, const float4* __restrict__ posm, const float* __restrict__ rho,
                              float4* __restrict__ posmr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(gtag[i] & 0x80000000u)) return;
    reinterpret_cast<float*>(&posmr[i])[3] = posm[i].w / rho[i];
}
__global__ void k_init_ctl(SolveCtl* ctl, SolveCtl* ring, SolveCtl init, uint32_t* chain_open, SolveCtl* ctl_b, SolveCtl init_b) {
    *ctl = init;
    if (ctl_b) *ctl_b = init_b;  // (the step's second solve, whose parameters are known as well: one launch for both)
    if (chain_open) { chain_open[0] = 1u; chain_open[1] = 0u; }  // (Readback::chain_ok / chain_stage: the first solve of a chained step)
    if (ring) { ring[0] = init; ring[1] = init; }  // (dfsph.hip spec_decide: the test rides in the apply pass; iteration k reads spec_ring[k & 1])
}
template <typename Eval, typename Apply>
World::SolveResult World::run_solve(StepCtx c, int which, float tol, int min_iter, int max_iter, uint32_t mode, Eval&& eval,
                                    Apply&& apply, bool spec_apply, int chain_stage, int from, bool chain_open) {
    // Every convergence test publishes its outcome to host-mapped memory (k_finalize_error; in a decomposed run k_decide,
    // behind the all-reduce), and the host waits for the test count it enqueued — it then decides (and enqueues what
    // follows) while the batch's last apply pass is still running.
    static const bool no_publish = getenv("SALVA_HIP_NO_PUBLISH") != nullptr;  // (diagnostics: A/B against the copy + wait)
    SolveCtl* const pub = no_publish ? nullptr : h_pub + which;
    c.ctl = d_ctl.p + which;
    if (from == 0) {
        SolveCtl& init = h_ctl[NUM_SOLVES + which];
        init = SolveCtl{0u, 0u, 0.0f, 0u, tol, (uint32_t)std::max(min_iter, 0), mode, 0u};
        h_ctl[which] = init;
        // (one tiny kernel with the record as its argument instead of up to three copies from pageable host memory, each of which
        // stalls the host until its staging copy is done; the first solve of a chained step also opens the chain)
        const bool have = which == 1 && pre_init1_valid && memcmp(&pre_init1, &init, sizeof(SolveCtl)) == 0 && !chain_open;
        if (!have) {
            const bool both = which == 0 && pre_init1_valid;  // (World::dfsph_solve has said what the pressure solve will start from)
            k_init_ctl<<<1, 1, 0, stream>>>(d_ctl.p + which, spec_apply ? spec_ring.p : nullptr, init, chain_open ? &d_rb.p->chain_ok : nullptr,
                                            both ? d_ctl.p + 1 : nullptr, both ? pre_init1 : init);
            SALVA_HIP_CHECK(hipGetLastError());
        }
        if (which == 1) pre_init1_valid = false;
        if (pub) { pub->done = 0u; pub->iters = 0u; pub->err = 0.0f; __atomic_store_n(&pub->seq, 0u, __ATOMIC_RELEASE); }
    }
    // First batch: what the previous step's solve needed (iters applies + the converged evaluate) — consecutive steps
    // need about the same, so the usual cost is one read-back per solve; a batch that overshoots only enqueues kernels
    // that return at once, one that falls short continues in doubling batches.
    int i = from, batch = std::max(2, std::min<int>((int)last_iters[which] + 1, max_iter));
    const bool multi = comm && comm->size() > 1;
    // Two launches a solve does not need (round 6; a kernel that returns at once still costs a launch of 2 200 workgroups, ~5 us, and
    // a one-block test ~4.6 us plus its gaps — 23 us of a 0.61 ms free-fall step):
    //  * the tests of iterations i < min_iter cannot end the solve (`err <= tol && i >= min`): they are not launched — the next test
    //    that is counts them (`skipped`: iters and seq advance as if they had failed).  In a decomposed run that is an all-reduce less.
    //  * the apply behind the LAST test of a batch runs only if that test fails, which the batch is sized not to expect: it is
    //    enqueued by whoever continues the solve (`owed`), and never when the solve has converged.  Not in decomposed runs (a ghost
    //    refresh rides behind every apply) and not where the test rides in the apply pass itself (spec_apply).
    const bool lazy_apply = !multi && !spec_apply;
    uint32_t skipped = 0u;
    auto test = [&](int it, bool last_of_batch, const uint32_t* gate, uint32_t* close, uint32_t stage) {
        if (it < min_iter && it + 1 < max_iter && !last_of_batch) { ++skipped; return; }
        if (!multi) launch_finalize_error(partials.p, nlaunch, (uint32_t)std::max<size_t>(fluids.size(), 1), model_counts.p, d_ctl.p + which, pub, stream, gate, close, stage, skipped);
        else finalize_solve(d_ctl.p + which, pub, skipped);
        skipped = 0u;
    };
    if (from > 0 && solve_owes_apply[which]) apply(c, from - 1);  // (the apply the batch before left to its successor)
    solve_owes_apply[which] = false;
    if (chain_stage) {
        // Chained (device_types.h StepCtx::gate): ONE batch and no wait.  The batch is what the previous step needed plus, while the
        // count is rising (the impact: 2, 4, 14, 18, 30 ... iterations in consecutive steps), what it rose by last time — a surplus
        // iteration costs three kernels that return at once, a batch that falls short costs the chain (every kernel behind this
        // solve returns at once and the host comes back here with `from` = this batch).  Its LAST test closes the chain when it
        // fails, unless the batch runs to max_iter: then the solve is over whatever that test says.
        const int rise = (int)last_iters[which] > (int)prev_iters[which] ? std::min((int)last_iters[which] - (int)prev_iters[which], 8) : 0;
        const int nbatch = std::min(batch + rise, max_iter);
        c.gate_stage = (uint32_t)chain_stage;  // (its own last test may shut the gate: the apply behind that test still runs)
        for (int k = 0; k < nbatch; ++k) {
            eval(c, k);
            const bool closes = k == nbatch - 1 && nbatch < max_iter;
            test(k, k == nbatch - 1, c.gate, closes ? &d_rb.p->chain_ok : nullptr, (uint32_t)chain_stage);
            if (closes && lazy_apply) solve_owes_apply[which] = true;  // (only a continuation needs it)
            else apply(c, k);
        }
        chain_batch[which] = nbatch;
        return SolveResult{0u, 0.0f};  // (pending: World::substep reads the outcome from the end-of-step publication)
    }
    if (from > 0) batch = (from <= 2) ? 4 : 8;
    while (i < max_iter) {
        const int nbatch = std::min(batch, max_iter - i);
        bool owes = false;
        for (int k = 0; k < nbatch; ++k) {
            if (spec_apply && multi) {
                // A decomposed solve with speculative applies (round 6): the all-reduced test and the apply pass do not wait for
                // each other.  Main stream: error sums -> all-reduce -> k_decide_ring (the record of iteration it + 1).  Second
                // stream: the apply pass, into the other w buffer, reading the record of iteration `it` only.  Then the ghost
                // refresh of what the apply wrote, behind both — one communicator, one operation at a time.  A converged test leaves
                // the apply and its refresh unused (w stays where it was), exactly as in the single domain (dfsph.hip spec_decide).
                const int it = i + k;
                c.spec_k = it; c.spec_external = 1u; c.spec_pub = nullptr;
                eval(c, it);
                SALVA_HIP_CHECK(hipEventRecord(ev_spec_eval, stream));
                SALVA_HIP_CHECK(hipStreamWaitEvent(stream2, ev_spec_eval, 0));
                spec_dist_stream = stream2;
                apply(c, it);  // (the kernel only: World::dfsph_solve's lambda leaves the refresh to the lines below)
                spec_dist_stream = nullptr;
                SALVA_HIP_CHECK(hipEventRecord(ev_spec_apply, stream2));
                const uint32_t nm = (uint32_t)std::max<size_t>(fluids.size(), 1);
                const size_t tm = dist_time_begin(1);
                launch_sum_partials(partials.p, nlaunch, nm, spec_ring.p + (it & 1), d_sums.p, stream);
                comm->allreduce_sum_f32(d_sums.p, (int)nm, stream);
                launch_decide_ring(d_sums.p, nm, model_counts.p, spec_ring.p, it, pub, stream);
                dist_time_end(tm);
                SALVA_HIP_CHECK(hipStreamWaitEvent(stream, ev_spec_apply, 0));
                refresh_f4((it & 1) ? w.p : w2.p);  // what the apply of iteration `it` wrote: the buffer of parity it + 1
                continue;
            }
            if (spec_apply) { c.spec_k = i + k; c.spec_pub = pub; }
            eval(c, i + k);
            if (!spec_apply) test(i + k, k == nbatch - 1, nullptr, nullptr, 0u);
            if (lazy_apply && k == nbatch - 1 && i + nbatch < max_iter) owes = true;
            else apply(c, i + k);
        }
        if (pub) {
            const uint32_t expect = (uint32_t)(i + nbatch);
            // spin on the host-mapped block; `pause` keeps the sibling hyper-thread usable (loopback tests run one host thread
            // per rank), and every ~100 us the stream is queried so that a fault on it surfaces instead of spinning forever
            auto last_query = std::chrono::steady_clock::now();
            for (uint32_t spins = 0; __atomic_load_n(&pub->seq, __ATOMIC_ACQUIRE) < expect; ++spins) {
                __builtin_ia32_pause();
                if ((spins & 0xffu) != 0xffu) continue;
                const auto now = std::chrono::steady_clock::now();
                if (now - last_query < std::chrono::microseconds(100)) continue;
                last_query = now;
                const hipError_t e = hipStreamQuery(stream);
                if (e != hipSuccess && e != hipErrorNotReady) SALVA_HIP_CHECK(e);
                if (e == hipSuccess && __atomic_load_n(&pub->seq, __ATOMIC_ACQUIRE) < expect)
                    throw HipError(SALVA_HIP_E_HIP, "internal error: a solver batch finished without publishing its control block");
            }
            h_ctl[which].done = pub->done; h_ctl[which].iters = pub->iters; h_ctl[which].err = pub->err;
        } else {
            // (no publication: read the newest record back — after i + nbatch iterations that is spec_ring[(i + nbatch) & 1])
            const SolveCtl* src = spec_apply ? spec_ring.p + ((i + nbatch) & 1) : d_ctl.p + which;
            SALVA_HIP_CHECK(hipMemcpyAsync(&h_ctl[which], src, sizeof(SolveCtl), hipMemcpyDeviceToHost, stream));
            wait_stream();
        }
        i += nbatch;
        if (h_ctl[which].done) break;
        if (owes) apply(c, i - 1);  // the batch's last test failed: the apply it counted
        batch = (i <= 2) ? 4 : 8;
    }
    prev_iters[which] = last_iters[which];
    last_iters[which] = h_ctl[which].iters;
    return SolveResult{h_ctl[which].iters, h_ctl[which].err};
}

// predict_advection's loop over `fluid.nonpressure_forces` (dfsph_solver.rs:580-603): fluids in slot order, forces in list order.
void World::run_forces(const StepCtx& c) {
    for (uint32_t f = 0; f < fluids.size(); ++f) {
        if (fluids[f].n == 0) continue;
        for (const SalvaHipForceDesc& d : fluids[f].forces) {
            switch (d.kind) {
                case SALVA_HIP_FORCE_XSPH: launch_xsph(c, lds, f, d.p[0], d.p[1], inv_dt_prev, stream); break;
                case SALVA_HIP_FORCE_ARTIFICIAL: launch_artificial_viscosity(c, lds, f, d.p[0], d.p[1], d.p[2], d.p[3], d.p[4], stream); break;
                case SALVA_HIP_FORCE_DFSPH_VISCOSITY: {
                    // DFSPHViscosity::solve (dfsph_viscosity.rs:290-327); timestep.dt() / inv_dt() are the previous step's here
                    const float coef = d.p[0], max_err = d.p[3];
                    const int min_it = (int)d.p[1], max_it = (int)d.p[2];
                    visc_beta.ensure((size_t)36 * n, stream, false, 1.1f); visc_target.ensure((size_t)6 * n, stream, false, 1.1f);
                    visc_u0.ensure(n, stream, false, 1.1f); visc_u1.ensure(n, stream, false, 1.1f); visc_va.ensure(n, stream, false, 1.1f);
                    launch_visc_betas(c, lds, f, visc_beta.p, stream);
                    launch_visc_va(c, dt_prev, visc_va.p, stream);
                    if (comm) refresh_f4(visc_va.p);
                    launch_visc_strain(c, lds, f, 0, coef, visc_va.p, visc_beta.p, visc_target.p, visc_u0.p, visc_u1.p, stream);
                    const SolveResult rv = run_solve(
                        c, 2, max_err, min_it, max_it, 0u,
                        [&](const StepCtx& cc, int) {
                            launch_visc_strain(cc, lds, f, 1, coef, visc_va.p, visc_beta.p, visc_target.p, visc_u0.p, visc_u1.p, stream);
                        },
                        [&](const StepCtx& cc, int) {
                            // (u of the inner ghost plane was computed here from refreshed v + a dt: no exchange needed)
                            launch_visc_accel(cc, lds, f, inv_dt_prev, dt_prev, visc_u0.p, visc_u1.p, visc_va.p, stream);
                            if (comm) refresh_f4(visc_va.p);
                        });
                    fluids[f].force_iters.resize(fluids[f].forces.size(), 0u);
                    fluids[f].force_errs.resize(fluids[f].forces.size(), 0.0f);
                    fluids[f].force_iters[&d - fluids[f].forces.data()] = rv.iters;
                    fluids[f].force_errs[&d - fluids[f].forces.data()] = rv.err;
                    break;
                }
                case SALVA_HIP_FORCE_HE2014:
                    // He2014SurfaceTension::solve (he2014_surface_tension.rs:109-181): colors -> gradcs -> forces
                    he_colors.ensure(n, stream, false, 1.1f); he_gradcs.ensure(n, stream, false, 1.1f);
                    launch_he2014_colors(c, lds, f, he_colors.p, stream);
                    // the outer ghost plane's colors are sums over an incomplete neighbourhood, and the inner plane's gradcs read them
                    if (comm) refresh_f32(he_colors.p);
                    launch_he2014_gradc(c, lds, f, he_colors.p, he_gradcs.p, stream);
                    launch_he2014_forces(c, lds, f, d.p[0], d.p[1], he_gradcs.p, stream);
                    break;
                case SALVA_HIP_FORCE_WCSPH_TENSION: launch_wcsph_tension(c, lds, f, d.p[0], stream); break;
                case SALVA_HIP_FORCE_CUSTOM: {
                    // a host `NonPressureForce::solve` at its place in the list (nonpressure_force.rs:10-30)
                    if (!force_cb) throw HipError(SALVA_HIP_E_INVALID, "a SALVA_HIP_FORCE_CUSTOM entry needs salva_hip_set_force_callback");
                    // (in a decomposed run the callback runs on every rank and works on the rank's local view:
                    // salva_hip_get_local / _get_local_contacts / _force_add_local_accelerations)
                    last_ctx = c; last_ctx.ctl = nullptr; have_last_ctx = true;  // what the contact export reads
                    wait_stream();
                    in_force_cb = true;
                    int rc = 0;
                    try {
                        rc = force_cb(force_user, force_owner, f, (uint32_t)(&d - fluids[f].forces.data()), dt_prev, inv_dt_prev);
                    } catch (...) {
                        in_force_cb = false; have_last_ctx = false;
                        throw;
                    }
                    in_force_cb = false; have_last_ctx = false;
                    if (rc != 0) throw HipError(SALVA_HIP_E_INVALID, "the force callback reported an error");
                    break;
                }
                case SALVA_HIP_FORCE_AKINCI2013:
                    launch_akinci_normals(c, lds, f, stream);
                    // (normals of the inner ghost plane are complete: rho was refreshed on both planes)
                    launch_akinci_forces(c, lds, f, d.p[0], d.p[1], stream);
                    break;
                default: break;
            }
        }
    }
}

// An evaluate pass of a decomposed run, overlapped with the ghost exchange that precedes it: the tiles whose halo box
// touches no ghost plane (all but the two tile layers at the faces) start on the second stream as soon as everything before
// the exchange is done; the exchange itself (gather, grouped send / recv with the neighbours, scatter: latency-bound, tens
// of microseconds) proceeds on the main stream, followed by the border tiles; the main stream then waits for the interior.
// (Not for the first evaluate of a solve: its control block is initialised on the main stream after the exchange was
// enqueued, and the interior launch would read the previous solve's `done`.)
template <typename Launch>
void World::evaluate_split(const StepCtx& c, int iteration, Launch&& launch) {
    // (nothing to overlap with when no ghost is refreshed: a one-rank communicator, or a rank whose faces hold no particle)
    if (!comm || !overlap_exchange || iteration == 0 || nghost_lo + nghost_hi + nborder_lo + nborder_hi == 0) { launch(c, stream); return; }
    StepCtx ci = c, cb = c;
    ci.phase = 1; cb.phase = 2;
    SALVA_HIP_CHECK(hipStreamWaitEvent(stream2, ev_pre_refresh, 0));
    launch(ci, stream2);
    SALVA_HIP_CHECK(hipEventRecord(ev_interior, stream2));
    launch(cb, stream);
    SALVA_HIP_CHECK(hipStreamWaitEvent(stream, ev_interior, 0));
}

// Worlds with a few masses (device_types.h StepCtx::two_mass): a single-domain DFSPH world with the default kernels in which every
// non-empty fluid has one particle mass (FluidSlot::vol_uniform x density0, the product launch_stage_to_sorted forms) and two, three
// or four different masses occur — BASELINE config 4 has two.  Sets mass_classes / mass_cmask / nmass.
bool World::decide_two_mass() {
    mass_cmask = 0ull; nmass = 0u;
    for (float& m : mass_classes) m = 0.0f;
    if (two_mass_off || no_planes || comm || prm.solver != SALVA_HIP_SOLVER_DFSPH || prm.kernel_density != 0 || prm.kernel_gradient != 0) return false;
    if (fluids.size() < 2 || fluids.size() > 32 || bounds.size() > 32) return false;
    float ms[4];
    uint32_t distinct = 0;
    for (const FluidSlot& f : fluids) {
        if (f.n == 0) continue;
        const float m = f.vol_uniform * f.density0;
        if (!(m > 0.0f) || !std::isfinite(m)) return false;  // (NaN: the fluid's volumes differ)
        bool seen = false;
        for (uint32_t k = 0; k < distinct; ++k) seen |= ms[k] == m;
        if (seen) continue;
        if (distinct == max_masses) return false;  // one mass too many: the general kernels
        ms[distinct++] = m;
    }
    if (distinct < 2) return false;
    std::sort(ms, ms + distinct);
    for (uint32_t k = 0; k < distinct; ++k) mass_classes[k] = ms[k];
    for (uint32_t f = 0; f < fluids.size(); ++f) {
        if (!fluids[f].n) continue;  // (an empty fluid has no particle whose class could be asked for)
        const float m = fluids[f].vol_uniform * fluids[f].density0;
        for (uint32_t k = 0; k < distinct; ++k)
            if (ms[k] == m) mass_cmask |= (uint64_t)k << (2u * f);
    }
    nmass = distinct;
    return true;
}

void World::set_cfl(int mode, float coeff, int min_sub, int max_sub) {
    if (mode < 0 || mode > 2) throw HipError(SALVA_HIP_E_INVALID, "cfl mode must be 0 (off), 1 (the reference's commented clamp) or 2 (the same, cut at the end of the step)");
    if (mode && (!(coeff > 0.0f) || !std::isfinite(coeff))) throw HipError(SALVA_HIP_E_INVALID, "cfl_coeff must be positive");
    if (mode && (min_sub < 1 || max_sub < min_sub)) throw HipError(SALVA_HIP_E_INVALID, "need 1 <= min_num_substeps <= max_num_substeps");
    cfl_mode = mode; cfl_coeff = coeff; cfl_min_sub = min_sub; cfl_max_sub = max_sub;
}
// TimestepManager::max_substep (timestep_manager.rs:36-46) + the body of compute_substep the reference left commented out (:90-93):
//   max_sq_vel = max over all fluid particles of |v + a * remaining_time|^2        (f32::max ignores a NaN operand)
//   computed   = particle_radius * 2 / sqrt(max_sq_vel) * cfl_coeff
//   substep    = clamp(computed, total / max_num_substeps, total / min_num_substeps)
// The maximum is order-independent and |.|^2 is evaluated as the reference does ((x x + y y) + z z, no contraction: the library is
// compiled -ffp-contract=off), so the substep is bit for bit the CPU's for the same v and a.  Non-negative floats order like
// their bit patterns: one atomicMax per wave.
__global__ __launch_bounds__(BLOCK) void k_cfl_max(uint32_t n, const float4* __restrict__ vel, const float4* __restrict__ acc,
                                                   const uint32_t* __restrict__ gtag, float remaining, uint32_t* __restrict__ out_bits) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t bits = 0u;
    if (i < n && !(gtag && (gtag[i] & 0x80000000u))) {  // (a ghost is its owner's to report)
        const float4 v = vel[i], a = acc[i];
        const float ux = v.x + a.x * remaining, uy = v.y + a.y * remaining, uz = v.z + a.z * remaining;
        const float sq = (ux * ux + uy * uy) + uz * uz;
        if (sq == sq) bits = __float_as_uint(sq);
    }
    bits = wave_max_u32(bits);
    if ((threadIdx.x & (WAVE - 1)) == 0 && bits) atomicMax(out_bits, bits);
}
float World::choose_substep(const StepCtx& c) {
    uint32_t* const d_bits = &d_rb.p->cfl_max_bits;
    SALVA_HIP_CHECK(hipMemsetAsync(d_bits, 0, sizeof(uint32_t), stream));
    if (n) k_cfl_max<<<nblk(n), BLOCK, 0, stream>>>(n, c.vel, c.acc, comm ? gtag[cur].p : nullptr, step_remaining, d_bits);
    SALVA_HIP_CHECK(hipGetLastError());
    float max_sq = 0.0f;
    if (comm && comm->size() > 1) {
        // max over the ranks through the sum all-reduce: every rank writes its value into its own slot of a zeroed row (x + 0 = x)
        const int size = comm->size();
        d_sums.ensure((size_t)std::max(size, 8));
        SALVA_HIP_CHECK(hipMemsetAsync(d_sums.p, 0, (size_t)size * sizeof(float), stream));
        SALVA_HIP_CHECK(hipMemcpyAsync(d_sums.p + comm->rank(), d_bits, sizeof(float), hipMemcpyDeviceToDevice, stream));
        comm->allreduce_sum_f32(d_sums.p, size, stream);
        std::vector<float> all((size_t)size, 0.0f);
        SALVA_HIP_CHECK(hipMemcpyAsync(all.data(), d_sums.p, (size_t)size * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        for (float v : all) max_sq = std::max(max_sq, v);
    } else {
        SALVA_HIP_CHECK(hipMemcpyAsync(&h_rb->cfl_max_bits, d_bits, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        wait_stream();
        memcpy(&max_sq, &h_rb->cfl_max_bits, sizeof(float));
    }
    const float total = step_total;
    const float min_substep = total / (float)cfl_max_sub, max_substep = total / (float)cfl_min_sub;
    const float computed = prm.particle_radius * 2.0f / std::sqrt(max_sq) * cfl_coeff;
    float sub = computed > max_substep ? max_substep : (computed < min_substep ? min_substep : computed);  // na::clamp
    // (mode 2: cut at the remaining time; a remainder of float residue — below 1e-4 of the step — goes along with this substep instead
    // of becoming a last one of a few ulps with inv_dt ~ 1e6; the CPU restatement the tests hold this against does the same)
    if (cfl_mode == 2 && (sub > step_remaining || step_remaining - sub < total * 1e-4f)) sub = step_remaining;
    return sub;
}

// Can this step's solves be chained (device_types.h StepCtx::gate)?  Everything between the first solve and the end of the step has
// to be a kernel that honours the gate and needs no host decision: a single domain, no CFL choice (a read-back inside the solver), no
// host force callback, no force with a solve of its own (DFSPHViscosity).  SALVA_HIP_NO_CHAIN=1 switches it off (A/B, tests).
bool World::chain_allowed() const {
    if (chain_off || comm || cfl_mode || spec_mode || prm.solver != SALVA_HIP_SOLVER_DFSPH) return false;
    for (const FluidSlot& f : fluids)
        for (const SalvaHipForceDesc& d : f.forces)
            if (d.kind == SALVA_HIP_FORCE_CUSTOM || d.kind == SALVA_HIP_FORCE_DFSPH_VISCOSITY) return false;
    return true;
}

// DFSPHSolver::step (dfsph_solver.rs:667-708)
// `resume`: 0 = the step's solver part from its start.  Chained, it returns with everything enqueued and nothing waited for
// (chain_pending); World::substep learns from the end-of-step publication whether both solves converged within their batches and,
// if one did not, calls again with resume = 1 (continue the divergence solve, then everything behind it) or 2 (continue the pressure
// solve, then the position update) — the classic way, one wait per batch.
void World::dfsph_solve(StepCtx& c, float& dt, const float g[3], SalvaHipStepStats& st, int resume) {
    if (resume) { dt_prev = chain_dt_prev; inv_dt_prev = chain_inv_dt_prev; }  // (TimestepManager's state as the chained attempt found it)
    chain_dt_prev = dt_prev; chain_inv_dt_prev = inv_dt_prev;
    // divergence_solve (:466-503).  NOTE the dt lag: inv_dt is still the previous step's here (0 on the first step).
    const float inv_dt_lag = inv_dt_prev;
    const bool timers = prm.enable_timers != 0;
    if (timers && !resume) SALVA_HIP_CHECK(hipEventRecord(evc[3], stream));  // counters.custom (:492)
    // Speculative applies (dfsph.hip, spec_decide): the convergence test rides in the apply pass instead of a launch of its own.
    // Worth ~3 us per iteration (measured: a 50-iteration step 5.13 -> 4.99 ms); the apply that follows the converging evaluate is
    // then computed in vain (~30 us once per solve), so: only when the previous step's solve ran 16 iterations or more; not with boundary reaction forces (an
    // apply that is thrown away must not have added to them) and not in decomposed runs (the test sits behind an all-reduce).
    // Decomposed runs (round 6): the same double buffer lets the apply run BESIDE the all-reduced test instead of behind it
    // (World::run_solve); there the wasted apply is cheaper than the all-reduces it hides from four iterations on.
    const bool multi_rank = comm && comm->size() > 1;
    const bool spec_apply = !resume && !spec_apply_off && !any_wants_forces &&
                            (multi_rank ? (!spec_dist_off && last_iters[0] >= 4u) : (!comm && last_iters[0] >= 16u));
    if (spec_apply) { w2.ensure(n, stream, false, 1.1f); spec_ring.ensure(2); c.w2 = w2.p; c.spec_ring = spec_ring.p; }
    // Chained: neither solve is waited for (the w / w2 swap below is a host decision on the iteration count: not with speculative applies)
    const bool chain = !resume && !spec_apply && chain_allowed();
    // ... except that the divergence solve keeps its waits while its iteration count is RISING (the impact: 2, 4, 14, 18, 30, 44
    // iterations in consecutive steps): every such step's batch would fall short, and a broken chain costs more than the wait it was
    // meant to save (the gated kernels behind it, a publication, the continuation).  The chain then starts behind it.
    const bool chain_div = chain && !(last_iters[0] > prev_iters[0]);
    StepCtx cg = c;   // the context of everything BEHIND the first chained solve: gated
    if (chain) cg.gate = &d_rb.p->chain_ok;
    const StepCtx& ca = chain_div ? cg : c;  // ... which the kernels between the two solves are only when the first one is chained
    auto div_eval = [&](const StepCtx& cc, int it) {
        if (it == 0 && fused_first_divergence) return;  // (k_density_alpha_div_p3 wrote kappa and the error partials already)
        evaluate_split(cc, it, [&](const StepCtx& cs, hipStream_t s) { launch_divergence(cs, lds, s); });
    };
    auto div_apply = [&](const StepCtx& cc, int) {
        // decomposed runs: kappa of the inner ghost plane was computed here from refreshed w — the applies of the
        // owned particles read nothing else, so only w travels, once per iteration
        launch_divergence_apply(cc, lds, inv_dt_lag, spec_dist_stream ? spec_dist_stream : stream);
        if (comm && !spec_dist_stream) refresh_f4(w.p);  // (a speculative decomposed apply: run_solve refreshes the buffer it wrote)
    };
    const float div_tol = prm.max_divergence_error * inv_dt_prev * 0.01f;
    SolveResult rd{0u, 0.0f};
    // (the pressure solve's control block is initialised by the divergence solve's launch — unless that solve has to open the chain)
    pre_init1_valid = false;
    if (resume == 0 && (!chain || chain_div)) {
        pre_init1 = SolveCtl{0u, 0u, 0.0f, 0u, prm.max_density_error, (uint32_t)std::max(prm.min_pressure_iter, 0), 0u, 0u};
        pre_init1_valid = true;
    }
    if (resume <= 1) {
        rd = run_solve(c, 0, div_tol, prm.min_divergence_iter, prm.max_divergence_iter, 0u, div_eval, div_apply, spec_apply, chain_div ? 1 : 0,
                       resume == 1 ? chain_batch[0] : 0, chain_div);
        if (spec_apply && multi_rank) {
            static const bool trace = getenv("SALVA_HIP_DIST_TRACE") != nullptr;
            if (trace) fprintf(stderr, "salva_hip dist[%d]: divergence solve with applies beside the all-reduce, %u iterations\n", comm->rank(), rd.iters);
        }
        if (spec_apply && (rd.iters & 1u)) {  // an odd number of committed applies: w lives in the second buffer
            std::swap(w.p, w2.p); std::swap(w.cap, w2.cap);
            c.w = w.p; c.w2 = w2.p; cg.w = w.p; cg.w2 = w2.p;
        }
        if (timers) SALVA_HIP_CHECK(hipEventRecord(evc[4], stream));  // :501
        st.n_divergence_iters = (int32_t)rd.iters;
        st.divergence_error = rd.err;
        launch_finish_divergence(ca, g[0], g[1], g[2], acc_user, stream);  // update_velocities + dv = 0 + gravity
        run_forces(ca);
    }
    // timestep.advance (:702): dt := total step (or, opted in, the CFL substep), inv_dt := 1/dt
    if (cfl_mode) dt = choose_substep(c);
    const float inv_dt = (dt == 0.0f) ? 0.0f : 1.0f / dt;
    if (resume <= 1) {
        launch_integrate(ca, dt, stream);
        if (comm) refresh_f4(w.p);
    }
    // pressure_solve (:432-464)
    const SolveResult rp = run_solve(
        cg, 1, prm.max_density_error, prm.min_pressure_iter, prm.max_pressure_iter, 0u,
        [&](const StepCtx& cc, int it) { evaluate_split(cc, it, [&](const StepCtx& cs, hipStream_t s) { launch_pred_density(cs, lds, dt, s); }); },
        [&](const StepCtx& cc, int) {
            launch_pressure_apply(cc, lds, inv_dt, stream);
            if (comm) refresh_f4(w.p);
        }, false, chain ? 2 : 0, resume == 2 ? chain_batch[1] : 0, chain && !chain_div);
    st.n_pressure_iters = (int32_t)rp.iters;
    st.density_error = rp.err;
    launch_update_positions(cg, dt, bbox_partials.p, nullptr, stream);  // (the per-block boxes are folded by the end-of-step publication)
    fold_bbox_blocks = n ? num_blocks(n) : 0u; fold_bbox_gate = cg.gate;
    dt_prev = dt;
    inv_dt_prev = inv_dt;
    chain_pending = chain;
    chain_div_pending = chain_div;
}

// IISPHSolver::step (iisph_solver.rs:643-711)
void World::iisph_solve(StepCtx& c, float& dt, const float g[3], SalvaHipStepStats& st) {
    st.n_divergence_iters = 0;
    st.divergence_error = 0.0f;
    launch_iisph_begin(c, g[0], g[1], g[2], acc_user, stream);
    run_forces(c);  // forces still see the previous inv_dt (:654-662)
    if (cfl_mode) dt = choose_substep(c);  // timestep.advance (:662)
    const float inv_dt = (dt == 0.0f) ? 0.0f : 1.0f / dt;
    launch_integrate(c, dt, stream);
    if (comm) refresh_f4(w.p);             // a ghost's own forces were summed over an incomplete neighbourhood
    if (!iisph_dii_fused) launch_iisph_dii(c, lds, dt, stream);  // also p = 0.5 * p_prev (a per-particle operation: right for ghosts too)
    // (d_ii depends on positions only: right on the inner ghost plane without an exchange)
    launch_iisph_pred_density(c, lds, dt, stream);
    if (!iisph_dii_fused) launch_iisph_aii(c, lds, dt, stream);  // (fused: a_ii came out of the density pass with d_ii)
    float* const pa = kappa.p;
    float* const pb = kappa2.p;
    const float omega = 0.5f;  // :53
    // pressure_solve (:422-456): iteration j reads p_j and writes p_{j+1}; the two buffers alternate (the reference swaps)
    const SolveResult rp = run_solve(
        c, 1, prm.max_density_error, prm.min_pressure_iter, prm.max_pressure_iter, 1u,
        [&](const StepCtx& cc, int j) {
            const float* pr = (j & 1) ? pb : pa;
            float* pw = (j & 1) ? pa : pb;
            launch_iisph_dij_pj(cc, lds, dt, pr, stream);
            launch_iisph_next_pressure(cc, lds, dt, omega, pr, pw, stream);
            if (comm) refresh_f32(pw);
        },
        [&](const StepCtx&, int) {});
    st.n_pressure_iters = (int32_t)rp.iters;
    st.density_error = rp.err;
    const float* p = (rp.iters & 1u) ? pb : pa;
    launch_iisph_velocity_changes(c, lds, dt, p, stream);
    launch_iisph_finish(c, dt, p, bbox_partials.p, nullptr, stream);  // (the per-block boxes are folded by the end-of-step publication)
    fold_bbox_blocks = n ? num_blocks(n) : 0u; fold_bbox_gate = nullptr;
    dt_prev = dt;
    inv_dt_prev = inv_dt;
}

// ------------------------------------------------------------------------------------------------ step
int World::step(float dt, const float g[3], SalvaHipStepStats* stats) {
    use_device();
    SalvaHipStepStats st{};
    st.nparticles = n;
    {   // self.counters.reset() (liquid_world.rs:73); the pass counters of this implementation are cumulative
        const uint64_t sp = counters.speculative_passes, dp = counters.discarded_passes;
        const uint64_t keep4[6] = {counters.chained_passes, counters.chain_breaks, counters.pregrid_adopted, counters.pregrid_dropped,
                                   counters.light_class_passes, counters.sparse_class_passes};
        counters = SalvaHipCounters{};
        counters.speculative_passes = sp; counters.discarded_passes = dp;
        counters.chained_passes = keep4[0]; counters.chain_breaks = keep4[1]; counters.pregrid_adopted = keep4[2]; counters.pregrid_dropped = keep4[3];
        counters.light_class_passes = keep4[4]; counters.sparse_class_passes = keep4[5];
    }
    sticky.clear();  // init_with_fluids runs at the top of every step, substeps or not (liquid_world.rs:76)
    substeps.clear();  // (also for a step that runs no substep at all: salva_hip_get_substeps then agrees with counters.nsubsteps = 0)
    // TimestepManager::is_done (timestep_manager.rs:56-58): no substep at all for dt <= eps
    if ((n == 0 && !comm) || !(dt > FLT_EPSILON)) {
        if (stats) *stats = st;
        return SALVA_HIP_OK;
    }
    if (comm && acc_user)
        throw HipError(SALVA_HIP_E_INVALID, "host-set accelerations (SALVA_HIP_DIRTY_ACCELERATIONS) are not carried through the slab decomposition: "
                                            "apply them as velocity changes, or use a single domain");
    // `while !self.timestep_manager.is_done()` (liquid_world.rs:85): one substep of the whole step as the reference runs today
    // (compute_substep returns total_step_size, timestep_manager.rs:88); with salva_hip_set_cfl the clamp the reference left
    // commented out (:90-93) decides each substep's length inside the solver (`timestep.advance`, dfsph_solver.rs:702).
    substeps.clear();
    step_total = dt;
    step_remaining = dt;
    if (cfl_mode) {
        upload_tables();  // (any_wants_forces)
        if (any_wants_forces && !coupling_cb)
            for (const BoundarySlot& b : bounds)
                if (b.wants_forces && (b.sampling || b.dyn_kind))
                    // the reference clears a coupled boundary's forces and transmits their impulse in EVERY substep with that substep's
                    // dt (fluids_pipeline.rs:262, :266-287): one wrench per step() cannot carry that.  The caller's own loop can.
                    throw HipError(SALVA_HIP_E_INVALID, "CFL sub-stepping with a coupled boundary that wants forces needs salva_hip_set_coupling_callback "
                                                        "(update_boundaries / transmit_forces per substep, as the reference's manager does), or salva_hip_set_cfl off");
    }
    while (!(step_remaining <= FLT_EPSILON)) {  // is_done, timestep_manager.rs:56-58
        float used = dt;
        int rc;
        try {
            call_coupling(0, dt_prev);  // coupling.update_boundaries(&timestep, ...) (liquid_world.rs:94-103): dt() is the last substep's
            for (;;) {
                try { rc = substep(used, g, st); break; }
                catch (const FoldRetry&) { used = dt; pre.valid = false; flags_clean = false; }  // (the grid changes: again from the top)
            }
            if (rc == SALVA_HIP_OK) call_coupling(1, used);  // coupling.transmit_forces(&timestep, boundaries) (:146): this substep's dt
        } catch (...) {
            if (stats) *stats = st;  // (what the failed substep got to: the caller's report is filled either way)
            throw;
        }
        substeps.push_back(used);
        step_remaining -= used;  // advance, :84
        ++counters.nsubsteps;
        if (rc != SALVA_HIP_OK) { if (stats) *stats = st; return rc; }
        if (substeps.size() > 4096) throw HipError(SALVA_HIP_E_INVALID, "internal error: the substep loop does not terminate");
    }
    if (stats) *stats = st;
    return SALVA_HIP_OK;
}

// The grid part of the NEXT step, behind this step's end-of-step publication (world.h PreGrid): the same launches World::substep
// makes at its start — cell keys + counts, the counting sort by cell, the non-empty tiles, the per-tile counts, their scan, the
// publication of the totals — on the positions this step leaves behind, for the grid this step ran on, into the OTHER set of
// tables, every kernel gated by Readback::pre_ok.
void World::pre_enqueue_grid(uint32_t nslots_bound) {
    GridTabs& T = gtab[gsel ^ 1];
    const size_t ncf = gf.ncells();
    const uint32_t ntiles = (uint32_t)gf.ntiles();
    const uint32_t* gate = &d_rb.p->pre_ok;
    T.cell_start_f.ensure(ncf + 1, stream, false, 1.5f);
    T.cell_rank.ensure(n, stream, false, 1.1f);
    T.tile_flags.ensure((size_t)ntiles + 1, stream, false, 1.5f);
    T.tile_rank.ensure((size_t)ntiles + 1, stream, false, 1.5f);
    T.tile_ids.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
    T.slot_desc.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
    T.tile_cnt.ensure((size_t)nslots_bound + 1, stream, false, 1.5f);
    T.tile_off.ensure((size_t)nslots_bound + 1, stream, false, 1.5f);
    pre.check_mass = !(mass_known && mass_uniform == 0.0f);
    SALVA_HIP_CHECK(hipMemsetAsync(T.cell_start_f.p, 0, (ncf + 1) * sizeof(uint32_t), stream));
    launch_cell_keys(posm[cur].p, n, sc.h, gf.device(nullptr), T.keys[0].p, T.idx[0].p, d_flags.p, pre.check_mass ? mass_slots.p : nullptr,
                     T.cell_start_f.p, T.cell_rank.p, stream, gate);
    {
        const size_t tb = cell_sort_temp_bytes((uint32_t)ncf);
        ensure_cub_temp(tb);
        cell_sort(cub_temp.p, tb, n, (uint32_t)ncf, T.keys[0].p, T.cell_rank.p, T.cell_start_f.p, T.keys[1].p, T.idx[0].p, T.idx[1].p, stream, gate);
    }
    {
        const size_t tb = std::max(scan_tiles_temp_bytes(nslots_bound + 1), scan_temp_bytes(ntiles + 1));
        ensure_cub_temp(tb);
        launch_tile_slots(gf.device(T.cell_start_f.p), ntiles, T.tile_flags.p, T.tile_rank.p, T.tile_ids.p, cub_temp.p, tb, stream, gate, split_s_cur);
        StepCtx cp = make_ctx();
        cp.gf = gf.device(T.cell_start_f.p);
        cp.tile_off = T.tile_off.p; cp.tile_ids = T.tile_ids.p; cp.tile_rank = T.tile_rank.p; cp.slot_desc = T.slot_desc.p;
        cp.gate = gate;
        launch_tile_count(cp, nslots_bound, T.tile_cnt.p, T.slot_desc.p, stream);
        scan_tiles(cub_temp.p, tb, T.tile_cnt.p, T.tile_off.p, nslots_bound + 1, stream);
    }
    pre.seq = publish_enqueue(T.tile_off.p + nslots_bound, false, false, nullptr, gate);
    pre.n = n; pre.ncf = ncf; pre.ntiles = ntiles; pre.nslots_bound = nslots_bound; pre.gf = gf; pre.split_s = split_s_cur;
    pre.valid = true;
}
// The next step could not use it.  If its launches ran (the gate was open) they have raised flags and mass marks that belong to no step.
void World::pre_drop() {
    ++pre_dropped; ++counters.pregrid_dropped;
    if (!h_rb->pre_ok) return;
    SALVA_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, sizeof(uint32_t), stream));
    SALVA_HIP_CHECK(hipMemsetAsync(mass_slots.p, 0, MASS_SLOTS * sizeof(uint32_t), stream));
}

void World::call_coupling(int phase, float dt) {
    if (!coupling_cb) return;
    if (coupling_cb(coupling_user, coupling_owner, phase, dt) != 0)
        throw HipError(SALVA_HIP_E_INVALID, "the coupling callback reported an error");
}

// One substep: the body of the `while` of LiquidWorld::step_with_coupling (liquid_world.rs:85-147).  `dt` comes in as the step's
// total length and goes out as the substep the solver advanced by.
int World::substep(float& dt, const float g[3], SalvaHipStepStats& st) {
    const bool timers = prm.enable_timers != 0;
    if (timers) SALVA_HIP_CHECK(hipEventRecord(ev[0], stream));
    // (what decides below whether the grid part the previous step enqueued for this one still describes the world)
    const bool world_touched = tables_dirty || !sorted_valid || !bbox_known || b_dirty;
    upload_tables();
    // (the end-of-step publication of the previous step left the flags clear; anything else — the first step, a step that threw —
    // clears them here)
    if (!flags_clean) SALVA_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, sizeof(uint32_t), stream));
    flags_clean = false;

    // ---- persistent particle arrays (double buffered for the sort)
    ensure_particle_capacity(n);

    // ---- (re)build the sorted working set from the canonical arrays after host edits
    if (!sorted_valid) {
        mass_known = false;  // (the host edited the particles: their masses are read again by this step's k_cell_keys)
        launch_stage_to_sorted(n, st_pos.p, st_vel.p, st_dv.p, st_model.p, rho0_tab.p, arrays(cur), stream);
        sorted_valid = true;
        if (comm) {  // global particle ids replace the host-order permutation; nothing is a ghost yet
            launch_iota_u32(n, gid_offset, perm[cur].p, stream);
            SALVA_HIP_CHECK(hipMemsetAsync(gtag[cur].p, 0, (size_t)n * sizeof(uint32_t), stream));
        }
    }
    staging_current = false;
    if (comm) { dist_prepare(); st.nparticles = n_owned; }  // migration + ghost planes: changes n

    // ---- per-step scratch
    acc.ensure(n, stream, false, 1.1f); w.ensure(n, stream, false, 1.1f); rho.ensure(n, stream, false, 1.1f);
    posmr.ensure(n, stream, false, 1.1f);
    alpha.ensure(n, stream, false, 1.1f); kappa.ensure(n, stream, false, 1.1f); nff.ensure(n, stream, false, 1.1f);
    nfb.ensure(n, stream, false, 1.1f);
    bool has_akinci = false;
    for (auto& f : fluids) for (auto& d : f.forces) has_akinci |= d.kind == SALVA_HIP_FORCE_AKINCI2013;
    if (has_akinci) normal.ensure(n, stream, false, 1.1f);
    if (prm.solver == SALVA_HIP_SOLVER_IISPH) { kappa2.ensure(n); rho_star.ensure(n); aii.ensure(n); dii.ensure(n); dijpj.ensure(n); iisph_q.ensure(n); iisph_pr.ensure(n); }
    bbox_partials.ensure(6 * std::max<size_t>(std::max<size_t>(num_blocks(n), bbox_blocks(n)), 1024));
    two_mass = decide_two_mass();
    if (two_mass) nffb.ensure(n, stream, false, 1.1f);
    if (two_mass && nmass > 2u) nffc.ensure(n, stream, false, 1.1f);

    // ---- cell bounding box (known from the previous step's position update unless the host moved particles)
    if (!bbox_known) {
        launch_bbox(posm[cur].p, n, sc.h, bbox_partials.p, d_rb.p->bbox, d_flags.p, stream);
        SALVA_HIP_CHECK(hipMemcpyAsync(h_rb->bbox, d_rb.p->bbox, sizeof(int32_t) * 6, hipMemcpyDeviceToHost, stream));
        wait_stream();
        bbox_known = true;
    }
    // Fold the fluid grid when its box is mostly empty (device_types.h TileGrid): more than 4 cells per particle + 2^20 — a block at
    // rest fills a cell with eight particles, the box of an L-shaped or splashing scene a few times the cells it occupies; leaked or
    // sprayed particles falling away from the scene are what this is for (tools/r05/soak.sh: the bench scene's box grows from 1.2 x 10^5
    // to 2.3 x 10^8 cells in a thousand steps).  The periods stay at least 64 cells and at least as long as the boundary grid is wide
    // (a folded fluid tile addresses the boundary cells modulo its own periods: tile.h TileCells::build), so the boundary grid has
    // to exist first.  Not in decomposed runs (ghost planes are found by absolute cell coordinate) and not with dynamic contact
    // sampling (dcs.hip decodes cell coordinates from the keys).  SALVA_HIP_NO_FOLD=1: never; SALVA_HIP_FOLD_CELLS=P: every axis
    // longer than P cells to exactly P (a power of two >= 8; the tests' way to fold small scenes).
    {
        // (fold_relax: a fold that piled the bulk of the fluid onto itself — the rule below looks at the box, not at where the particles
        // sit — was found out by the tile totals of an earlier attempt, which then threw FoldRetry: every level loosens the fold
        // eightfold, the third gives it up.  Sticky for the world.)
        const uint32_t forced = fold_forced ? fold_forced << (2u * std::min(fold_relax, 3u)) : 0u;
        constexpr bool tiles_pow2 = (TX & (TX - 1)) == 0 && (TY & (TY - 1)) == 0 && (TZ & (TZ - 1)) == 0;  // (a period is a whole number of tiles)
        // (a decomposed run finds its ghost planes by absolute x-cell: it folds y and z only.  Dynamically sampled colliders read cells
        // back from the keys: dcs.hip picks the image by the particle's position.)
        const bool can_fold = tiles_pow2 && !fold_off && fold_relax < 3u;
        if (can_fold && nb && b_dirty) build_boundary_grid();
        // (once it folds, it folds tight — to half a cell per particle if the periods allow: the particles that have left the scene
        // then land on the tiles of the bulk instead of owning a tile each; a tile with one particle costs a quarter of a full one)
        const double loosen = (double)(1u << (3u * std::min(fold_relax, 3u)));
        FoldRule rule{forced ? 0.0 : 4.0 * (double)n + 1048576.0, forced ? 0.0 : std::max(0.5 * (double)n, 262144.0) * loosen,
                      {forced ? forced : 64u, forced ? forced : 64u, forced ? forced : 64u}, {!comm, true, true}};
        if (nb)
            for (int a = 0; a < 3; ++a) {
                static const int T[3] = {TX, TY, TZ};
                const uint64_t bcells = (uint64_t)gb.nt[a] * T[a];
                while (rule.min_period[a] < bcells) rule.min_period[a] *= 2u;
            }
        try {
            dims_from_bbox(h_rb->bbox, gf, can_fold ? &rule : nullptr);
        } catch (const HipError& e) {
            // the looser fold does not fit the cell-table budget: back to the tighter one, and live with what its tiles hold
            if (e.code != SALVA_HIP_E_CAPACITY || fold_relax == 0u || fold_locked) throw;
            --fold_relax; fold_locked = true;
            throw FoldRetry{};
        }
    }
    const size_t ncf = gf.ncells();
    const uint32_t ntiles = (uint32_t)gf.ntiles();
    // only the cell table and one flag per tile are dense over the bounding box; every other per-tile table is compact
    // over the non-empty tiles ("slots"), of which there are at most min(ntiles, n)
    // Splitting of over-full tiles (device_types.h StepCtx::split_s): where the plane layouts run (one or two particle masses, single
    // domain), and while the tiles beyond the three-per-CU layouts are a minority — decided from the totals of the step before
    // (split_on, below).  SALVA_HIP_NO_SPLIT=1: never; SALVA_HIP_SPLIT_S=k: always, at k halo particles (tests).
    split_s_cur = 0u;
    if (split_forced) split_s_cur = split_forced;
    else if (!split_off && split_on && !comm && spec_off && mass_known && (mass_uniform != 0.0f || two_mass)) split_s_cur = TILE_SPLIT_S;
    const uint32_t nslots_bound = (uint32_t)std::min<uint64_t>((uint64_t)ntiles * (split_s_cur ? (uint64_t)TX : 1ull), n);
    // ---- the grid part of this step may be on the device already (world.h PreGrid): adopt the other set of tables, or drop it
    bool adopted = false;
    if (pre.valid) {
        pre.valid = false;
        adopted = h_rb->pre_ok && !world_touched && !timers && !comm && pre.n == n && pre.ncf == ncf && pre.ntiles == ntiles &&
                  pre.nslots_bound == nslots_bound && pre.split_s == split_s_cur && memcmp(&pre.gf, &gf, sizeof(GridDims)) == 0;
        if (adopted) { gsel ^= 1; ++pre_adopted; ++counters.pregrid_adopted; }
        else pre_drop();
    }
    G().cell_start_f.ensure(ncf + 1, stream, false, 1.5f);
    G().tile_flags.ensure((size_t)ntiles + 1, stream, false, 1.5f);
    G().tile_rank.ensure((size_t)ntiles + 1, stream, false, 1.5f);
    G().tile_ids.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
    G().slot_desc.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
    slot_info.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
    G().tile_cnt.ensure((size_t)nslots_bound + 1, stream, false, 1.5f);
    G().tile_off.ensure((size_t)nslots_bound + 1, stream, false, 1.5f);
    d_maxhalo.ensure(4);
    partials.ensure((size_t)std::max<uint32_t>(nslots_bound, 1u) * std::max<size_t>(fluids.size(), 1), stream, false, 1.5f);
    // every tile wastes less than one 64-particle slice
    const uint32_t ns_cap = n / WAVE + nslots_bound + 1;
    tile_list_stats.ensure(tile_list_stats_bytes(std::max<uint32_t>(nslots_bound, 1u)), stream, false, 1.5f);
    if (two_mass) {  // (per slot; Tile::setup reads them in every tile kernel, k_nbr_tile writes them)
        tile_mass_bits.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
        tile_massb_bits.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
        if (nmass > 2u) tile_masscd_bits.ensure(std::max<uint32_t>(nslots_bound, 1u), stream, false, 1.5f);
    }

    // ---- One pass over the step.  The sizes of the tile tables (number of non-empty tiles, largest halo, slices) and the
    // longest neighbour list are results of this step's own kernels; waiting for them costs two host round trips with an
    // idle GPU.  When nothing forbids it the step is SPECULATIVE: launch shapes, LDS sizes and buffers are taken from the
    // previous step's totals plus a margin, the kernels clamp themselves to what they were given, and the true totals are
    // read back once, at the end, with everything else.  If they exceeded the prediction (rare: they change by a few
    // slots per step) the step is simply run again from the untouched pre-sort buffers with exact sizes.
    bool has_custom = false;
    for (auto& f : fluids) for (auto& d : f.forces) has_custom |= d.kind == SALVA_HIP_FORCE_CUSTOM;
    const bool has_dyn = has_dynamic_sampling();
    double dcs_ms = 0.0;
    // a pass can be repeated from the untouched pre-sort buffers iff it has no side effect outside the world's own arrays
    // (a communicator of one rank exchanges nothing: its passes are as repeatable as the plain world's)
    const bool solo = comm && !comm->has_lo() && !comm->has_hi();
    const bool can_redo = (!comm || solo) && !any_wants_forces && !has_custom && !has_dyn;
    // (mass_known: the kernels of a pass are chosen by StepCtx::mass_uniform, which a speculative pass — it does not wait for the
    // publication that carries it — can only inherit; a host edit since the last publication may have changed the masses)
    const bool can_speculate = !spec_off && can_redo && !b_dirty && pred_valid && pred_n == n && mass_known;
    // The neighbour-list capacity check (longest list <= ELL capacity) costs a read-back with an idle GPU in the middle of the
    // step although it fails about once per run (the capacity follows the longest list seen so far): where the pass can be
    // repeated, check at the end of the step with the read-back that happens there anyway, and repeat on overflow.

    // It stays in the middle of the step until the capacity has held once for the current set of objects (`lists_checked`,
    // cleared by every edit of the fluids or boundaries): the first step of a scene — whose lists are longer than the
    // initial capacity in almost any dense scene — is then not computed twice.
    //
    // State a pass touches, and where a discarded pass leaves it (extend this list with every new side effect, or exclude the
    // feature in can_redo):
    //   particle arrays of buffer cur^1 (the sort's output: positions, velocities, dv / pressures, models, permutation) ....
    //       recomputed by the repeated pass; buffer `cur` (the pre-sort state) is read-only until the pass commits
    //   cur, dt_prev / inv_dt_prev, h_rb->bbox, last_iters ........ snapshot below, restored on discard
    //   per-step scratch (acc, w, rho, alpha, kappa, lists, tables, partials, control blocks) ........ rewritten from the start
    //   d_flags ........ cleared at the top of every attempt
    //   counters / stats ........ filled after the loop; discarded_passes counts the discards
    //   boundary force accumulators, host force callbacks, DynamicContactSampling push-outs, ghost exchanges ........ not
    //       repeatable: can_redo is false for worlds that have them
    int32_t bbox_pre[6];
    memcpy(bbox_pre, h_rb->bbox, sizeof(bbox_pre));
    const float dt_prev0 = dt_prev, inv_dt_prev0 = inv_dt_prev;
    const int cur0 = cur;
    uint32_t last_iters0[NUM_SOLVES], prev_iters0[NUM_SOLVES];
    memcpy(last_iters0, last_iters, sizeof(last_iters0));
    memcpy(prev_iters0, prev_iters, sizeof(prev_iters0));
    StepCtx c{};
    for (int attempt = 0;; ++attempt) {
    bool spec = can_speculate && attempt == 0;
    chain_pending = false;
    const bool defer_lists = can_redo && !defer_off && attempt == 0 && lists_checked;
    if (attempt > 0) SALVA_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, sizeof(uint32_t), stream));  // (whatever the discarded pass flagged)
    const bool have_grid = adopted && attempt == 0;  // keys, sort, tile tables and the totals publication are enqueued already
    // (a scene known to hold different masses — two fluids of different density0 — is not asked again until the host edits the
    // particles: every wave of the lighter fluid would raise a flag, 71 us per launch at 2 x 10^6 particles)
    check_mass = have_grid ? pre.check_mass : !(mass_known && mass_uniform == 0.0f);
    // ---- grid: keys -> radix sort -> reorder -> cell table   (hgrid.clear + insert_fluids_to_grid, liquid_world.rs:90-91)
    if (!have_grid) {
        TileGrid gv = gf.device(nullptr);
        // The sort is a counting sort by cell (grid.hip cell_sort: same result as the radix sort it replaced, bit for bit) while the
        // cell table is not much larger than the particle set — it costs a memset and a scan of that table, where the radix sort
        // with k_cell_start writes it once: a bounding box blown up by a few strays (4 x 10^8 cells around 25 k particles) keeps
        // the radix sort.  SALVA_HIP_RADIX_SORT=1 / 0 forces one or the other.
        const bool counting = ncf + 1 < 0x7fffffffull && (sort_mode == 0 || (sort_mode < 0 && ncf <= 16ull * n + (1ull << 20)));
        if (counting) {
            G().cell_rank.ensure(n, stream, false, 1.1f);
            SALVA_HIP_CHECK(hipMemsetAsync(G().cell_start_f.p, 0, (ncf + 1) * sizeof(uint32_t), stream));
        }
        launch_cell_keys(posm[cur].p, n, sc.h, gv, G().keys[0].p, G().idx[0].p, d_flags.p, check_mass ? mass_slots.p : nullptr,
                         counting ? G().cell_start_f.p : nullptr, counting ? G().cell_rank.p : nullptr, stream);
        if (has_dyn) {  // coupling.update_boundaries (liquid_world.rs:94-103): may push particles, cells stay
            // (host clock: the pass ends with a read-back of the emitted count, so the stream is drained when it returns)
            if (timers) wait_stream();
            const auto t1 = std::chrono::steady_clock::now();
            run_dynamic_sampling();
            dcs_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        }
        if (counting) {
            const size_t tb = cell_sort_temp_bytes((uint32_t)ncf);
            ensure_cub_temp(tb);
            cell_sort(cub_temp.p, tb, n, (uint32_t)ncf, G().keys[0].p, G().cell_rank.p, G().cell_start_f.p, G().keys[1].p, G().idx[0].p, G().idx[1].p, stream);
        } else {
            const int end_bit = bits_for(ncf);
            const size_t tb = sort_pairs_temp_bytes(n, end_bit);
            ensure_cub_temp(tb);
            sort_pairs(cub_temp.p, tb, G().keys[0].p, G().keys[1].p, G().idx[0].p, G().idx[1].p, n, end_bit, stream);
            launch_cell_start(G().keys[1].p, n, (uint32_t)ncf, G().cell_start_f.p, stream);
        }
    }
    // The particle arrays are permuted into the sorted order AFTER the tile tables have been counted: those need the cell table
    // only, and the host then waits for their totals while the GPU moves the 136 bytes per particle of the reorder.
    auto reorder = [&]() {
        launch_reorder_fluid(n, G().idx[1].p, arrays(cur), arrays(cur ^ 1), w.p, stream);
        cur ^= 1;
        if (timers) SALVA_HIP_CHECK(hipEventRecord(evc[0], stream));
        if (acc_user && !comm) launch_gather_f4(n, perm[cur].p, st_acc.p, acc.p, stream);
        if (comm) dist_build_lists();
    };
    build_boundary_grid();  // insert_boundaries_to_grid (liquid_world.rs:106) + boundary volumes, only when dirty
    if (timers) SALVA_HIP_CHECK(hipEventRecord(evc[1], stream));

    // ---- tile tables: per-tile halo sizes / slice counts -> prefix -> flat halo slot tables
    nlaunch = 0;  // not known yet
    spec_mode = false; halo_cap = bhalo_cap = 0xffffffffu; nslices_cap = 0xffffffffu; halo_len = bhalo_len = ~0ull;
    c = make_ctx();
    TileAcc tt{};  // the totals the launch shapes and buffers of this pass are cut for
    uint32_t nslices = 0;
    {
        const size_t tb = std::max(scan_tiles_temp_bytes(nslots_bound + 1), scan_temp_bytes(ntiles + 1));
        ensure_cub_temp(tb);
        if (!have_grid) {
        launch_tile_slots(gf.device(G().cell_start_f.p), ntiles, G().tile_flags.p, G().tile_rank.p, G().tile_ids.p, cub_temp.p, tb, stream, nullptr, split_s_cur);
        // (k_tile_count zeroes the entries of its surplus workgroups and the scan's extra element itself: no memset —
        // unless there is no workgroup at all, a rank that holds no particle)
        if (nslots_bound == 0) SALVA_HIP_CHECK(hipMemsetAsync(G().tile_cnt.p, 0, sizeof(TileAcc), stream));
        launch_tile_count(c, nslots_bound, G().tile_cnt.p, G().slot_desc.p, stream);
        scan_tiles(cub_temp.p, tb, G().tile_cnt.p, G().tile_off.p, nslots_bound + 1, stream);
        }
        if (spec) {
            // previous totals + margin; the LDS must hold the padded halo (else: no speculation this step)
            const TileAcc& l = pred_tt;
            tt = l;
            const uint32_t m = spec_tight ? 0u : 1u;  // (SALVA_HIP_SPEC_TIGHT: no margin at all — the tests' way to force misses)
            tt.nonempty = std::min<uint32_t>(nslots_bound, l.nonempty + m * std::max<uint32_t>(16u, l.nonempty / 16u));
            tt.max_s = (l.max_s + m * std::max<uint32_t>(32u, l.max_s / 16u) + m * 63u) & ~(m * 63u);
            tt.max_sb = nb ? ((l.max_sb + m * std::max<uint32_t>(32u, l.max_sb / 8u) + m * 63u) & ~(m * 63u)) : 0u;
            tt.nsl = std::min<uint32_t>(n / WAVE + nslots_bound + 1, l.nsl + m * std::max<uint32_t>(64u, l.nsl / 32u));
            tt.s = l.s + m * (l.s / 8 + 4096); tt.sb = l.sb + m * (l.sb / 4 + 4096);
            TileLds probe; probe.max_halo_fluid = tt.max_s; probe.max_halo_boundary = tt.max_sb;
            if (probe.bytes(52, 32, 6) > 160u * 1024u || tt.max_s >= 65536u || tt.max_sb >= 65536u) spec = false;
        }
        if (!spec) {
            const uint32_t seq = have_grid ? pre.seq : publish_enqueue(G().tile_off.p + nslots_bound, false, false);
            reorder();
            publish_wait(seq, true, false, false);
            tt = h_rb->tile_total;
            if (check_mass) {  // every particle of the working set has the same mass: the evaluate kernels stage 24 bytes per halo slot
                float m;
                memcpy(&m, &h_rb->mass_mm[0], sizeof(m));
                mass_uniform = (h_rb->mass_mm[0] == h_rb->mass_mm[1] && m > 0.0f && std::isfinite(m) && !no_planes) ? m : 0.0f;
                mass_known = true;
            }
        } else {
            reorder();  // (mass_uniform: what the last exact pass found — nothing has touched the particles since, see can_speculate)
        }
        nlaunch = tt.nonempty;
        lds.max_halo_fluid = tt.max_s;
        lds.max_halo_boundary = tt.max_sb;
        lds.max_sum = spec ? 0u : tt.max_sum;  // (a speculative pass knows the two maxima only: TileLds::sum_slots falls back to their sum)
        lds.max_raw = spec ? 0u : tt.max_raw;
        if (tile_trace)
            fprintf(stderr, "salva_hip tiles: nonempty %u max_s %u max_sb %u max_sum %u max_raw %u heavy %u light %u tiny %u split_s %u mass_uniform %g | chained %llu breaks %llu pregrid %llu dropped %llu\n",
                    tt.nonempty, tt.max_s, tt.max_sb, tt.max_sum, tt.max_raw, tt.heavy, tt.nlight, tt.ntiny, split_s_cur, (double)mass_uniform,
                    (unsigned long long)chain_steps, (unsigned long long)chain_breaks, (unsigned long long)pre_adopted, (unsigned long long)pre_dropped);
        if (!spec) {
            // next step's splitting: on while the over-full tiles are few (each costs a second workgroup and a third more staging, and
            // buys every other tile of every pass its third resident neighbour); off again when they are the rule — a uniformly
            // compressed fluid is better served by the two-tiles-per-CU layouts than by twice the tiles (smaller
            // tiles lose).  `heavy` counts whole over-full tiles, or — while splitting — the parts they were cut into.
            if (!split_on) split_on = tt.heavy > 0u && (uint64_t)tt.heavy * 5u <= tt.nonempty;
            else if ((uint64_t)tt.heavy * 2u > tt.nonempty) split_on = false;
        }
        // Workgroup size: one wave per 64-particle slice of the tile the average PARTICLE lives in (TileAcc::wsl / nsl; fuller tiles
        // loop over their extra slices) — the plain average over the tiles drops to four waves as soon as a few thousand stray
        // particles own a tile each, and the full tiles, where nearly all the work is, then run on half the waves.  Never fewer waves
        // than the halo-table build needs threads, and never more than EIGHT: the neighbour-sum kernels hold 80 VGPRs so that three
        // tiles of eight waves share a CU (24 of its waves), k_nbr_tile 64 for four; a ninth wave per tile costs each of them a
        // whole resident tile.  Where the fluid is compressed past 512 particles per tile — every bench scene, from the impact on —
        // round 4's limit of twelve waves did exactly that: 6.2 against 4.9 ms per step in steps 100..150 of config 2, 3.2 against
        // 2.5 in steps 150..200 (profiles/r05_experiments/r05l_waves_ab.log; seven waves lose to eight there, and six / seven to
        // eight on the block at rest: 1.31 / 1.32 against 1.20 ms).  The particle-weighted size lost while the limit was twelve
        // (r05k_fold_ab.log, session 1: it asked for nine and ten waves) and wins under the limit of eight (r05m_weighted_ab.log:
        // 1000 steps of config 2 2.73 -> 2.58 ms per step, 500 of config 3 2.84 -> 2.41, 300 of config 4 9.98 -> 9.57).
        {
#ifndef SALVA_TILE_WAVES_CAP
#define SALVA_TILE_WAVES_CAP 8
#endif
            static_assert(SALVA_TILE_WAVES_CAP <= TILE_MAX_WAVES, "the launch bounds are written for TILE_MAX_WAVES");
            const uint32_t avg = (n + tt.nonempty - 1) / std::max<uint32_t>(tt.nonempty, 1u);
            const uint32_t lo = (HCELLS + WAVE - 1) / WAVE, hi = std::min<uint32_t>(std::max<uint32_t>(tt.max_nsl, lo), (uint32_t)(SALVA_TILE_WAVES_CAP));
#ifdef SALVA_TILE_WAVES_PLAIN_AVERAGE  // (A/B: round 4's rule)
            const uint32_t weighted = 0u;
#else
            const uint32_t weighted = tt.nsl ? (uint32_t)((double)tt.wsl / (double)tt.nsl + 0.5) : 0u;
#endif
            lds.threads = WAVE * std::min<uint32_t>(std::max<uint32_t>(std::max<uint32_t>((avg + WAVE - 1) / WAVE, weighted), lo), hi);
        }
#ifdef SALVA_HIP_DIAG
        if (const char* e = getenv("SALVA_HIP_TILE_THREADS")) lds.threads = (uint32_t)atoi(e);
#endif
        if (gf.folded() && !fold_locked && !spec) {
            // Did the fold pile the fluid onto itself?  (a long sheet of fluid without an enclosing boundary folds onto its
            // own bulk, cells hold several times the particles, and a scene that ran fine on the unfolded table dies of "halo does
            // not fit".)  The totals say so before any solver kernel has run: loosen the fold and run the pass again — the
            // particle arrays are merely permuted so far.
            TileLds probe; probe.max_halo_fluid = tt.max_s; probe.max_halo_boundary = tt.max_sb; probe.max_sum = tt.max_sum;
            if (tt.max_s >= 65536u || tt.max_sb >= 65536u || probe.bytes(52, 32, 6) > 160u * 1024u) {
                ++fold_relax;
                throw FoldRetry{};
            }
        }
        if (lds.max_halo_fluid >= 65536u || lds.max_halo_boundary >= 65536u)
            throw HipError(SALVA_HIP_E_CAPACITY, "more than 65535 particles in one tile halo");
        if (tt.nsl > ns_cap) throw HipError(SALVA_HIP_E_HIP, "internal error: slice count exceeds its bound");
        // slot tables: one fixed-stride row per tile when that costs at most ~3x the compact size (dense scenes), so
        // that a tile kernel can fetch its rows before it knows its sizes; compact rows otherwise (sparse scenes)
        {
            const uint64_t st_f = (lds.max_halo_fluid + 63u) & ~63u, st_b = nb ? ((lds.max_halo_boundary + 63u) & ~63u) : 0u;
            const uint64_t strided = (uint64_t)nlaunch * (st_f + st_b), compact = tt.s + tt.sb;
            const bool use = strided <= 3 * compact + (1u << 20) && !getenv("SALVA_HIP_COMPACT_HALO");
            halo_stride = use ? (uint32_t)st_f : 0u;
            bhalo_stride = use ? (uint32_t)st_b : 0u;
        }
#ifdef SALVA_HIP_DIAG
        // persistent pipeline kernels (pipe.h): one wave per slice of the fullest tile, at most PIPE_MAX_WAVES
        {
            pipe.enabled = halo_stride > 0 && !getenv("SALVA_HIP_NO_PIPELINE");
            pipe.scap = halo_stride;
            pipe.sbcap = bhalo_stride;
            pipe.num_cus = (uint32_t)num_cus;
            pipe.nlaunch = nlaunch;
            uint32_t waves = std::min<uint32_t>(std::max<uint32_t>(tt.max_nsl, 4u), (uint32_t)PIPE_MAX_WAVES);
            if (const char* e = getenv("SALVA_HIP_PIPE_WAVES")) waves = std::min<uint32_t>(std::max(atoi(e), 1), PIPE_MAX_WAVES);
            pipe.threads = waves * WAVE;
        }
#endif
        const size_t need_f = halo_stride ? (size_t)nlaunch * halo_stride : (size_t)tt.s;
        const size_t need_b = halo_stride ? (size_t)nlaunch * bhalo_stride : (size_t)tt.sb;
        halo_src.ensure(need_f ? need_f : 1, stream, false, 1.2f);
        bhalo_src.ensure(need_b ? need_b : 1, stream, false, 1.2f);
        nslices = tt.nsl;
        if (spec) {  // what the kernels clamp themselves to (StepCtx::spec)
            spec_mode = true;
            halo_cap = lds.max_halo_fluid; bhalo_cap = lds.max_halo_boundary; nslices_cap = nslices;
            halo_len = need_f; bhalo_len = need_b;
        }
        // Launch classes (device_types.h StepCtx::slot_order).  Sparse slots: worth a launch of their own per pass once they are many —
        // a thousand of them hold a CU's LDS for a round and a third of the chip each pass.  Light slots: the class the round-5 review
        // asked for (tiles under the three-per-CU limit in one launch, those above it in another) — built, bit-identical, and a
        // LOSS on the scene it was meant for: in steps 300-399 of config 2 (1015 full tiles, 860 light ones) 2.35 against 2.00 ms per
        // step, 2.71 against 2.36 in steps 900-999 (profiles/r06_experiments/r06j_light_class_lost.log) — a second launch per pass
        // ends in a second tail of straggling tiles (~13 us per pass here), which is more than the third resident tile gives back
        // to the light half.  Opt-in: SALVA_HIP_LIGHT=1 (when the fullest halo is beyond a three-per-CU layout and the light slots
        // are >= 256).  SALVA_HIP_NO_CLASSES=1: no class ever; SALVA_HIP_CLASSES=1: both, whenever there is a slot of the kind and
        // one outside it (tests).
        class_ntiny = class_nlight = 0u;
        if (!spec && !classes_off) {
            if (tt.ntiny > 0u && tt.ntiny < tt.nonempty && (classes_forced || tt.ntiny >= 512u)) class_ntiny = tt.ntiny;
            const bool beyond = tt.max_s > P3_DS_THREE || tt.max_raw > P2_DS_THREE || tt.max_sum > FIXED_DS_SMALL;
            const uint32_t nfull = tt.nonempty - tt.nlight - tt.ntiny;
            if (tt.nlight > 0u && nfull > 0u && (classes_forced || (light_on && beyond && tt.nlight >= 256u))) class_nlight = tt.nlight;
            // (sparse slots without a launch of their own are light ones — tile_is_light holds for them — when the light class runs)
            if (class_nlight && !class_ntiny) class_nlight += tt.ntiny;
        }
        if (class_ntiny || class_nlight) slot_order.ensure(nlaunch, stream, false, 1.5f);
        if (class_nlight) ++counters.light_class_passes;
        if (class_ntiny) ++counters.sparse_class_passes;
        {
            const uint32_t keep_t = class_ntiny, keep_l = class_nlight;
            class_ntiny = class_nlight = 0u;  // (the table builder itself runs over every slot in one launch)
            c = make_ctx();
            launch_tile_halo_fill(c, halo_src.p, bhalo_src.p, slot_info.p, stream, (keep_t || keep_l) ? slot_order.p : nullptr, keep_l, keep_t);
            class_ntiny = keep_t; class_nlight = keep_l;
        }
        c = make_ctx();

        // ---- neighbour lists   (compute_contacts, contacts.rs:154-252): one pass into fixed-capacity ELL rows; if a list
        // turns out longer than the capacity the pass is repeated with room to spare (rare: the capacity follows the
        // longest list seen so far).  Speculative passes check at the end of the step instead.
        for (int nattempt = 0;; ++nattempt) {
            const bool r1 = nbr_ff.ensure((size_t)nslices * cap_ff * WAVE + 1, stream, false, 1.1f);
            const bool r2 = nbr_fb.ensure(nb ? (size_t)nslices * cap_fb * WAVE + 1 : 1, stream, false, 1.1f);
            slice_near.ensure((size_t)nslices + 1, stream, false, 1.1f);
            (void)r1; (void)r2;
            c = make_ctx();
            // (a pass whose list statistics are only looked at with the end-of-step publication lets that publication fold them)
            fold_stats = (spec || defer_lists) && n > 0;
            launch_nbr_build(c, lds, tile_list_stats.p, fold_stats ? nullptr : reinterpret_cast<unsigned long long*>(&d_rb.p->ncontacts_ff),
                             &d_rb.p->max_cnt_ff, comm ? reinterpret_cast<unsigned long long*>(&d_rb.p->ncontacts_own_ff) : nullptr, stream);
            if (spec || defer_lists) break;
            static_assert(offsetof(Readback, max_cnt_ff) == offsetof(Readback, ncontacts_ff) + 2 * sizeof(uint64_t), "list statistics travel in one copy");
            publish_and_wait(nullptr, true, false);
            const uint32_t need_ff = (h_rb->max_cnt_ff + 1) / 2, need_fb = (h_rb->max_cnt_fb + 1) / 2;
            if (need_ff <= cap_ff && need_fb <= cap_fb) break;
            if (nattempt >= 2) throw HipError(SALVA_HIP_E_HIP, "internal error: neighbour list capacity did not converge");
            if (need_ff > cap_ff) cap_ff = (need_ff + need_ff / 4 + 4u) & ~3u;
            if (need_fb > cap_fb) cap_fb = (need_fb + need_fb / 4 + 4u) & ~3u;
        }
#ifdef SALVA_HIP_DIAG
        // kernel-development builds: bank-conflict-aware list order (diag/sched.hip), SALVA_HIP_SCHED=1
        if (sched_mode > 0) launch_list_schedule(c, lds, stream);
#endif
    }
    if (timers) SALVA_HIP_CHECK(hipEventRecord(ev[1], stream));

    // ---- solver   (evaluate_kernels + compute_densities + solver.step, liquid_world.rs:123-144)
    // (DFSPH: the first evaluate of the divergence solve rides in the density pass when the plane layout applies, dfsph.hip)
    fused_first_divergence = prm.solver == SALVA_HIP_SOLVER_DFSPH && !no_fused_div && launch_density_alpha_div(c, lds, stream);
    // (single-domain IISPH: d_ii rides in the density pass, dfsph.hip k_density_alpha<true>)
    // (not with CFL sub-stepping: d_ii carries dt^2, and the substep is only chosen inside the solver)
    iisph_dii_fused = prm.solver == SALVA_HIP_SOLVER_IISPH && !comm && !no_fused_div && !cfl_mode;
    if (!fused_first_divergence) launch_density_alpha(c, lds, iisph_dii_fused ? dt : 0.0f, stream);
    if (comm) {
        refresh_f32(rho.p);
        // (the density pass wrote posmr.w = m / rho from each rank's OWN sum; a ghost's rho has just been replaced by its owner's,
        // so its volume follows — XSPH takes the neighbour's volume from there.  ADVICE r03)
        if (nghost_lo + nghost_hi) k_ghost_posmr<<<nblk(n), BLOCK, 0, stream>>>(n, gtag[cur].p, posm[cur].p, rho.p, posmr.p);
    }
    if (timers) SALVA_HIP_CHECK(hipEventRecord(evc[2], stream));
    if (prm.solver == SALVA_HIP_SOLVER_DFSPH) dfsph_solve(c, dt, g, st);
    else iisph_solve(c, dt, g, st);

    // ---- end of step: next bbox + flags (+ in a speculative pass: the true table totals and list statistics)
    static_assert(offsetof(Readback, bbox) == offsetof(Readback, flags) + sizeof(uint32_t), "flags and bbox travel in one copy");
    if (timers) SALVA_HIP_CHECK(hipEventRecord(ev[2], stream));
    {
        // The next step's grid part rides behind this step's publication when the cell box has been standing still (world.h PreGrid):
        // the device compares the box this step's position update found with the one this step ran on, and opens the gate if equal.
        bool has_custom_f = false;
        for (auto& f : fluids) for (auto& d : f.forces) has_custom_f |= d.kind == SALVA_HIP_FORCE_CUSTOM;
        const bool counting_now = ncf + 1 < 0x7fffffffull && (sort_mode == 0 || (sort_mode < 0 && ncf <= 16ull * n + (1ull << 20)));
        const bool stable = bbox_used_valid && memcmp(bbox_used_last, bbox_pre, sizeof(bbox_pre)) == 0;
        const bool want_pre = !pre_off && attempt == 0 && !spec && !comm && !timers && !has_dyn && !has_custom_f && !cfl_mode && counting_now && stable &&
                              n > 0 && nslots_bound > 0;
        memcpy(bbox_used_last, bbox_pre, sizeof(bbox_pre)); bbox_used_valid = true;
        PrePub pp{want_pre ? 1 : 0, chain_pending ? 1 : 0, {bbox_pre[0], bbox_pre[1], bbox_pre[2], bbox_pre[3], bbox_pre[4], bbox_pre[5]}};
        const uint32_t seq_end = publish_enqueue(spec ? G().tile_off.p + nslots_bound : nullptr, spec || defer_lists, true, &pp);
        if (want_pre) pre_enqueue_grid(nslots_bound);
        publish_wait(seq_end, spec, spec || defer_lists, true);
    }
    flags_clean = true;  // (k_publish_readback cleared them behind the copy)
    if (comm && prm.enable_timers) dist_time_fold();  // (the stream has drained up to the publication: every pair has completed)
    if (defer_lists && !spec) {
        const uint32_t need_ff = (h_rb->max_cnt_ff + 1) / 2, need_fb = (h_rb->max_cnt_fb + 1) / 2;
        if (need_ff > cap_ff || need_fb > cap_fb) {
            // a list was cut at the capacity: everything this pass computed is discarded; the pre-sort buffers are intact
            ++counters.discarded_passes;
            cur = cur0; dt_prev = dt_prev0; inv_dt_prev = inv_dt_prev0;
            memcpy(h_rb->bbox, bbox_pre, sizeof(bbox_pre));
            memcpy(last_iters, last_iters0, sizeof(last_iters0)); memcpy(prev_iters, prev_iters0, sizeof(prev_iters0));
            if (need_ff > cap_ff) cap_ff = (need_ff + need_ff / 4 + 4u) & ~3u;
            if (need_fb > cap_fb) cap_fb = (need_fb + need_fb / 4 + 4u) & ~3u;
            if (pre.valid) { pre.valid = false; pre_drop(); }
            continue;
        }
    }
    if (spec) {
        const TileAcc& a = h_rb->tile_total;
        const uint32_t need_ff = (h_rb->max_cnt_ff + 1) / 2, need_fb = (h_rb->max_cnt_fb + 1) / 2;
        const bool ok = a.nonempty <= tt.nonempty && a.max_s <= tt.max_s && a.max_sb <= tt.max_sb && a.nsl <= tt.nsl &&
                        (halo_stride || (a.s <= tt.s && a.sb <= tt.sb)) && need_ff <= cap_ff && need_fb <= cap_fb;
        if (!ok) {
            // the prediction did not hold: everything this pass computed is discarded; the pre-sort buffers are intact
            ++spec_misses; ++counters.speculative_passes; ++counters.discarded_passes;
            cur = cur0; dt_prev = dt_prev0; inv_dt_prev = inv_dt_prev0;
            memcpy(h_rb->bbox, bbox_pre, sizeof(bbox_pre));
            memcpy(last_iters, last_iters0, sizeof(last_iters0)); memcpy(prev_iters, prev_iters0, sizeof(prev_iters0));
            if (need_ff > cap_ff) cap_ff = (need_ff + need_ff / 4 + 4u) & ~3u;
            if (need_fb > cap_fb) cap_fb = (need_fb + need_fb / 4 + 4u) & ~3u;
            if (pre.valid) { pre.valid = false; pre_drop(); }
            continue;
        }
    }
    if (spec) ++counters.speculative_passes;
    if (chain_pending) {
        // A chained step (World::dfsph_solve): the publication says whether both solves converged within the batches they were given.
        chain_pending = false;
        auto adopt = [&](int which) {  // the outcome of a solve nobody waited for
            prev_iters[which] = last_iters[which];
            last_iters[which] = h_rb->solve[which][1];
            h_ctl[which].done = h_rb->solve[which][0]; h_ctl[which].iters = h_rb->solve[which][1];
            memcpy(&h_ctl[which].err, &h_rb->solve[which][2], sizeof(float));
        };
        uint32_t stage = h_rb->chain_ok ? 0u : h_rb->chain_stage;
        if (stage == 0u) {
            if (chain_div_pending) adopt(0);
            adopt(1);
            ++chain_steps; ++counters.chained_passes;
        } else {
            // one of them fell short: every kernel behind it returned at once.  Continue that solve the classic way (batches with a
            // wait each) and run what follows it; the flags raised so far were published (and cleared) by the first publication.
            if (stage != 1u && stage != 2u) throw HipError(SALVA_HIP_E_HIP, "internal error: a chained step broke at an unknown stage");
            ++chain_breaks; ++counters.chain_breaks;
            pre.valid = false;  // (its gate stayed shut: Readback::pre_ok needs the chain)
            const uint32_t flags0 = h_rb->flags;
            if (stage == 2u && chain_div_pending) adopt(0);
            dfsph_solve(c, dt, g, st, (int)stage);
            if (timers) SALVA_HIP_CHECK(hipEventRecord(ev[2], stream));  // (the solver's part of the step ends here, not at the first publication)
            publish_and_wait(nullptr, false, true);
            h_rb->flags |= flags0;
        }
        if (stage != 1u) { st.n_divergence_iters = (int32_t)h_ctl[0].iters; st.divergence_error = h_ctl[0].err; }
        if (stage == 0u) { st.n_pressure_iters = (int32_t)h_ctl[1].iters; st.density_error = h_ctl[1].err; }
    }
    pred_tt = h_rb->tile_total; pred_n = n; pred_valid = true;
    lists_checked = true;  // (every path to here has compared the longest lists with the capacity)
    break;
    }  // attempts
    acc_user = false;
    // decomposed runs: the contacts whose first particle this rank owns (fluid) / whose first particle lies in its slab
    // (boundary-boundary, k_boundary_volumes) — the ranks' counts add up to the undivided domain's counters.cd.ncontacts
    st.ncontacts = comm ? h_rb->ncontacts_own_ff + (nb ? h_rb->ncontacts_own_fb : 0) + ncontacts_bb
                        : h_rb->ncontacts_ff + (nb ? h_rb->ncontacts_fb : 0) + ncontacts_bb;
    bbox_known = true;
    last_ctx = c; last_ctx.ctl = nullptr; last_dt = dt; have_last_ctx = true;  // (dt: the substep the solver advanced by)
    if (timers) {
        // ev[2] was recorded before the end-of-step publication, whose arrival the host has seen through host-mapped memory — which
        // says nothing about the EVENT's completion signal: without the synchronisation hipEventElapsedTime can still answer
        // hipErrorNotReady (it did, on a fresh box: VERDICT r04).  Every other event of the step precedes ev[2] on the stream.
        // An interval that cannot be read is reported as NaN ("untimed"), never as 0.
        const double untimed = std::numeric_limits<double>::quiet_NaN();
        bool ok = hipEventSynchronize(ev[2]) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        auto ms = [&](hipEvent_t from, hipEvent_t to) -> double {
            float t = 0.0f;
            if (!ok) return untimed;
            if (hipEventElapsedTime(&t, from, to) != hipSuccess) { (void)hipGetLastError(); return untimed; }
            return (double)t;
        };
        const double a = ms(ev[0], ev[1]), b = ms(ev[1], ev[2]);
        // the reference's tree (liquid_world.rs:73-156); every interval is taken on the world's stream.  The timers are resumed and
        // paused in every substep (:88-147): they add up over the substeps of a step.
        st.grid_ms += (float)a; st.solver_ms += (float)b; st.step_ms += (float)(a + b);
        counters.step_time += a + b;
        counters.stages.collision_detection_time += a;
        counters.stages.solver_time += b;
        counters.cd.boundary_update_time += dcs_ms;  // DynamicContactSampling runs inside the step (liquid_world.rs:94-103)
        const double gi = ms(ev[0], evc[1]);
        counters.cd.grid_insertion_time += std::isnan(gi) ? gi : std::max(0.0, gi - dcs_ms);
        counters.cd.neighborhood_search_time += ms(evc[1], ev[1]);
        counters.solver.pressure_resolution_time += ms(evc[2], ev[2]);
        if (prm.solver == SALVA_HIP_SOLVER_DFSPH) counters.custom += ms(evc[3], evc[4]);
    }
    counters.cd.ncontacts = st.ncontacts;
    counters.n_divergence_iters = st.n_divergence_iters; counters.n_pressure_iters = st.n_pressure_iters;
    st.reserved[0] = (float)lds.max_halo_fluid; st.reserved[1] = (float)lds.max_halo_boundary; st.reserved[2] = (float)lds.threads;
    st.reserved[3] = (float)((double)(h_rb->ncontacts_ff + (nb ? h_rb->ncontacts_fb : 0)) / (double)std::max<uint32_t>(n, 1u));  // list entries per local particle
    st.reserved[4] = (float)(n - owned_count());  // ghosts
    if (h_rb->flags & 1u) {
        bbox_known = false;
        throw HipError(SALVA_HIP_E_NUMERIC, "zero density / boundary denominator or NaN detected (the reference would panic)");
    }
    if (h_rb->flags & 2u) {
        bbox_known = false;
        throw HipError(SALVA_HIP_E_HIP, "internal error: particle outside the cell table");
    }
    if (h_rb->flags & 8u) {
        bbox_known = false;
        throw HipError(SALVA_HIP_E_HIP, "internal error: a third particle mass in a world the host took for a two-mass world");
    }
    if (h_rb->flags & 4u)
        // k_dist_flags: such a particle was handed to the adjacent rank, which does not own its cells either — it would
        // never be mirrored as a ghost and its contacts across the next face would be lost
        throw HipError(SALVA_HIP_E_INVALID, "a particle moved across more than one slab in a single step (slabs too thin for this time step): "
                                            "results of this step are not reliable");
    return SALVA_HIP_OK;
}

// LiquidWorld::particles_intersecting_aabb (liquid_world.rs:210-243): particles whose distance to the box is below the
// particle radius.  The reference walks the cells of its (last step's) grid; here every particle's current position is
// tested, which finds the same particles plus those that entered the box's cells since the grid was built.
// Where a query finds its fluid particles.  Single domain: the staging arrays (host order; index = position in the concatenation of
// the fluids).  Decomposed run: the rank's working set after its last step — ghosts are skipped (their owner reports them), the
// index is the particle's GLOBAL id and the fluid slot rides in bits 8.. of the kind word.
struct QuerySrc {
    const float4* pos; uint32_t n;
    const uint32_t* gtag;   // nullptr: no ghosts in `pos`
    const uint32_t* gid;    // nullptr: index = i
    const uint32_t* model;  // nullptr: the slot follows from the index (collect_query)
};
__device__ __forceinline__ bool query_skip(const QuerySrc& q, uint32_t i) { return q.gtag && (q.gtag[i] & 0x80000000u); }
__device__ __forceinline__ void query_emit(const QuerySrc& q, uint32_t i, uint32_t kind, unsigned int* counter, uint32_t cap, uint32_t* out_kind,
                                           uint32_t* out_index) {
    const uint32_t k = atomicAdd(counter, 1u);
    if (k < cap) { out_kind[k] = kind | (q.model ? q.model[i] << 8 : 0u); out_index[k] = q.gid ? q.gid[i] : i; }
}
__global__ __launch_bounds__(BLOCK) void k_aabb_query(QuerySrc q, float3 lo, float3 hi, float r2,
                                                      uint32_t kind, unsigned int* __restrict__ counter, uint32_t cap,
                                                      uint32_t* __restrict__ out_kind, uint32_t* __restrict__ out_index) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= q.n || query_skip(q, i)) return;
    const float4 p = q.pos[i];
    const float dx = fmaxf(fmaxf(lo.x - p.x, p.x - hi.x), 0.0f), dy = fmaxf(fmaxf(lo.y - p.y, p.y - hi.y), 0.0f),
                dz = fmaxf(fmaxf(lo.z - p.z, p.z - hi.z), 0.0f);
    if (!(dx * dx + dy * dy + dz * dz < r2)) return;
    query_emit(q, i, kind, counter, cap, out_kind, out_index);
}
// Single domain: the staging arrays, refreshed from the working set.  Decomposed run (after its first step): the owned particles of
// the working set with their global ids — a query is a per-rank operation there, each rank reports what it owns; the union over
// the ranks is the undivided world's answer (boundary particles near a slab face are held, and reported, by both neighbours).
QuerySrc World::query_fluid_source() {
    if (comm && dist_started && sorted_valid) return QuerySrc{posm[cur].p, n, gtag[cur].p, perm[cur].p, model[cur].p};
    if (comm) throw HipError(SALVA_HIP_E_INVALID, "a query in a decomposed run needs a completed step (global ids exist from then on)");
    ensure_staging_current();
    return QuerySrc{st_pos.p, n, nullptr, nullptr, nullptr};
}

// (kind, global index) pairs collected on the device -> sorted (kind, slot, index-in-slot) triples on the host
uint64_t World::collect_query(unsigned int* d_count, uint32_t* d_kind, uint32_t* d_index, uint32_t cap, uint32_t* kinds, uint32_t* slots,
                              uint32_t* indices) {
    unsigned int total = 0;
    SALVA_HIP_CHECK(hipMemcpyAsync(&total, d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    const uint32_t m = std::min<uint32_t>(total, cap);
    if (m == 0 || !kinds || !slots || !indices) return total;
    std::vector<uint32_t> hk(m), hi_(m);
    SALVA_HIP_CHECK(hipMemcpy(hk.data(), d_kind, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SALVA_HIP_CHECK(hipMemcpy(hi_.data(), d_index, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> keys(m);
    for (uint32_t k = 0; k < m; ++k) keys[k] = ((uint64_t)hk[k] << 32) | hi_[k];
    map_query_hits(keys, kinds, slots, indices);
    return total;
}
// (kind word << 32 | global index) keys -> sorted (kind, slot, index-in-slot) triples
void World::map_query_hits(std::vector<uint64_t>& keys, uint32_t* kinds, uint32_t* slots, uint32_t* indices) {
    const uint32_t m = (uint32_t)keys.size();
    std::sort(keys.begin(), keys.end());
    for (uint32_t k = 0; k < m; ++k) {
        uint32_t kind = (uint32_t)(keys[k] >> 32);
        uint64_t g = keys[k] & 0xffffffffull;
        uint32_t s = 0;
        if (comm && (kind & 0xffu) == 0) {  // decomposed run: the slot came with the kind word, the index is a global id
            kinds[k] = 0u; slots[k] = kind >> 8; indices[k] = (uint32_t)g;
            continue;
        }
        kind &= 0xffu;
        if (kind == 0) { while (s + 1 < fluids.size() && g >= fluids[s].n) { g -= fluids[s].n; ++s; } }
        else { while (s + 1 < bounds.size() && g >= bounds[s].n) { g -= bounds[s].n; ++s; } }
        kinds[k] = kind; slots[k] = s; indices[k] = (uint32_t)g;
    }
}

uint64_t World::particles_in_aabb(const float mins[3], const float maxs[3], uint64_t capacity, uint32_t* kinds, uint32_t* slots,
                                  uint32_t* indices) {
    use_device();
    const QuerySrc qf = query_fluid_source(), qb{bst_pos.p, nb, nullptr, nullptr, nullptr};
    const uint32_t cap = (uint32_t)std::min<uint64_t>(capacity, 0xfffffff0ull);
    DevBuf<unsigned int> cnt;
    DevBuf<uint32_t> dk, di;
    cnt.ensure(1); dk.ensure(std::max(cap, 1u)); di.ensure(std::max(cap, 1u));
    SALVA_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(unsigned int), stream));
    const float r = prm.particle_radius;
    const float3 lo = make_float3(mins[0], mins[1], mins[2]), hi = make_float3(maxs[0], maxs[1], maxs[2]);
    if (qf.n) k_aabb_query<<<nblk(qf.n), BLOCK, 0, stream>>>(qf, lo, hi, r * r, 0u, cnt.p, cap, dk.p, di.p);
    if (nb) k_aabb_query<<<nblk(nb), BLOCK, 0, stream>>>(qb, lo, hi, r * r, 1u, cnt.p, cap, dk.p, di.p);
    return collect_query(cnt.p, dk.p, di.p, cap, kinds, slots, indices);
}

// LiquidWorld::particles_intersecting_shape (liquid_world.rs:245-280) for a ball / cuboid posed by (t, q): the particle must lie
// in a grid cell the shape's world AABB touches (hgrid.cells_intersecting_aabb: cell range floor(mins / h) .. floor(maxs / h),
// hgrid.rs:122-133) and `shape.distance_to_point(pos, pt, solid = true) <= particle_radius`.  The point goes into the shape's
// frame with the inverse isometry (conjugate quaternion); ball: max(|p| - radius, 0); cuboid: |max(|p| - half_extents, 0)|.
struct ShapeQuery {
    float t[3], q[4];       // isometry
    int kind; float p[3];   // shape
    int clo[3], chi[3];     // cell range of the world AABB
    float h, r;             // cell width, particle radius
};
__global__ __launch_bounds__(BLOCK) void k_shape_query(QuerySrc q, ShapeQuery s, uint32_t kind,
                                                       unsigned int* __restrict__ counter, uint32_t cap, uint32_t* __restrict__ out_kind,
                                                       uint32_t* __restrict__ out_index) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= q.n || query_skip(q, i)) return;
    const float4 pt = q.pos[i];
    bool bad = false;
    const int cx = cell_coord(pt.x, s.h, bad), cy = cell_coord(pt.y, s.h, bad), cz = cell_coord(pt.z, s.h, bad);
    if (bad || cx < s.clo[0] || cx > s.chi[0] || cy < s.clo[1] || cy > s.chi[1] || cz < s.clo[2] || cz > s.chi[2]) return;
    // inverse isometry: q^-1 * (pt - t)
    const float vx = pt.x - s.t[0], vy = pt.y - s.t[1], vz = pt.z - s.t[2];
    const float qx = -s.q[0], qy = -s.q[1], qz = -s.q[2], qw = s.q[3];
    const float tx = (qy * vz - qz * vy) * 2.0f, ty = (qz * vx - qx * vz) * 2.0f, tz = (qx * vy - qy * vx) * 2.0f;
    const float lx = vx + qw * tx + (qy * tz - qz * ty), ly = vy + qw * ty + (qz * tx - qx * tz), lz = vz + qw * tz + (qx * ty - qy * tx);
    float d;
    if (s.kind == SALVA_HIP_SHAPE_BALL) {
        d = fmaxf(sqrtf(lx * lx + ly * ly + lz * lz) - s.p[0], 0.0f);
    } else if (s.kind == SALVA_HIP_SHAPE_CAPSULE) {  // distance to the segment (0, -hh..hh, 0), minus the radius
        const float ey = ly - fminf(fmaxf(ly, -s.p[0]), s.p[0]);
        d = fmaxf(sqrtf(lx * lx + ey * ey + lz * lz) - s.p[1], 0.0f);
    } else if (s.kind == SALVA_HIP_SHAPE_CYLINDER) {  // beyond the caps and beyond the side
        const float ey = fmaxf(fabsf(ly) - s.p[0], 0.0f), er = fmaxf(sqrtf(lx * lx + lz * lz) - s.p[1], 0.0f);
        d = sqrtf(ey * ey + er * er);
    } else {
        const float dx = fmaxf(fabsf(lx) - s.p[0], 0.0f), dy = fmaxf(fabsf(ly) - s.p[1], 0.0f), dz = fmaxf(fabsf(lz) - s.p[2], 0.0f);
        d = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    if (!(d <= s.r)) return;
    query_emit(q, i, kind, counter, cap, out_kind, out_index);
}
uint64_t World::particles_in_shape(const float t[3], const float q[4], const SalvaHipShape& shape, uint64_t capacity, uint32_t* kinds,
                                   uint32_t* slots, uint32_t* indices) {
    use_device();
    const int nparams = shape_param_count(shape.kind);
    // the pose must be a rigid motion and the shape non-degenerate (the reference takes an Isometry and a parry shape, which
    // cannot be anything else): NaN / huge values would otherwise reach an undefined float -> int conversion below
    for (int a = 0; a < 3; ++a)
        if (!std::isfinite(t[a])) throw HipError(SALVA_HIP_E_INVALID, "shape query: non-finite translation");
    {
        float qn = 0.0f;
        for (int a = 0; a < 4; ++a) {
            if (!std::isfinite(q[a])) throw HipError(SALVA_HIP_E_INVALID, "shape query: non-finite rotation");
            qn += q[a] * q[a];
        }
        if (fabsf(qn - 1.0f) > 1.0e-3f) throw HipError(SALVA_HIP_E_INVALID, "shape query: the rotation must be a unit quaternion (x, y, z, w)");
    }
    for (int a = 0; a < nparams; ++a)
        if (!(shape.params[a] > 0.0f) || !std::isfinite(shape.params[a]))
            throw HipError(SALVA_HIP_E_INVALID, "shape query: radius / half extents must be positive and finite");
    const QuerySrc qf = query_fluid_source(), qb{bst_pos.p, nb, nullptr, nullptr, nullptr};
    ShapeQuery s{};
    for (int a = 0; a < 3; ++a) { s.t[a] = t[a]; s.p[a] = shape.params[a]; }
    for (int a = 0; a < 4; ++a) s.q[a] = q[a];
    s.kind = shape.kind; s.h = sc.h; s.r = prm.particle_radius;
    // world AABB (parry compute_aabb; dcs.hip)
    float ext[3];
    shape_world_extent(shape, q, ext);
    for (int a = 0; a < 3; ++a) {  // (clamped like the cell coordinates of the particles: tile.h cell_coord)
        s.clo[a] = (int)std::min(std::max(floorf((t[a] - ext[a]) / sc.h), -1073741824.0f), 1073741824.0f);
        s.chi[a] = (int)std::min(std::max(floorf((t[a] + ext[a]) / sc.h), -1073741824.0f), 1073741824.0f);
    }
    const uint32_t cap = (uint32_t)std::min<uint64_t>(capacity, 0xfffffff0ull);
    DevBuf<unsigned int> cnt;
    DevBuf<uint32_t> dk, di;
    cnt.ensure(1); dk.ensure(std::max(cap, 1u)); di.ensure(std::max(cap, 1u));
    SALVA_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(unsigned int), stream));
    if (qf.n) k_shape_query<<<nblk(qf.n), BLOCK, 0, stream>>>(qf, s, 0u, cnt.p, cap, dk.p, di.p);
    if (nb) k_shape_query<<<nblk(nb), BLOCK, 0, stream>>>(qb, s, 1u, cnt.p, cap, dk.p, di.p);
    return collect_query(cnt.p, dk.p, di.p, cap, kinds, slots, indices);
}

// LiquidWorld::particles_intersecting_shape for a shape whose geometry stays with the host (any parry `Shape`: the reference's
// query is generic, liquid_world.rs:247-280): the two parry calls — `shape.compute_aabb(pos)` and `shape.distance_to_point(pos,
// &pt, true)` — are callbacks; the device keeps the cell filter (hgrid.cells_intersecting_aabb, hgrid.rs:122-133) and hands the
// particles of those cells to the host in one copy.
__global__ __launch_bounds__(BLOCK) void k_cell_range_query(QuerySrc q, int clo0, int clo1, int clo2, int chi0, int chi1, int chi2, float h,
                                                            uint32_t kind, unsigned int* __restrict__ counter, uint32_t cap,
                                                            uint32_t* __restrict__ out_kind, uint32_t* __restrict__ out_index,
                                                            float4* __restrict__ out_pos) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= q.n || query_skip(q, i)) return;
    const float4 pt = q.pos[i];
    bool bad = false;
    const int cx = cell_coord(pt.x, h, bad), cy = cell_coord(pt.y, h, bad), cz = cell_coord(pt.z, h, bad);
    if (bad || cx < clo0 || cx > chi0 || cy < clo1 || cy > chi1 || cz < clo2 || cz > chi2) return;
    const uint32_t k = atomicAdd(counter, 1u);
    if (k < cap) { out_kind[k] = kind | (q.model ? q.model[i] << 8 : 0u); out_index[k] = q.gid ? q.gid[i] : i; out_pos[k] = pt; }
}
uint64_t World::particles_in_host_shape(const SalvaHipHostQueryShape& shape, uint64_t capacity, uint32_t* kinds, uint32_t* slots,
                                        uint32_t* indices) {
    use_device();
    if (!shape.aabb || !shape.distance) throw HipError(SALVA_HIP_E_INVALID, "a host query shape needs both callbacks");
    float mins[3], maxs[3];
    shape.aabb(shape.user, mins, maxs);
    int clo[3], chi[3];
    for (int a = 0; a < 3; ++a) {
        if (!(mins[a] <= maxs[a]) || !std::isfinite(mins[a]) || !std::isfinite(maxs[a]))
            throw HipError(SALVA_HIP_E_INVALID, "host query shape: the aabb callback returned an empty, infinite or NaN box");
        clo[a] = (int)std::min(std::max(floorf(mins[a] / sc.h), -1073741824.0f), 1073741824.0f);
        chi[a] = (int)std::min(std::max(floorf(maxs[a] / sc.h), -1073741824.0f), 1073741824.0f);
    }
    const QuerySrc qf = query_fluid_source(), qb{bst_pos.p, nb, nullptr, nullptr, nullptr};
    // Candidates go into scratch buffers the world keeps (three fresh (n + nb)-sized buffers per call were 190 MB at
    // 8 x 10^6 particles, allocated and freed synchronously): sized for what queries have needed so far; the kernel counts every
    // candidate and stores those that fit, so a box that holds more than the buffers do costs one repetition with room.
    const uint32_t all = qf.n + nb;
    unsigned int m = 0;
    for (int attempt = 0;; ++attempt) {
        const uint32_t ccap = std::min<uint32_t>(std::max<uint32_t>(hq_need, 65536u), std::max(all, 1u));
        hq_cnt.ensure(1); hq_kind.ensure(ccap, stream, false, 1.25f); hq_index.ensure(ccap, stream, false, 1.25f); hq_pos.ensure(ccap, stream, false, 1.25f);
        SALVA_HIP_CHECK(hipMemsetAsync(hq_cnt.p, 0, sizeof(unsigned int), stream));
        if (qf.n) k_cell_range_query<<<nblk(qf.n), BLOCK, 0, stream>>>(qf, clo[0], clo[1], clo[2], chi[0], chi[1], chi[2], sc.h, 0u, hq_cnt.p, ccap, hq_kind.p, hq_index.p, hq_pos.p);
        if (nb) k_cell_range_query<<<nblk(nb), BLOCK, 0, stream>>>(qb, clo[0], clo[1], clo[2], chi[0], chi[1], chi[2], sc.h, 1u, hq_cnt.p, ccap, hq_kind.p, hq_index.p, hq_pos.p);
        SALVA_HIP_CHECK(hipGetLastError());
        SALVA_HIP_CHECK(hipMemcpyAsync(&m, hq_cnt.p, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        if (m <= ccap) break;
        if (attempt >= 1) throw HipError(SALVA_HIP_E_HIP, "internal error: the candidate count of a host shape query grew between two passes");
        hq_need = m + m / 4u;
    }
    DevBuf<uint32_t>& dk = hq_kind; DevBuf<uint32_t>& di = hq_index; DevBuf<float4>& dp = hq_pos;
    if (m == 0) return 0;
    std::vector<uint32_t> hk(m), hi_(m);
    std::vector<float4> hp(m);
    SALVA_HIP_CHECK(hipMemcpy(hk.data(), dk.p, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SALVA_HIP_CHECK(hipMemcpy(hi_.data(), di.p, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SALVA_HIP_CHECK(hipMemcpy(hp.data(), dp.p, m * sizeof(float4), hipMemcpyDeviceToHost));
    std::vector<float> pts(3 * (size_t)m), dist((size_t)m, std::numeric_limits<float>::infinity());
    for (unsigned int k = 0; k < m; ++k) { pts[3 * k] = hp[k].x; pts[3 * k + 1] = hp[k].y; pts[3 * k + 2] = hp[k].z; }
    shape.distance(shape.user, m, pts.data(), dist.data());
    std::vector<uint64_t> keys;
    keys.reserve(m);
    for (unsigned int k = 0; k < m; ++k)
        if (dist[k] <= prm.particle_radius) keys.push_back(((uint64_t)hk[k] << 32) | hi_[k]);  // liquid_world.rs:263, :272 (NaN: not a hit)
    const uint64_t total = keys.size();
    if (total == 0 || !kinds || !slots || !indices) return total;
    if (keys.size() > capacity) {  // the first `capacity` in the sorted order, like the device arms
        std::sort(keys.begin(), keys.end());
        keys.resize((size_t)capacity);
    }
    map_query_hits(keys, kinds, slots, indices);
    return total;
}

// The contact lists of the last step (they describe the positions the step started from, as the reference's
// ContactManager does after `step`).  offsets: n+1 entries; entries are written only if `capacity` holds them all.
uint64_t World::get_fluid_contacts(uint32_t slot, int boundary, uint64_t* offsets, uint32_t* j_model, uint32_t* j, uint64_t capacity) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    if (comm) throw HipError(SALVA_HIP_E_INVALID, "host-order contact export does not exist in a multi-GPU run: use salva_hip_get_local_contacts");
    if (!have_last_ctx || !sorted_valid) throw HipError(SALVA_HIP_E_INVALID, "no completed step: there are no contact lists yet");
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    std::vector<uint32_t> cnt(nn, 0);
    DevBuf<uint32_t> d_cnt;
    d_cnt.ensure(std::max<size_t>(n, 1));
    if (!(boundary && nb == 0) && nn) {
        launch_unsort_u32(n, perm[cur].p, boundary ? nfb.p : nff.p, d_cnt.p, stream);
        SALVA_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt.p + off, nn * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    std::vector<uint64_t> offs(nn + 1, 0);
    for (uint64_t k = 0; k < nn; ++k) offs[k + 1] = offs[k] + cnt[k];
    const uint64_t total = offs[nn];
    if (offsets) memcpy(offsets, offs.data(), (nn + 1) * sizeof(uint64_t));
    if (!j_model || !j || capacity < total || total == 0) return total;
    std::vector<uint32_t> moff(std::max<size_t>(fluids.size(), 1), 0), boff(std::max<size_t>(bounds.size(), 1), 0);
    for (uint32_t s = 0; s < fluids.size(); ++s) moff[s] = (uint32_t)fluid_offset(s);
    for (uint32_t s = 0; s < bounds.size(); ++s) boff[s] = (uint32_t)boundary_offset(s);
    DevBuf<uint64_t> d_offs;
    DevBuf<uint32_t> d_moff, d_boff, d_jm, d_j;
    d_offs.ensure(nn + 1); d_moff.ensure(moff.size()); d_boff.ensure(boff.size()); d_jm.ensure(total); d_j.ensure(total);
    SALVA_HIP_CHECK(hipMemcpyAsync(d_offs.p, offs.data(), (nn + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(d_moff.p, moff.data(), moff.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(d_boff.p, boff.data(), boff.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    flags_clean = false;
    launch_export_contacts(last_ctx, G().keys[1].p, slot, boundary, d_offs.p, d_moff.p, d_boff.p, d_jm.p, d_j.p, stream);
    SALVA_HIP_CHECK(hipMemcpyAsync(j_model, d_jm.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(j, d_j.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    return total;
}

// ---- the working set as it is (include/salva_hip.h "local view"): sorted order, global ids; per rank in a decomposed run
__global__ void k_local_ghost_flags(uint32_t n, const uint32_t* __restrict__ gtag, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (gtag && (gtag[i] & 0x80000000u)) ? 1 : 0;
}
void World::get_local(uint32_t* ids, uint32_t* fluid_slots, uint8_t* is_ghost, float* positions, float* velocities, float* densities, float* volumes) {
    use_device();
    if (!sorted_valid || (!have_last_ctx && !in_force_cb)) throw HipError(SALVA_HIP_E_INVALID, "no completed step: the working set has no order yet");
    if (n == 0) return;
    if (ids) SALVA_HIP_CHECK(hipMemcpyAsync(ids, perm[cur].p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (fluid_slots) SALVA_HIP_CHECK(hipMemcpyAsync(fluid_slots, model[cur].p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (is_ghost) {
        DevBuf<uint8_t> flags;
        flags.ensure(n);
        k_local_ghost_flags<<<nblk(n), BLOCK, 0, stream>>>(n, comm ? gtag[cur].p : nullptr, flags.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(is_ghost, flags.p, (size_t)n, hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    scratch_f.ensure(3 * (size_t)n, stream, false, 1.1f);
    for (int k = 0; k < 2; ++k) {
        float* out = k == 0 ? positions : velocities;
        if (!out) continue;
        // inside a force callback fluid.velocities equal w = v + dv (dfsph_solver.rs:688-693), as in force_get_state
        const float4* src = k == 0 ? posm[cur].p : (in_force_cb ? last_ctx.w : vel[cur].p);
        k_unpack_xyz<<<nblk(n), BLOCK, 0, stream>>>(n, src, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(out, scratch_f.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (densities) SALVA_HIP_CHECK(hipMemcpyAsync(densities, rho.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (volumes) {  // (they ride in vel.w, device_types.h)
        k_unpack_w<<<nblk(n), BLOCK, 0, stream>>>(n, vel[cur].p, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(volumes, scratch_f.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream));
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
}
uint64_t World::get_local_contacts(int boundary, uint64_t* offsets, uint32_t* j_model, uint32_t* j, uint64_t capacity) {
    use_device();
    if (!have_last_ctx || !sorted_valid) throw HipError(SALVA_HIP_E_INVALID, "no completed step: there are no contact lists yet");
    std::vector<uint32_t> cnt(n, 0);
    if (!(boundary && nb == 0) && n) {
        SALVA_HIP_CHECK(hipMemcpyAsync(cnt.data(), boundary ? nfb.p : nff.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    std::vector<uint64_t> offs((size_t)n + 1, 0);
    for (uint64_t k = 0; k < n; ++k) offs[k + 1] = offs[k] + cnt[k];
    const uint64_t total = offs[n];
    if (offsets) memcpy(offsets, offs.data(), ((size_t)n + 1) * sizeof(uint64_t));
    if (!j_model || !j || capacity < total || total == 0) return total;
    std::vector<uint32_t> boff(std::max<size_t>(bounds.size(), 1), 0);
    for (uint32_t s = 0; s < bounds.size(); ++s) boff[s] = (uint32_t)boundary_offset(s);
    DevBuf<uint64_t> d_offs;
    DevBuf<uint32_t> d_boff, d_jm, d_j;
    d_offs.ensure((size_t)n + 1); d_boff.ensure(boff.size()); d_jm.ensure(total); d_j.ensure(total);
    SALVA_HIP_CHECK(hipMemcpyAsync(d_offs.p, offs.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(d_boff.p, boff.data(), boff.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    flags_clean = false;
    launch_export_contacts_local(last_ctx, G().keys[1].p, boundary, d_offs.p, d_boff.p, d_jm.p, d_j.p, stream);
    SALVA_HIP_CHECK(hipMemcpyAsync(j_model, d_jm.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipMemcpyAsync(j, d_j.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    return total;
}
__global__ void k_add_acc_local(uint32_t n, const float* __restrict__ in, float4* __restrict__ acc) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    float4 a = acc[s];
    a.x += in[3 * (size_t)s]; a.y += in[3 * (size_t)s + 1]; a.z += in[3 * (size_t)s + 2];
    acc[s] = a;
}
void World::force_add_local_accelerations(const float* acc_h) {
    use_device();
    if (!in_force_cb) throw HipError(SALVA_HIP_E_INVALID, "only available inside a force callback");
    if (!acc_h) throw HipError(SALVA_HIP_E_INVALID, "null accelerations");
    if (n == 0) return;
    scratch_f.ensure(3 * (size_t)n, stream, false, 1.1f);
    SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, acc_h, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream));
    // (what lands on a ghost is overwritten by the refresh of w that follows the forces, world_dist)
    k_add_acc_local<<<nblk(n), BLOCK, 0, stream>>>(n, scratch_f.p, last_ctx.acc);
    SALVA_HIP_CHECK(hipGetLastError());
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
}

void World::get_force_stats(uint32_t slot, uint32_t force, int32_t* iters, float* err) {
    if (slot >= fluids.size() || force >= fluids[slot].forces.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot / force index out of range");
    const FluidSlot& f = fluids[slot];
    if (iters) *iters = force < f.force_iters.size() ? (int32_t)f.force_iters[force] : 0;
    if (err) *err = force < f.force_errs.size() ? f.force_errs[force] : 0.0f;
}

// ------------------------------------------------------------------------------------------------ downloads
void World::get_fluid(uint32_t slot, float* pos, float* vel_out) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    if (nn == 0) return;
    ensure_staging_current();
    scratch_f.ensure(3 * nn, stream, false, 1.1f);
    if (pos) {
        k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_pos.p + off, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(pos, scratch_f.p, 3 * nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (vel_out) {
        k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_vel.p + off, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(vel_out, scratch_f.p, 3 * nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
}

// ---- asynchronous read-back.  The reference's users read fluid.positions / velocities after every step
// (integrations/rapier/testbed_plugin.rs:361-367); the synchronous salva_hip_get_fluid costs 0.6 ms per step at 10^6 particles
// (un-sort into the staging arrays, unpack, copy, stream synchronisation, per array, serial with the step: 1.91 against 1.31 ms per
// step over the bench protocol, tools/pcie_probe.py — round 3's "5.57 ms" compared different steps of the scene).  Here: two scatter kernels on the main stream write (x, y, z) in host order straight from the sorted
// working set (~10 us), the copy stream takes them out behind an event while the main stream already runs the next step, and
// the host collects them with salva_hip_wait_download.  A destination in pinned memory (salva_hip_host_alloc / _register) is
// written by the DMA engine directly; a pageable one goes through the library's pinned buffers and one memcpy in the wait.
__global__ void k_scatter_xyz(uint32_t n, const uint32_t* __restrict__ perm, uint32_t off, uint32_t nn, const float4* __restrict__ src,
                              float* __restrict__ dst) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n) return;
    const uint32_t h = perm[s] - off;
    if (h >= nn) return;  // (another fluid's particle)
    const float4 v = src[s];
    dst[3 * (size_t)h] = v.x; dst[3 * (size_t)h + 1] = v.y; dst[3 * (size_t)h + 2] = v.z;
}
static bool host_pointer_is_pinned(const void* p) {
    hipPointerAttribute_t a{};
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }  // (plain malloc'ed memory is unknown to the runtime)
    return a.type == hipMemoryTypeHost;
}
void World::get_fluid_async(uint32_t slot, float* pos, float* vel_out) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    if (comm && dist_started) throw HipError(SALVA_HIP_E_INVALID, "host-order fluid arrays are not maintained in a multi-GPU run: use salva_hip_get_owned");
    wait_download();  // one download in flight
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    if (nn == 0 || (!pos && !vel_out)) return;
    if (!dl_stream) {
        SALVA_HIP_CHECK(hipStreamCreateWithFlags(&dl_stream, hipStreamNonBlocking));
        SALVA_HIP_CHECK(hipEventCreateWithFlags(&ev_dl_ready, hipEventDisableTiming));
        SALVA_HIP_CHECK(hipEventCreateWithFlags(&ev_dl_done, hipEventDisableTiming));
    }
    const size_t bytes = 3 * nn * sizeof(float);
    float* dsts[2] = {pos, vel_out};
    const float4* sorted_src[2] = {posm[cur].p, vel[cur].p};
    const float4* staged_src[2] = {st_pos.p + off, st_vel.p + off};
    const bool from_sorted = sorted_valid && !staging_current;
    for (int a = 0; a < 2; ++a) {
        if (!dsts[a]) continue;
        dl_dev[a].ensure(3 * nn, stream, false, 1.1f);
        if (from_sorted) k_scatter_xyz<<<nblk(n), BLOCK, 0, stream>>>(n, perm[cur].p, (uint32_t)off, (uint32_t)nn, sorted_src[a], dl_dev[a].p);
        else k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, staged_src[a], dl_dev[a].p);
    }
    SALVA_HIP_CHECK(hipGetLastError());
    SALVA_HIP_CHECK(hipEventRecord(ev_dl_ready, stream));
    SALVA_HIP_CHECK(hipStreamWaitEvent(dl_stream, ev_dl_ready, 0));
    dl = PendingDownload{};
    dl.bytes = bytes;
    for (int a = 0; a < 2; ++a) {
        if (!dsts[a]) continue;
        dl.dst[a] = dsts[a];
        dl.staged[a] = !host_pointer_is_pinned(dsts[a]);
        float* target = dsts[a];
        if (dl.staged[a]) {
            if (h_dl_cap[a] < bytes) {
                if (h_dl[a]) SALVA_HIP_CHECK(hipHostFree(h_dl[a]));
                h_dl[a] = nullptr; h_dl_cap[a] = 0;
                const size_t cap = bytes + bytes / 8;
                SALVA_HIP_CHECK(hipHostMalloc((void**)&h_dl[a], cap, hipHostMallocDefault));
                h_dl_cap[a] = cap;
            }
            target = h_dl[a];
        }
        SALVA_HIP_CHECK(hipMemcpyAsync(target, dl_dev[a].p, bytes, hipMemcpyDeviceToHost, dl_stream));
    }
    SALVA_HIP_CHECK(hipEventRecord(ev_dl_done, dl_stream));
    dl.active = true;
}
void World::wait_download() {
    if (!dl.active) return;
    use_device();
    dl.active = false;
    SALVA_HIP_CHECK(hipEventSynchronize(ev_dl_done));
    for (int a = 0; a < 2; ++a)
        if (dl.dst[a] && dl.staged[a]) memcpy(dl.dst[a], h_dl[a], dl.bytes);
}

void World::get_fluid_field(uint32_t slot, int field, float* out) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    if (!out) throw HipError(SALVA_HIP_E_INVALID, "null output");
    if (comm && dist_started) throw HipError(SALVA_HIP_E_INVALID, "per-fluid host-order access is not available in a multi-GPU run: use salva_hip_get_owned");
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    if (nn == 0) return;
    const bool vec = field == SALVA_HIP_FIELD_VELOCITY_CHANGE || field == SALVA_HIP_FIELD_ACCELERATION;
    const size_t width = vec ? 3 : 1;
    scratch_f.ensure(std::max<size_t>(3 * nn, n), stream, false, 1.1f);
    auto finish = [&](const float* src) {
        SALVA_HIP_CHECK(hipMemcpyAsync(out, src, width * nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    };
    switch (field) {
        case SALVA_HIP_FIELD_VELOCITY_CHANGE:
            ensure_staging_current();
            k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_dv.p + off, scratch_f.p);
            finish(scratch_f.p);
            return;
        case SALVA_HIP_FIELD_PRESSURE:
            ensure_staging_current();
            k_unpack_w<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_dv.p + off, scratch_f.p);
            finish(scratch_f.p);
            return;
        case SALVA_HIP_FIELD_VOLUME:
            ensure_staging_current();
            k_unpack_w<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_pos.p + off, scratch_f.p);
            finish(scratch_f.p);
            return;
        case SALVA_HIP_FIELD_ACCELERATION:
            // accelerations are zero after every step (integrate_and_clear_accelerations) unless the host set them
            if (acc_user) {
                k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, st_acc.p + off, scratch_f.p);
                finish(scratch_f.p);
            } else {
                memset(out, 0, 3 * nn * sizeof(float));
            }
            return;
        default: break;
    }
    // per-step solver scratch lives in sorted order: valid only right after a step
    if (!sorted_valid || !have_last_ctx)
        throw HipError(SALVA_HIP_E_INVALID, "solver scratch fields are only available right after a step");
    switch (field) {
        case SALVA_HIP_FIELD_DENSITY: launch_unsort_f32(n, perm[cur].p, rho.p, scratch_f.p, stream); break;
        case SALVA_HIP_FIELD_ALPHA: launch_unsort_f32(n, perm[cur].p, alpha.p, scratch_f.p, stream); break;
        case SALVA_HIP_FIELD_NUM_FLUID_CONTACTS: launch_unsort_u32_as_f32(n, perm[cur].p, nff.p, scratch_f.p, stream); break;
        case SALVA_HIP_FIELD_NUM_BOUNDARY_CONTACTS:
            if (nb == 0) { memset(out, 0, nn * sizeof(float)); return; }
            launch_unsort_u32_as_f32(n, perm[cur].p, nfb.p, scratch_f.p, stream);
            break;
        default: throw HipError(SALVA_HIP_E_INVALID, "unknown field");
    }
    finish(scratch_f.p + off);
}

void World::get_boundary(uint32_t slot, float* volumes, float* forces) {
    use_device();
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    const uint64_t nn = bounds[slot].n, off = boundary_offset(slot);
    if (nn == 0) return;
    if (volumes) {
        upload_tables();
        build_boundary_grid();
        scratch_f4.ensure(nb);
        scratch_f.ensure(nb, stream, false, 1.1f);
        launch_unsort_f4(nb, bperm.p, bposv.p, scratch_f4.p, stream);
        k_unpack_w<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f4.p + off, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(volumes, scratch_f.p, nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (forces) {
        scratch_f.ensure(3 * nn, stream, false, 1.1f);
        k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, bforce.p + off, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(forces, scratch_f.p, 3 * nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
}

// ------------------------------------------------------------------------------------------------ rigid-body coupling
// integrations/rapier/fluids_pipeline.rs, StaticSampling arm.  q * v of nalgebra's UnitQuaternion (geometry/
// quaternion_ops.rs): t = 2 q.vec x v; v' = v + w t + q.vec x t.
__global__ void k_boundary_pose(uint32_t n, const float4* __restrict__ local, SalvaHipRigidPose p, float4* __restrict__ pos,
                                float4* __restrict__ vel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 pt = local[i];
    const float qx = p.rotation[0], qy = p.rotation[1], qz = p.rotation[2], qw = p.rotation[3];
    // every product is rounded on its own (no FMA contraction): these positions feed the exact d^2 <= h^2 contact test, and
    // sample points 2r apart make pairs that sit exactly on d = h (found by the literal basic3 scene: 250 boundary-boundary
    // contacts fewer than the CPU with contracted products)
    const float tx = (opaque(qy * pt.z) - opaque(qz * pt.y)) * 2.0f, ty = (opaque(qz * pt.x) - opaque(qx * pt.z)) * 2.0f,
                tz = (opaque(qx * pt.y) - opaque(qy * pt.x)) * 2.0f;
    const float cx = opaque(qy * tz) - opaque(qz * ty), cy = opaque(qz * tx) - opaque(qx * tz), cz = opaque(qx * ty) - opaque(qy * tx);
    float4 o = pos[i];  // .w (volume slot) untouched
    o.x = ((opaque(tx * qw) + cx) + pt.x) + p.translation[0];
    o.y = ((opaque(ty * qw) + cy) + pt.y) + p.translation[1];
    o.z = ((opaque(tz * qw) + cz) + pt.z) + p.translation[2];
    pos[i] = o;
    float4 v = vel[i];  // .w carries the boundary's model id
    if (p.has_body) {
        // body.velocity_at_point(pt) with the local point, as the reference writes it (:183)
        const float dx = pt.x - p.world_com[0], dy = pt.y - p.world_com[1], dz = pt.z - p.world_com[2];
        v.x = p.linvel[0] + (opaque(p.angvel[1] * dz) - opaque(p.angvel[2] * dy));
        v.y = p.linvel[1] + (opaque(p.angvel[2] * dx) - opaque(p.angvel[0] * dz));
        v.z = p.linvel[2] + (opaque(p.angvel[0] * dy) - opaque(p.angvel[1] * dx));
    } else {
        v.x = v.y = v.z = 0.0f;
    }
    vel[i] = v;
}

// per-block partial sums of f and (x - c) x f in f64 (6 doubles per block)
__global__ void k_boundary_wrench(uint32_t n, const float4* __restrict__ pos, const float4* __restrict__ force, float cx, float cy,
                                  float cz, double* __restrict__ partial) {
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pos[i], f = force[i];
        const float rx = p.x - cx, ry = p.y - cy, rz = p.z - cz;
        a[0] += f.x; a[1] += f.y; a[2] += f.z;
        a[3] += (double)(ry * f.z - rz * f.y); a[4] += (double)(rz * f.x - rx * f.z); a[5] += (double)(rx * f.y - ry * f.x);
    }
    __shared__ double sh[BLOCK / WAVE][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double v = a[k];
        for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
        if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = 0;
        for (uint32_t w = 0; w < blockDim.x / WAVE; ++w) v += sh[w][threadIdx.x];
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

void World::set_boundary_sampling(uint32_t slot, uint64_t nn, const float* local_points, uint32_t memberships, uint32_t filter) {
    const bool keep_forces = slot < bounds.size() ? bounds[slot].wants_forces : false;
    set_boundary(slot, nn, local_points, nullptr, memberships, filter, keep_forces);
    auto buf = std::make_shared<DevBuf<float4>>();
    buf->ensure(nn ? nn : 1);
    if (nn) {
        SALVA_HIP_CHECK(hipMemcpyAsync(buf->p, bst_pos.p + boundary_offset(slot), nn * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    bounds[slot].sampling = buf;
}

void World::update_boundary_pose(uint32_t slot, const SalvaHipRigidPose& pose) {
    use_device();
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    BoundarySlot& b = bounds[slot];
    if (!b.sampling && !b.dyn_kind)
        throw HipError(SALVA_HIP_E_INVALID, "boundary has no sampling method (salva_hip_set_boundary_sampling / _dynamic_sampling)");
    for (int k = 0; k < 4; ++k)
        if (!std::isfinite(pose.rotation[k])) throw HipError(SALVA_HIP_E_INVALID, "non-finite pose");
    if (b.dyn_kind) {  // the projection itself runs inside the step, where the reference runs it
        if (pose.has_body) {
            const bool wants = pose.is_dynamic != 0;
            if (wants != b.wants_forces) { b.wants_forces = wants; tables_dirty = true; }
        }
        b.dyn_pose = pose;
        return;
    }
    const uint64_t off = boundary_offset(slot);
    if (pose.has_body) {
        const bool wants = pose.is_dynamic != 0;
        if (wants != b.wants_forces) { b.wants_forces = wants; tables_dirty = true; }
    }
    if (b.n) {
        k_boundary_pose<<<nblk(b.n), BLOCK, 0, stream>>>((uint32_t)b.n, b.sampling->p, pose, bst_pos.p + off, bst_vel.p + off);
        SALVA_HIP_CHECK(hipGetLastError());
        SALVA_HIP_CHECK(hipMemsetAsync(bforce.p + off, 0, b.n * sizeof(float4), stream));  // boundary.clear_forces(true) :262
    }
    b_dirty = true; have_last_ctx = false;
}

// ------------------------------------------------------------------------------------------------ DynamicContactSampling
// ColliderCouplingSet::register_coupling(boundary, collider, ColliderSampling::DynamicContactSampling) (fluids_pipeline.rs:42-43,
// 96-114): the boundary starts empty; every step re-emits its particles from the fluid near the collider.
void World::set_boundary_dynamic_sampling(uint32_t slot, const SalvaHipShape& shape, uint32_t memberships, uint32_t filter) {
    const int np = shape_param_count(shape.kind);
    for (int a = 0; a < np; ++a)
        if (!(shape.params[a] > 0.0f) || !std::isfinite(shape.params[a])) throw HipError(SALVA_HIP_E_INVALID, "shape parameters must be positive");
    const bool keep_forces = slot < bounds.size() ? bounds[slot].wants_forces : false;
    set_boundary(slot, 0, nullptr, nullptr, memberships, filter, keep_forces);
    BoundarySlot& b = bounds[slot];
    b.sampling.reset();
    b.dyn_kind = shape.kind;
    b.dyn_shape = shape;
    b.dyn_pose = SalvaHipRigidPose{};
    b.dyn_pose.rotation[3] = 1.0f;
    b.dyn_src = std::make_shared<DevBuf<uint32_t>>();
}

void World::set_boundary_dynamic_sampling_host(uint32_t slot, const SalvaHipHostShape& shape, uint32_t memberships, uint32_t filter) {
    if (!shape.aabb || !shape.project) throw HipError(SALVA_HIP_E_INVALID, "a host shape needs both callbacks");
    const bool keep_forces = slot < bounds.size() ? bounds[slot].wants_forces : false;
    set_boundary(slot, 0, nullptr, nullptr, memberships, filter, keep_forces);
    BoundarySlot& b = bounds[slot];
    b.sampling.reset();
    b.dyn_kind = SALVA_HIP_SHAPE_HOST;
    b.dyn_shape = SalvaHipShape{};
    b.dyn_host = shape;
    b.dyn_pose = SalvaHipRigidPose{};
    b.dyn_pose.rotation[3] = 1.0f;
    b.dyn_src = std::make_shared<DevBuf<uint32_t>>();
}

// unregister_coupling (fluids_pipeline.rs:116-125): the boundary stays, with the particles it holds now, as a plain boundary;
// nothing of the sampling method — in particular no host callback or user pointer — is kept
void World::clear_boundary_sampling(uint32_t slot) {
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    BoundarySlot& b = bounds[slot];
    b.sampling.reset();
    b.dyn_kind = 0;
    b.dyn_shape = SalvaHipShape{};
    b.dyn_host = SalvaHipHostShape{};
    b.dyn_src.reset(); b.dyn_src_model.reset();
    // the particles keep the velocities k_boundary_pose / k_dcs_emit wrote for the moving collider (the reference keeps them after
    // unregister_coupling too): from now on make_ctx must not take the upload's "at rest" for them
    if (b.n) b.vel_zero = false;
}

// parameters of a built-in collider shape (include/salva_hip.h); throws for any other kind
int shape_param_count(int kind) {
    switch (kind) {
        case SALVA_HIP_SHAPE_BALL: return 1;
        case SALVA_HIP_SHAPE_CUBOID: return 3;
        case SALVA_HIP_SHAPE_CAPSULE: case SALVA_HIP_SHAPE_CYLINDER: return 2;
        default: throw HipError(SALVA_HIP_E_INVALID, "unknown shape kind (ball, cuboid, capsule (y) and cylinder (y) are built in; other parry shapes belong to the host)");
    }
}

bool World::has_dynamic_sampling() const {
    for (const BoundarySlot& b : bounds) if (b.dyn_kind) return true;
    return false;
}

uint64_t World::boundary_len(uint32_t slot) const {
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    return bounds[slot].n;
}

// Change the particle count of one boundary in place: the rows of the boundaries behind it move, its own rows are left for
// the caller to fill (forces zeroed).  No allocation once the buffers have grown.
void World::resize_boundary_slot(uint32_t slot, uint64_t nn) {
    BoundarySlot& b = bounds[slot];
    const uint64_t old_n = b.n;
    if (old_n == nn) return;
    const uint64_t off = boundary_offset(slot), old_total = nb, new_total = old_total - old_n + nn;
    if (new_total >= 0xfffffff0ull) throw HipError(SALVA_HIP_E_CAPACITY, "too many boundary particles");
    const uint64_t tail_src = off + old_n, tail_len = old_total - tail_src;
    DevBuf<float4>* bufs[3] = {&bst_pos, &bst_vel, &bforce};
    for (DevBuf<float4>* buf : bufs) {
        buf->ensure(std::max<uint64_t>(new_total, 1), stream, true, 1.5f);
        if (tail_len) {
            scratch_f4.ensure(tail_len, stream, false, 1.5f);
            SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f4.p, buf->p + tail_src, tail_len * sizeof(float4), hipMemcpyDeviceToDevice, stream));
            SALVA_HIP_CHECK(hipMemcpyAsync(buf->p + off + nn, scratch_f4.p, tail_len * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        }
    }
    b.n = nn;
    nb = (uint32_t)new_total;
    b_dirty = true; have_last_ctx = false;
}

// The DynamicContactSampling arm of ColliderCouplingManager::update_boundaries (fluids_pipeline.rs:193-259, :262) for every
// boundary registered with it, in slot order, at the reference's point of the substep: the fluid cell keys exist (grid
// insertion, liquid_world.rs:90-91), the boundaries are not in the grid yet (:106).  Works on the sorted working set of the
// previous step (posm[cur] / vel[cur], G().keys[0] = this step's keys in that order).
void World::run_dynamic_sampling() {
    TileGrid gv = gf.device(nullptr);
    for (uint32_t slot = 0; slot < bounds.size(); ++slot) {
        BoundarySlot& b = bounds[slot];
        if (!b.dyn_kind) continue;
        uint32_t cnt = 0;
        const float4* emit_src = nullptr;  // the compacted (projection, source particle) rows
        // (decomposed run: a failure of the rank-local part must not leave the other ranks waiting in the collective below —
        // it is parked and travels with the counts, dist_gather_emitted)
        HipError local_error(SALVA_HIP_E_INVALID, "");
        bool local_failed = false;
        try {
        if (n && b.dyn_kind == SALVA_HIP_SHAPE_HOST) {
            // the host's shape: box tests on the device, the projections on the host, the rest of the loop body on the device
            float mins[3], maxs[3];
            b.dyn_host.aabb(b.dyn_host.user, mins, maxs);
            for (int a = 0; a < 3; ++a)
                if (!(mins[a] <= maxs[a])) throw HipError(SALVA_HIP_E_INVALID, "host shape: the aabb callback returned an empty or NaN box");
            const DcsParams prm_d = dcs_params_host(mins, maxs, sc.h, prm.particle_radius, dt_prev);
            dcs_cand.ensure(n, stream, false, 1.1f); dcs_out.ensure(n, stream, false, 1.1f); dcs_flag.ensure(n, stream, false, 1.1f);
            dcs_num.ensure(1);
            launch_dcs_gather(n, posm[cur].p, vel[cur].p, G().keys[0].p, gv, prm_d, dcs_cand.p, dcs_flag.p, stream);
            const size_t tb = select_flagged_temp_bytes(n);
            ensure_cub_temp(tb);
            select_flagged_f4(cub_temp.p, tb, dcs_cand.p, dcs_flag.p, dcs_out.p, dcs_num.p, n, stream);
            SALVA_HIP_CHECK(hipMemcpyAsync(&h_rb->dcs_count, dcs_num.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            wait_stream();
            const uint32_t ng = h_rb->dcs_count;
            if (ng) {
                dcs_h_f4.resize(ng); dcs_h_pts.resize(3 * (size_t)ng); dcs_h_proj.assign(3 * (size_t)ng, 0.0f); dcs_h_inside.assign(ng, 0);
                SALVA_HIP_CHECK(hipMemcpyAsync(dcs_h_f4.data(), dcs_out.p, (size_t)ng * sizeof(float4), hipMemcpyDeviceToHost, stream));
                wait_stream();
                for (uint32_t k = 0; k < ng; ++k) { dcs_h_pts[3 * k] = dcs_h_f4[k].x; dcs_h_pts[3 * k + 1] = dcs_h_f4[k].y; dcs_h_pts[3 * k + 2] = dcs_h_f4[k].z; }
                b.dyn_host.project(b.dyn_host.user, ng, dcs_h_pts.data(), dcs_h_proj.data(), dcs_h_inside.data());
                for (uint32_t k = 0; k < ng; ++k)
                    dcs_h_f4[k] = make_float4(dcs_h_proj[3 * k], dcs_h_proj[3 * k + 1], dcs_h_proj[3 * k + 2], dcs_h_inside[k] ? 1.0f : 0.0f);
                dcs_proj.ensure(ng, stream, false, 1.5f); dcs_cand2.ensure(ng, stream, false, 1.5f);
                SALVA_HIP_CHECK(hipMemcpyAsync(dcs_proj.p, dcs_h_f4.data(), (size_t)ng * sizeof(float4), hipMemcpyHostToDevice, stream));
                // (dcs_out holds the gathered candidates; its compaction goes back into dcs_cand, which is free again)
                launch_dcs_apply(ng, dcs_out.p, dcs_proj.p, posm[cur].p, vel[cur].p, perm[cur].p, comm ? gtag[cur].p : nullptr, prm_d,
                                 dcs_cand2.p, dcs_flag.p, stream);
                select_flagged_f4(cub_temp.p, tb, dcs_cand2.p, dcs_flag.p, dcs_cand.p, dcs_num.p, ng, stream);
                SALVA_HIP_CHECK(hipMemcpyAsync(&h_rb->dcs_count, dcs_num.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                wait_stream();  // (also: dcs_h_f4 has been read)
                cnt = h_rb->dcs_count;
                emit_src = dcs_cand.p;
            }
        } else if (n) {
            const DcsParams prm_d = dcs_params(b.dyn_shape, b.dyn_pose, sc.h, prm.particle_radius, dt_prev);
            dcs_cand.ensure(n, stream, false, 1.1f); dcs_out.ensure(n, stream, false, 1.1f); dcs_flag.ensure(n, stream, false, 1.1f);
            dcs_num.ensure(1);
            launch_dcs_project(n, posm[cur].p, vel[cur].p, G().keys[0].p, perm[cur].p, comm ? gtag[cur].p : nullptr, gv, prm_d, dcs_cand.p,
                               dcs_flag.p, stream);
            const size_t tb = select_flagged_temp_bytes(n);
            ensure_cub_temp(tb);
            select_flagged_f4(cub_temp.p, tb, dcs_cand.p, dcs_flag.p, dcs_out.p, dcs_num.p, n, stream);
            SALVA_HIP_CHECK(hipMemcpyAsync(&h_rb->dcs_count, dcs_num.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            wait_stream();
            cnt = h_rb->dcs_count;
            emit_src = dcs_out.p;
        }
        } catch (const HipError& e) {
            if (!comm) throw;
            local_failed = true; local_error = e; cnt = 0; emit_src = nullptr;
        } catch (const std::exception& e) {
            if (!comm) throw;
            local_failed = true; local_error = HipError(SALVA_HIP_E_INVALID, e.what()); cnt = 0; emit_src = nullptr;
        }
        // Decomposed run: each rank has emitted for the particles it OWNS; every rank then holds every rank's points (a collider's
        // contact layer: thousands of rows), so a boundary particle near a slab face has its whole boundary neighbourhood — its
        // volume — and acts on the fluid of both slabs, wherever its source particle lives.  (Mirroring more ghost planes instead
        // has no bound that holds: the point is the projection of the PREDICTED position, |v| dt + 1.5 h + the penetration depth
        // from its source.)  Each rank's force accumulator receives what its own fluid exerts: the rows are the same on every
        // rank, the per-rank forces add up.
        const uint32_t* emit_models = nullptr;
        if (comm) cnt = dist_gather_emitted(emit_src, cnt, &emit_src, &emit_models, local_failed ? &local_error : nullptr);
        resize_boundary_slot(slot, cnt);
        b_dirty = true;  // same count, new positions
        if (cnt) {
            const uint64_t off = boundary_offset(slot);
            b.dyn_src->ensure(cnt, stream, false, 1.5f);
            launch_dcs_emit(cnt, emit_src, b.dyn_pose, slot, bst_pos.p + off, bst_vel.p + off, b.dyn_src->p, stream);
            if (emit_models) {
                if (!b.dyn_src_model) b.dyn_src_model = std::make_shared<DevBuf<uint32_t>>();
                b.dyn_src_model->ensure(cnt, stream, false, 1.5f);
                SALVA_HIP_CHECK(hipMemcpyAsync(b.dyn_src_model->p, emit_models, (size_t)cnt * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            }
            SALVA_HIP_CHECK(hipMemsetAsync(bforce.p + off, 0, (size_t)cnt * sizeof(float4), stream));  // clear_forces(true) :262
        }
    }
}

// (fluid slot, index) of the fluid particle each point of a dynamically sampled boundary was projected from
void World::get_boundary_sources(uint32_t slot, uint32_t* fluid_slots, uint32_t* indices) {
    use_device();
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    const BoundarySlot& b = bounds[slot];
    if (!b.dyn_kind) throw HipError(SALVA_HIP_E_INVALID, "boundary is not dynamically sampled");
    if (!b.n) return;
    std::vector<uint32_t> src(b.n);
    SALVA_HIP_CHECK(hipMemcpyAsync(src.data(), b.dyn_src->p, b.n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (comm) {  // decomposed run: the global id of the source particle (salva_hip_get_local's `ids`) and its fluid
        if (indices) SALVA_HIP_CHECK(hipMemcpyAsync(indices, b.dyn_src->p, b.n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (fluid_slots) SALVA_HIP_CHECK(hipMemcpyAsync(fluid_slots, b.dyn_src_model->p, b.n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        return;
    }
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    for (uint64_t k = 0; k < b.n; ++k) {
        uint32_t f = 0;
        uint64_t o = 0;
        while (f + 1 < fluids.size() && src[k] >= o + fluids[f].n) { o += fluids[f].n; ++f; }
        if (fluid_slots) fluid_slots[k] = f;
        if (indices) indices[k] = (uint32_t)(src[k] - o);
    }
}

// ------------------------------------------------------------------------------------------------ host force callbacks
__global__ void k_add_acc_from_host(uint32_t n, const uint32_t* __restrict__ perm, uint32_t off, uint32_t nn,
                                    const float* __restrict__ in, float4* __restrict__ acc) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t g = perm[s] - off;  // host index within the fluid (wraps for other fluids)
    if (g >= nn) return;
    float4 a = acc[s];
    a.x += in[3 * g]; a.y += in[3 * g + 1]; a.z += in[3 * g + 2];
    acc[s] = a;
}

void World::force_get_state(uint32_t slot, float* positions, float* velocities, float* densities) {
    use_device();
    if (!in_force_cb) throw HipError(SALVA_HIP_E_INVALID, "only available inside a force callback");
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    if (nn == 0) return;
    scratch_f4.ensure(n);
    scratch_f.ensure(std::max<size_t>(3 * nn, n), stream, false, 1.1f);
    for (int k = 0; k < 2; ++k) {
        float* out = k == 0 ? positions : velocities;
        if (!out) continue;
        // fluid.velocities equal w = v + dv while the forces run (dfsph_solver.rs:688-693)
        launch_unsort_f4(n, perm[cur].p, k == 0 ? last_ctx.posm : last_ctx.w, scratch_f4.p, stream);
        k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f4.p + off, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(out, scratch_f.p, 3 * nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (densities) {
        launch_unsort_f32(n, perm[cur].p, last_ctx.rho, scratch_f.p, stream);
        SALVA_HIP_CHECK(hipMemcpyAsync(densities, scratch_f.p + off, nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
}

void World::force_add_accelerations(uint32_t slot, const float* acc_h) {
    use_device();
    if (!in_force_cb) throw HipError(SALVA_HIP_E_INVALID, "only available inside a force callback");
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    if (!acc_h) throw HipError(SALVA_HIP_E_INVALID, "null accelerations");
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    if (nn == 0) return;
    scratch_f.ensure(3 * nn, stream, false, 1.1f);
    SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, acc_h, 3 * nn * sizeof(float), hipMemcpyHostToDevice, stream));
    k_add_acc_from_host<<<nblk(n), BLOCK, 0, stream>>>(n, perm[cur].p, (uint32_t)off, (uint32_t)nn, scratch_f.p, last_ctx.acc);
    SALVA_HIP_CHECK(hipGetLastError());
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
}

// checkpoint restore: the two pieces of solver state that live next to the fluid arrays (velocity_changes, IISPH pressures)
void World::set_fluid_field(uint32_t slot, int field, const float* data) {
    use_device();
    if (slot >= fluids.size()) throw HipError(SALVA_HIP_E_INVALID, "fluid slot out of range");
    if (!data) throw HipError(SALVA_HIP_E_INVALID, "null data");
    if (field != SALVA_HIP_FIELD_VELOCITY_CHANGE && field != SALVA_HIP_FIELD_PRESSURE)
        throw HipError(SALVA_HIP_E_INVALID, "only velocity_changes and pressures can be set");
    const uint64_t nn = fluids[slot].n, off = fluid_offset(slot);
    if (nn == 0) return;
    ensure_staging_current();
    const size_t width = field == SALVA_HIP_FIELD_VELOCITY_CHANGE ? 3 : 1;
    scratch_f.ensure(width * nn, stream, false, 1.1f);
    SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, data, width * nn * sizeof(float), hipMemcpyHostToDevice, stream));
    if (field == SALVA_HIP_FIELD_VELOCITY_CHANGE) k_pack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f.p, st_dv.p + off, 1, 0.0f);
    else k_pack_w<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, scratch_f.p, st_dv.p + off);
    SALVA_HIP_CHECK(hipGetLastError());
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    sorted_valid = false; bbox_known = false; have_last_ctx = false;
}

void World::get_boundary_particles(uint32_t slot, float* positions, float* velocities) {
    use_device();
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    const uint64_t nn = bounds[slot].n, off = boundary_offset(slot);
    if (nn == 0) return;
    scratch_f.ensure(3 * nn, stream, false, 1.1f);
    for (int k = 0; k < 2; ++k) {
        float* out = k == 0 ? positions : velocities;
        if (!out) continue;
        k_unpack_xyz<<<nblk(nn), BLOCK, 0, stream>>>((uint32_t)nn, (k == 0 ? bst_pos.p : bst_vel.p) + off, scratch_f.p);
        SALVA_HIP_CHECK(hipMemcpyAsync(out, scratch_f.p, 3 * nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
}

void World::get_boundary_wrench(uint32_t slot, const float point[3], float force[3], float torque[3]) {
    use_device();
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    const BoundarySlot& b = bounds[slot];
    for (int k = 0; k < 3; ++k) force[k] = torque[k] = 0.0f;
    if (!b.n || !b.wants_forces) return;
    const uint64_t off = boundary_offset(slot);
    const uint32_t nblocks = (uint32_t)std::min<uint64_t>(nblk(b.n), 256);
    wrench_partial.ensure((size_t)256 * 6);  // per-step call in a coupled run: no allocation on the way
    k_boundary_wrench<<<nblocks, BLOCK, 0, stream>>>((uint32_t)b.n, bst_pos.p + off, bforce.p + off, point[0], point[1], point[2], wrench_partial.p);
    SALVA_HIP_CHECK(hipGetLastError());
    std::vector<double> h((size_t)nblocks * 6);
    SALVA_HIP_CHECK(hipMemcpyAsync(h.data(), wrench_partial.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t k = 0; k < nblocks; ++k)
        for (int j = 0; j < 6; ++j) acc[j] += h[(size_t)k * 6 + j];
    for (int k = 0; k < 3; ++k) { force[k] = (float)acc[k]; torque[k] = (float)acc[3 + k]; }
}

void World::clear_boundary_forces(uint32_t slot) {
    use_device();
    if (slot >= bounds.size()) throw HipError(SALVA_HIP_E_INVALID, "boundary slot out of range");
    const uint64_t nn = bounds[slot].n, off = boundary_offset(slot);
    if (nn) {
        SALVA_HIP_CHECK(hipMemsetAsync(bforce.p + off, 0, nn * sizeof(float4), stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
}

uint64_t World::device_bytes() const {
    uint64_t b = 0;
    auto add = [&](size_t x) { b += x; };
    add(st_pos.bytes()); add(st_vel.bytes()); add(st_dv.bytes()); add(st_acc.bytes()); add(st_model.bytes());
    for (int k = 0; k < 2; ++k) {
        add(posm[k].bytes()); add(vel[k].bytes()); add(dv[k].bytes()); add(model[k].bytes()); add(perm[k].bytes());
        for (const GridTabs& t : gtab) { add(t.keys[k].bytes()); add(t.idx[k].bytes()); }
        add(bkeys[k].bytes()); add(bidx[k].bytes());
    }
    add(acc.bytes()); add(w.bytes()); add(normal.bytes()); add(dii.bytes()); add(dijpj.bytes()); add(iisph_q.bytes()); add(iisph_pr.bytes()); add(posmr.bytes()); add(w2.bytes());
    add(rho.bytes()); add(alpha.bytes()); add(kappa.bytes()); add(kappa2.bytes()); add(rho_star.bytes()); add(aii.bytes());
    add(visc_beta.bytes()); add(visc_target.bytes()); add(visc_u0.bytes()); add(visc_u1.bytes()); add(visc_va.bytes()); add(he_colors.bytes()); add(he_gradcs.bytes());
    add(nff.bytes()); add(nfb.bytes()); add(gtab[0].cell_start_f.bytes()); add(gtab[1].cell_start_f.bytes()); add(halo_src.bytes()); add(bhalo_src.bytes());
    add(nbr_ff.bytes()); add(nbr_fb.bytes()); add(slice_near.bytes()); add(cub_temp.bytes()); add(scratch_f.bytes());
    add(scratch_f4.bytes()); add(bst_pos.bytes()); add(bst_vel.bytes()); add(bposv.bytes()); add(bvel.bytes());
    add(bforce.bytes()); add(bperm.bytes()); add(cell_start_b.bytes()); add(partials.bytes());
    return b;
}

// Average duration (microseconds) of one k_pred_density launch on the last step's lists, by HIP events on the
// world's own stream.  The kernel only rewrites scratch (kappa, partials), so the world's state is unaffected.
float World::time_pred_density(int reps) {
    use_device();
    if (!have_last_ctx || !sorted_valid || n == 0) throw HipError(SALVA_HIP_E_INVALID, "no completed step to time");
    if (reps < 1) reps = 1;
#ifdef SALVA_HIP_DIAG
    if (getenv("SALVA_HIP_TILE_TIMING")) tile_timing_report();
#endif
    flags_clean = false;  // (a between-step launch may raise an error flag: the next step clears them itself — ADVICE r04)
    launch_pred_density(last_ctx, lds, last_dt, stream);  // warm-up
    SALVA_HIP_CHECK(hipEventRecord(ev[0], stream));
    for (int r = 0; r < reps; ++r) launch_pred_density(last_ctx, lds, last_dt, stream);
    SALVA_HIP_CHECK(hipEventRecord(ev[1], stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0.0f;
    SALVA_HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    return ms * 1000.0f / (float)reps;
}


// Average duration (microseconds) of one launch of a neighbour-sum kernel on the last step's lists (HIP events on the world's
// stream): 0 k_pred_density, 1 k_divergence, 2 k_iisph_next_pressure, 3 k_iisph_dij_pj.  Scratch outputs only.
float World::time_kernel(int kernel, int reps) {
    use_device();
    if (!have_last_ctx || !sorted_valid || n == 0) throw HipError(SALVA_HIP_E_INVALID, "no completed step to time");
    if (kernel == 0) return time_pred_density(reps);
    if (reps < 1) reps = 1;
    const bool iisph = prm.solver == SALVA_HIP_SOLVER_IISPH;
    if (kernel == 4 && comm) throw HipError(SALVA_HIP_E_INVALID, "not available in a multi-GPU run");
    if ((kernel == 2 || kernel == 3) && !iisph) throw HipError(SALVA_HIP_E_INVALID, "an IISPH kernel needs an IISPH world");
    if (kernel == 1 && iisph) throw HipError(SALVA_HIP_E_INVALID, "k_divergence needs a DFSPH world");
    StepCtx cd = last_ctx;
    cd.ctl = nullptr;
    flags_clean = false;  // (see time_pred_density)
    if (kernel == 6) {  // the apply pass updates w in place: let it run on a copy (w2 is free outside a speculative solve)
        if (iisph) throw HipError(SALVA_HIP_E_INVALID, "k_divergence_apply needs a DFSPH world");
        w2.ensure(n, stream, false, 1.1f);
        SALVA_HIP_CHECK(hipMemcpyAsync(w2.p, w.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        cd.w = w2.p; cd.w2 = w2.p; cd.spec_k = -1; cd.bforce = nullptr;
    }
    auto launch = [&]() {
        switch (kernel) {
            case 1: launch_divergence(cd, lds, stream); break;
            case 6: launch_divergence_apply(cd, lds, inv_dt_prev, stream); break;
            case 2: launch_iisph_next_pressure(cd, lds, last_dt, 0.5f, kappa.p, kappa2.p, stream); break;
            case 3: launch_iisph_dij_pj(cd, lds, last_dt, kappa.p, stream); break;
            case 4:  // rebuilds the lists on the tables of the last step.  The positions have moved since those were built, so every
                     // particle's cell is taken from its sorted key (as after a DynamicContactSampling push-out): without that a
                     // particle that left its cell indexes the tile's cell table out of range and the kernel never returns.
                cd.stale_keys = G().keys[1].p;
                launch_nbr_build(cd, lds, tile_list_stats.p, reinterpret_cast<unsigned long long*>(&d_rb.p->ncontacts_ff), &d_rb.p->max_cnt_ff, nullptr, stream);
                break;
#ifdef SALVA_HIP_DIAG
            case 5: launch_list_schedule(cd, lds, stream); break;  // re-schedules the (already scheduled or not) lists: same work
#endif
            default: throw HipError(SALVA_HIP_E_INVALID, "unknown kernel id");
        }
    };
    launch();  // warm-up
    SALVA_HIP_CHECK(hipEventRecord(ev[0], stream));
    for (int r = 0; r < reps; ++r) launch();
    SALVA_HIP_CHECK(hipEventRecord(ev[1], stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0.0f;
    SALVA_HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    return ms * 1000.0f / (float)reps;
}

}  // namespace salva
