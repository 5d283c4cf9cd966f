// world_dist.hip — the multi-GPU (x-slab) side of World: migration, ghost planes, per-pass ghost refresh and the
// globally reduced convergence test.  Protocol and sizes: comm.h, dist.h, DESIGN.md §6.
#include <algorithm>
#include <climits>
#include <cstring>

#include <string>

#include "dcs.h"
#include "world.h"

namespace salva {

__global__ void k_iota_u32(uint32_t n, uint32_t base, uint32_t* out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = base + i;
}
void launch_iota_u32(uint32_t n, uint32_t base, uint32_t* out, hipStream_t s) {
    if (n) k_iota_u32<<<div_up(n, BLOCK), BLOCK, 0, s>>>(n, base, out);
}
__global__ void k_pack_owned(uint32_t n, const float4* __restrict__ posm, const float4* __restrict__ vel,
                             const uint32_t* __restrict__ gtag, const uint32_t* __restrict__ gid, const uint32_t* __restrict__ model,
                             unsigned int* counter, uint32_t cap, uint32_t* __restrict__ out_gid, float* __restrict__ out_pos,
                             float* __restrict__ out_vel, uint32_t* __restrict__ out_model) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (gtag[i] & GTAG_GHOST)) return;
    const uint32_t k = atomicAdd(counter, 1u);
    if (k >= cap) return;
    const float4 p = posm[i], v = vel[i];
    out_gid[k] = gid[i];
    out_model[k] = model[i];
    out_pos[3 * k] = p.x; out_pos[3 * k + 1] = p.y; out_pos[3 * k + 2] = p.z;
    out_vel[3 * k] = v.x; out_vel[3 * k + 1] = v.y; out_vel[3 * k + 2] = v.z;
}

void World::set_domain(Transport* transport, int lo, int hi, uint32_t gid_off) {
    if (!transport) throw HipError(SALVA_HIP_E_INVALID, "null transport");
    if (hi - lo + 1 < 2 * GHOST_PLANES) throw HipError(SALVA_HIP_E_INVALID, "a slab must span at least four cell planes");
    comm = transport;
    slab_lo = lo; slab_hi = hi; gid_offset = gid_off;
    sorted_valid = false; bbox_known = false; dist_started = false; tables_dirty = true; nbr_bounds_valid = false;
}

// Load balancing ("re-cut every M steps").  All ranks call it between two steps.  Every rank contributes the
// histogram of its owned particles over the cell planes; the sums are all-reduced, and every rank computes the same new
// cuts: the plane where the running count passes k N / size, kept between the old cuts on either side (so that a particle
// changes owner by at most one rank, which is what the migration phase can move) and at least four planes apart.
void World::rebalance(int32_t* new_lo, int32_t* new_hi) {
    use_device();
    if (!comm) throw HipError(SALVA_HIP_E_INVALID, "rebalance needs a domain (salva_hip_set_domain)");
    if (!dist_started || !bbox_known) throw HipError(SALVA_HIP_E_INVALID, "rebalance needs a completed step");
    const int size = comm->size(), rank = comm->rank();
    const unsigned long long OFF = 1ull << 40;
    // every rank's plane range: its slab, widened to what it actually holds at an open end
    const int my_lo = comm->has_lo() ? slab_lo : std::min(slab_lo, (int)h_rb->bbox[0]);
    const int my_hi = comm->has_hi() ? slab_hi : std::max(slab_hi, (int)h_rb->bbox[3]);
    std::vector<unsigned long long> rng(2 * (size_t)size, 0ull);
    rng[2 * rank] = (unsigned long long)((long long)my_lo + (long long)OFF);
    rng[2 * rank + 1] = (unsigned long long)((long long)my_hi + (long long)OFF);
    plane_hist.ensure(std::max<size_t>(2 * (size_t)size, 64));
    SALVA_HIP_CHECK(hipMemcpyAsync(plane_hist.p, rng.data(), rng.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    comm->allreduce_sum_u64(plane_hist.p, 2 * size, stream);
    SALVA_HIP_CHECK(hipMemcpyAsync(rng.data(), plane_hist.p, rng.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<long long> lo(size), hi(size);
    for (int r = 0; r < size; ++r) { lo[r] = (long long)rng[2 * r] - (long long)OFF; hi[r] = (long long)rng[2 * r + 1] - (long long)OFF; }
    const long long base = lo[0], len = hi[size - 1] - lo[0] + 1;
    if (len <= 0 || len > (1ll << 22)) throw HipError(SALVA_HIP_E_CAPACITY, "rebalance: the domain spans too many cell planes");
    plane_hist.ensure((size_t)len);
    SALVA_HIP_CHECK(hipMemsetAsync(plane_hist.p, 0, (size_t)len * sizeof(unsigned long long), stream));
    launch_plane_hist(n, posm[cur].p, gtag[cur].p, sc.h, (int)base, (int)len, plane_hist.p, stream);
    comm->allreduce_sum_u64(plane_hist.p, (int)len, stream);
    std::vector<unsigned long long> hist((size_t)len);
    SALVA_HIP_CHECK(hipMemcpyAsync(hist.data(), plane_hist.p, (size_t)len * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    unsigned long long total = 0;
    for (auto v : hist) total += v;
    // cut[r] = first plane of rank r (r = 1 .. size-1); old cuts are the current slab_lo's
    std::vector<long long> cut(size + 1);
    cut[0] = lo[0]; cut[size] = hi[size - 1] + 1;
    unsigned long long run = 0;
    long long p = 0;
    for (int r = 1; r < size; ++r) {
        const unsigned long long want = total * (unsigned long long)r / (unsigned long long)size;
        while (p < len && run + hist[(size_t)p] <= want) run += hist[(size_t)p++];
        long long cpl = base + p;
        const long long old_prev = (r - 1 >= 1) ? lo[r - 1] : LLONG_MIN / 4, old_next = (r + 1 < size) ? lo[r + 1] : LLONG_MAX / 4;
        // A particle changes owner by one rank at most — including the particles that crossed a face during the last step and
        // have not migrated yet (they sit up to GHOST_PLANES planes inside the neighbour's old slab): a cut therefore stays
        // GHOST_PLANES planes clear of the old cuts on either side (two adjacent slabs span >= 8 planes, so the range is never empty).
        cpl = std::max(cpl, old_prev + (long long)GHOST_PLANES);
        cpl = std::min(cpl, old_next - (long long)GHOST_PLANES);
        cpl = std::max(cpl, cut[r - 1] + 2 * (long long)GHOST_PLANES);     // slabs stay >= four planes thick
        cut[r] = cpl;
    }
    for (int r = size - 1; r >= 1; --r) cut[r] = std::min(cut[r], cut[r + 1] - 2 * (long long)GHOST_PLANES);
    for (int r = 1; r < size; ++r)
        if (cut[r] < cut[r - 1] + 2 * (long long)GHOST_PLANES || cut[r] < ((r - 1 >= 1) ? lo[r - 1] + (long long)GHOST_PLANES : LLONG_MIN / 4))
            throw HipError(SALVA_HIP_E_CAPACITY, "rebalance: the domain is too short for four cell planes per rank");
    if (comm->has_lo()) slab_lo = (int)cut[rank];
    if (comm->has_hi()) slab_hi = (int)cut[rank + 1] - 1;
    nbr_bounds_valid = false;
    // The next step's arrivals are not "at most two planes beyond a face and inside the sender's previous y/z box" any more
    // (dist_prepare's shortcut): migrants come from as deep inside a neighbour as its cut moved, and the planes a neighbour
    // mirrors may hold particles it has just received from ITS far neighbour.  Every rank therefore announces "no box" in the
    // next exchange and the step reduces the cell bounding box over the particles once (World::step).
    bbox_known = false;
    if (new_lo) *new_lo = slab_lo;
    if (new_hi) *new_hi = slab_hi;
}

DistArrays World::dist_arrays(int which) {
    return DistArrays{posm[which].p, vel[which].p, dv[which].p, model[which].p, perm[which].p, gtag[which].p};
}

// posm / vel / dv / model / perm(gid) / gtag in both buffers, keys and idx: capacity for `cap` particles, contents kept
void World::ensure_particle_capacity(size_t cap) {
    const float slack = comm ? 1.25f : 1.0f;
    for (int k = 0; k < 2; ++k) {
        posm[k].ensure(cap, stream, true, slack); vel[k].ensure(cap, stream, true, slack); dv[k].ensure(cap, stream, true, slack);
        model[k].ensure(cap, stream, true, slack); perm[k].ensure(cap, stream, true, slack);
        for (GridTabs& t : gtab) { t.keys[k].ensure(cap, stream, false, slack); t.idx[k].ensure(cap, stream, false, slack); }
        if (comm) gtag[k].ensure(cap, stream, true, slack);
    }
}

// Phase 1: drop last step's ghosts, send away what left the slab, take in what entered.  Phase 2: mirror the edge
// planes on the neighbours.  On return the arrays of `cur` hold [owned | ghosts from lo | ghosts from hi] and n counts all.
void World::dist_prepare() {
    // the per-fluid particle counts of the error averages are global and constant (no creation / deletion here)
    if (!dist_started) {
        const uint32_t nm = (uint32_t)std::max<size_t>(fluids.size(), 1);
        std::vector<unsigned long long> cnt(nm, 0);
        for (uint32_t f = 0; f < fluids.size(); ++f) cnt[f] = fluids[f].n;
        d_counters.ensure(std::max<size_t>(4, nm));
        SALVA_HIP_CHECK(hipMemcpyAsync(d_counters.p, cnt.data(), nm * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
        comm->allreduce_sum_u64(d_counters.p, (int)nm, stream);
        SALVA_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_counters.p, nm * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<uint32_t> c32(nm);
        for (uint32_t f = 0; f < nm; ++f) {
            if (cnt[f] >= 0xffffffffull) throw HipError(SALVA_HIP_E_CAPACITY, "more than 2^32 particles in one fluid");
            c32[f] = (uint32_t)cnt[f];
        }
        global_counts = c32;
        SALVA_HIP_CHECK(hipMemcpyAsync(model_counts.p, c32.data(), nm * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        d_sums.ensure(nm);
        // the first global id no particle has: where dist_add_particles continues numbering
        {
            const int size = comm->size(), rank = comm->rank();
            std::vector<unsigned long long> top((size_t)size, 0ull);
            top[rank] = (unsigned long long)gid_offset + n;
            plane_hist.ensure(std::max<size_t>((size_t)size, 64));
            SALVA_HIP_CHECK(hipMemcpyAsync(plane_hist.p, top.data(), top.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
            comm->allreduce_sum_u64(plane_hist.p, size, stream);
            SALVA_HIP_CHECK(hipMemcpyAsync(top.data(), plane_hist.p, top.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            SALVA_HIP_CHECK(hipStreamSynchronize(stream));
            gid_next = 0;
            for (auto t : top) gid_next = std::max<uint64_t>(gid_next, t);
        }
        dist_started = true;
    }
    const bool has_lo = comm->has_lo(), has_hi = comm->has_hi();
    if (!has_lo && !has_hi) {
        // a communicator of one rank: nobody to hand particles to or to mirror — the set stays as it is (every particle owned,
        // no ghost), and the step runs the single-domain schedule (finalize_solve, evaluate_split)
        n_owned = n;
        nborder_lo = nborder_hi = nghost_lo = nghost_hi = 0;
        send_lo_idx.ensure(1); send_hi_idx.ensure(1); ghost_lo_idx.ensure(1); ghost_hi_idx.ensure(1);
        fbuf_send.ensure(1); fbuf_recv.ensure(1);
        return;
    }
    if (!nbr_bounds_valid) {
        // the far ends of the neighbours' slabs: a leaver must land inside the adjacent slab (k_dist_flags, flag 4)
        const unsigned long long OFF = 1ull << 40;
        const uint64_t mine[2] = {(uint64_t)((long long)(has_lo ? slab_lo : INT32_MIN) + (long long)OFF),
                                  (uint64_t)((long long)(has_hi ? slab_hi : INT32_MAX) + (long long)OFF)};
        uint64_t from_lo[2] = {0, 0}, from_hi[2] = {0, 0};
        comm->exchange_counts(mine, mine, from_lo, from_hi, stream);
        nbr_lo_lo = has_lo ? (int)((long long)from_lo[0] - (long long)OFF) : INT32_MIN;
        nbr_hi_hi = has_hi ? (int)((long long)from_hi[1] - (long long)OFF) : INT32_MAX;
        nbr_bounds_valid = true;
    }
    // The cell bounding box of the set this step works on, without a reduction pass over it: h_rb->bbox (if valid) covers
    // every particle this rank held at the end of the last step, and what arrives — migrants and ghost planes — lies in the
    // two cell planes beyond a face at most and inside the sender's own box in y and z, which rides along in the spare
    // word of the two count exchanges (mode 1: y range, mode 2: z range; all-ones = "my box is not valid").
    const bool had_bbox = bbox_known;
    const uint64_t NO_BOX = ~0ull;
    auto pack_range = [&](int a) -> uint64_t {
        if (!had_bbox) return NO_BOX;
        return ((uint64_t)(uint32_t)h_rb->bbox[a] << 32) | (uint64_t)(uint32_t)h_rb->bbox[3 + a];
    };
    int32_t merged[6];
    memcpy(merged, h_rb->bbox, sizeof(merged));
    bool box_ok = had_bbox;
    auto merge_range = [&](int a, uint64_t w, bool present) {
        if (!present) return;
        if (w == NO_BOX) { box_ok = false; return; }
        merged[a] = std::min(merged[a], (int)(int32_t)(uint32_t)(w >> 32));
        merged[3 + a] = std::max(merged[3 + a], (int)(int32_t)(uint32_t)(w & 0xffffffffull));
    };
    for (int mode = 1; mode <= 2; ++mode) {
        dsel.ensure(dist_sel_bytes(n), stream, false, 1.25f);
        dpos.ensure(dist_sel_bytes(n), stream, false, 1.25f);
        const size_t tb = dist_scan_temp_bytes(n + 1);
        ensure_cub_temp(tb);
        uint32_t tot[3] = {0, 0, 0};
        launch_dist_select(n, posm[cur].p, gtag[cur].p, sc.h, slab_lo, slab_hi, has_lo, has_hi, mode, nbr_lo_lo, nbr_hi_hi, dsel.p, dpos.p,
                           cub_temp.p, tb, d_flags.p, tot, stream);
        xsend_lo.ensure(std::max<uint32_t>(tot[1], 1u), stream, false, 1.25f);
        xsend_hi.ensure(std::max<uint32_t>(tot[2], 1u), stream, false, 1.25f);
        launch_dist_pack(n, dist_arrays(cur), dist_arrays(cur ^ 1), sc.h, slab_lo, slab_hi, has_lo, has_hi, mode, nbr_lo_lo, nbr_hi_hi, dpos.p,
                         xsend_lo.p, xsend_hi.p, stream);
        const uint64_t range = pack_range(mode);  // (axis 1 = y with the migrants, axis 2 = z with the ghost planes)
        const uint64_t to_lo[2] = {tot[1], range}, to_hi[2] = {tot[2], range};
        uint64_t from_lo[2] = {0, 0}, from_hi[2] = {0, 0};
        comm->exchange_counts(to_lo, to_hi, from_lo, from_hi, stream);
        merge_range(mode, from_lo[1], has_lo);
        merge_range(mode, from_hi[1], has_hi);
        const uint32_t base = (mode == 1) ? tot[0] : n;
        const uint64_t total = (uint64_t)base + from_lo[0] + from_hi[0];
        if (total >= 0xfffffff0ull) throw HipError(SALVA_HIP_E_CAPACITY, "too many particles in one slab");
        ensure_particle_capacity(total);
        xrecv_lo.ensure(std::max<uint64_t>(from_lo[0], 1), stream, false, 1.25f);
        xrecv_hi.ensure(std::max<uint64_t>(from_hi[0], 1), stream, false, 1.25f);
        comm->sendrecv(xsend_lo.p, tot[1] * sizeof(DistRec), xsend_hi.p, tot[2] * sizeof(DistRec), xrecv_lo.p,
                       from_lo[0] * sizeof(DistRec), xrecv_hi.p, from_hi[0] * sizeof(DistRec), stream);
        if (mode == 1) cur ^= 1;  // the kept particles were compacted into the other buffer
        DistArrays dst = dist_arrays(cur);
        launch_dist_unpack((uint32_t)from_lo[0], base, xrecv_lo.p, dst, mode == 2 ? (GTAG_GHOST | GTAG_BORDER_LO) : 0u, stream);
        launch_dist_unpack((uint32_t)from_hi[0], base + (uint32_t)from_lo[0], xrecv_hi.p, dst,
                           mode == 2 ? (GTAG_GHOST | GTAG_BORDER_HI) : 0u, stream);
        if (mode == 1) {
            n = (uint32_t)total;
            n_owned = n;
        } else {
            nborder_lo = tot[1]; nborder_hi = tot[2];
            nghost_lo = (uint32_t)from_lo[0]; nghost_hi = (uint32_t)from_hi[0];
            n = (uint32_t)total;
        }
    }
    if (n == 0) throw HipError(SALVA_HIP_E_INVALID, "a slab without any particle is not supported");
    send_lo_idx.ensure(std::max(nborder_lo, 1u), stream, false, 1.25f); send_hi_idx.ensure(std::max(nborder_hi, 1u), stream, false, 1.25f);
    ghost_lo_idx.ensure(std::max(nghost_lo, 1u), stream, false, 1.25f); ghost_hi_idx.ensure(std::max(nghost_hi, 1u), stream, false, 1.25f);
    fbuf_send.ensure(std::max(nborder_lo + nborder_hi, 1u), stream, false, 1.25f);
    fbuf_recv.ensure(std::max(nghost_lo + nghost_hi, 1u), stream, false, 1.25f);
    if (box_ok) {
        if (has_lo) merged[0] = std::min(merged[0], slab_lo - GHOST_PLANES);
        if (has_hi) merged[3] = std::max(merged[3], slab_hi + GHOST_PLANES);
        memcpy(h_rb->bbox, merged, sizeof(merged));
        bbox_known = true;
    } else {
        bbox_known = false;  // first step / after host edits: reduce the box over the particles (World::step)
    }
}

// DynamicContactSampling in a decomposed run: this rank's emitted rows (point, sorted index of the source) -> the table of all
// ranks' rows in rank order, the same on every rank.  The transport has sums, not gathers: every rank writes its section of a
// zeroed table and the sections are added as 64-bit integers — x + 0 is exact on the bit patterns, and the 32-bit fluid words of
// two ranks that share a 64-bit word cannot carry into each other.  Two all-reduces (counts, rows) per sampled collider and
// step; collective — every rank registers the same dynamically sampled boundaries in the same slots.
uint32_t World::dist_gather_emitted(const float4* rows, uint32_t cnt, const float4** all_rows, const uint32_t** all_models,
                                    const HipError* local_error) {
    const int size = comm->size(), rank = comm->rank();
    // counts[size] = number of ranks whose local part of the pass failed (a NaN box from the host's aabb callback, an allocation
    // failure ...): it rides in the count all-reduce, so that every rank leaves the step together instead of one rank throwing
    // while the others wait in the collective (same pattern as dist_add_particles)
    std::vector<unsigned long long> counts((size_t)size + 1, 0ull);
    counts[rank] = local_error ? 0ull : cnt;
    counts[size] = local_error ? 1ull : 0ull;
    if (size > 1) {
        plane_hist.ensure(std::max<size_t>((size_t)size + 1, 64));
        SALVA_HIP_CHECK(hipMemcpyAsync(plane_hist.p, counts.data(), counts.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
        comm->allreduce_sum_u64(plane_hist.p, size + 1, stream);
        SALVA_HIP_CHECK(hipMemcpyAsync(counts.data(), plane_hist.p, counts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (local_error) throw *local_error;
    if (counts[size])
        throw HipError(SALVA_HIP_E_INVALID, "dynamic contact sampling failed on another rank of the decomposed world (its own error says why); "
                                            "every rank leaves the step");
    unsigned long long total = 0, before = 0;
    for (int r = 0; r < size; ++r) { if (r < rank) before += counts[r]; total += counts[r]; }
    if (total >= 0x3fffffffull) throw HipError(SALVA_HIP_E_CAPACITY, "too many dynamically sampled boundary particles");
    *all_rows = nullptr; *all_models = nullptr;
    if (total == 0) return 0;
    const size_t words = 2 * (size_t)total + ((size_t)total + 1) / 2;  // float4 rows, then uint32 fluids
    unsigned long long maxc = 0;
    for (int r = 0; r < size; ++r) maxc = std::max(maxc, counts[r]);
    // every rank's section — rows, then fluids — in a block of one stride, gathered (Transport::allgather_u64), then copied side by
    // side into the table every rank builds its boundary from.  (Round 5 summed a zeroed table through the sum all-reduce: over the
    // peer transport a 256-value pass per 256 table words, whatever the number of ranks.)
    const size_t blk = (2 * (size_t)maxc + ((size_t)maxc + 1) / 2 + 1) & ~(size_t)1;  // (even: float4 rows start every section)
    const size_t words_al = (words + 1) & ~(size_t)1;
    dcs_all.ensure(words_al + blk * ((size_t)size + 1), stream, false, 1.5f);
    float4* out_rows = reinterpret_cast<float4*>(dcs_all.p);
    uint32_t* out_models = reinterpret_cast<uint32_t*>(dcs_all.p + 2 * (size_t)total);
    if (size == 1) {
        launch_dcs_pack(cnt, rows, perm[cur].p, model[cur].p, out_rows, out_models, stream);
    } else {
        if (blk >= ((size_t)1 << 30)) throw HipError(SALVA_HIP_E_CAPACITY, "too many dynamically sampled boundary particles");
        unsigned long long* mine = dcs_all.p + words_al;
        unsigned long long* all = mine + blk;
        SALVA_HIP_CHECK(hipMemsetAsync(mine, 0, blk * sizeof(unsigned long long), stream));
        launch_dcs_pack(cnt, rows, perm[cur].p, model[cur].p, reinterpret_cast<float4*>(mine), reinterpret_cast<uint32_t*>(mine + 2 * (size_t)maxc), stream);
        comm->allgather_u64(mine, all, (int)blk, stream);
        unsigned long long at = 0;
        for (int r = 0; r < size; ++r) {
            const unsigned long long* src = all + (size_t)r * blk;
            if (counts[r]) {
                SALVA_HIP_CHECK(hipMemcpyAsync(out_rows + at, src, (size_t)counts[r] * sizeof(float4), hipMemcpyDeviceToDevice, stream));
                SALVA_HIP_CHECK(hipMemcpyAsync(out_models + at, src + 2 * (size_t)maxc, (size_t)counts[r] * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            }
            at += counts[r];
        }
    }
    (void)before;
    *all_rows = out_rows; *all_models = out_models;
    return (uint32_t)total;
}

void World::dist_build_lists() {
    launch_dist_lists(n, gtag[cur].p, send_lo_idx.p, send_hi_idx.p, ghost_lo_idx.p, ghost_hi_idx.p, stream);
}

// Refresh one per-particle field of the ghosts from its owners: gather the mirrored edge-plane particles into a dense
// buffer, one sendrecv with both neighbours, scatter into the ghost slots.  All on the world's stream.
// ---- event pairs around the exchanges of a step (world.h dist_times); only while the stage timers are on
size_t World::dist_time_begin(int kind) {
    if (!prm.enable_timers) return (size_t)-1;
    if (dist_ev_used + 2 > dist_ev.size()) {
        const size_t old = dist_ev.size();
        dist_ev.resize(old + 64, nullptr);
        for (size_t k = old; k < dist_ev.size(); ++k) SALVA_HIP_CHECK(hipEventCreate(&dist_ev[k]));
    }
    const size_t first = dist_ev_used;
    dist_ev_used += 2;
    dist_ev_pairs.emplace_back(first, kind);
    SALVA_HIP_CHECK(hipEventRecord(dist_ev[first], stream));
    return first;
}
void World::dist_time_end(size_t first) {
    if (first != (size_t)-1) SALVA_HIP_CHECK(hipEventRecord(dist_ev[first + 1], stream));
}
void World::dist_time_fold() {
    dist_times[0] = dist_times[1] = dist_times[2] = dist_times[3] = 0.0;
    for (const auto& pr : dist_ev_pairs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, dist_ev[pr.first], dist_ev[pr.first + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
        dist_times[2 * pr.second] += ms;
        dist_times[2 * pr.second + 1] += 1.0;
    }
    dist_ev_pairs.clear();
    dist_ev_used = 0;
}

void World::refresh_f32(float* field) {
    if (!comm->has_lo() && !comm->has_hi()) return;
    SALVA_HIP_CHECK(hipEventRecord(ev_pre_refresh, stream));  // (what evaluate_split lets the interior tiles start after)
    const size_t tm = dist_time_begin(0);
    float* sb = reinterpret_cast<float*>(fbuf_send.p);
    float* rb = reinterpret_cast<float*>(fbuf_recv.p);
    launch_gather_f32(nborder_lo, send_lo_idx.p, field, sb, stream);
    launch_gather_f32(nborder_hi, send_hi_idx.p, field, sb + nborder_lo, stream);
    comm->sendrecv(sb, nborder_lo * sizeof(float), sb + nborder_lo, nborder_hi * sizeof(float), rb, nghost_lo * sizeof(float),
                   rb + nghost_lo, nghost_hi * sizeof(float), stream);
    launch_scatter_f32(nghost_lo, ghost_lo_idx.p, rb, field, stream);
    launch_scatter_f32(nghost_hi, ghost_hi_idx.p, rb + nghost_lo, field, stream);
    dist_time_end(tm);
}
void World::refresh_f4(float4* field) {
    if (!comm->has_lo() && !comm->has_hi()) return;
    SALVA_HIP_CHECK(hipEventRecord(ev_pre_refresh, stream));
    const size_t tm = dist_time_begin(0);
    float4* sb = fbuf_send.p;
    float4* rb = fbuf_recv.p;
    launch_gather_idx_f4(nborder_lo, send_lo_idx.p, field, sb, stream);
    launch_gather_idx_f4(nborder_hi, send_hi_idx.p, field, sb + nborder_lo, stream);
    comm->sendrecv(sb, nborder_lo * sizeof(float4), sb + nborder_lo, nborder_hi * sizeof(float4), rb, nghost_lo * sizeof(float4),
                   rb + nghost_lo, nghost_hi * sizeof(float4), stream);
    launch_scatter_idx_f4(nghost_lo, ghost_lo_idx.p, rb, field, stream);
    launch_scatter_idx_f4(nghost_hi, ghost_hi_idx.p, rb + nghost_lo, field, stream);
    dist_time_end(tm);
}

// Error reduction + break test of an iterative solve; with a transport the per-fluid sums are all-reduced first.
void World::finalize_solve(SolveCtl* ctl, SolveCtl* pub, uint32_t skipped) {
    const unsigned ntiles = nlaunch;  // one partial per launched (non-empty) tile
    const uint32_t nm = (uint32_t)std::max<size_t>(fluids.size(), 1);
    if (!comm || comm->size() == 1) {
        // (folding this into the evaluate kernels through a last-workgroup reduction was measured 7x slower: the
        // device-scope release every workgroup needs writes the XCD's whole L2 back)
        launch_finalize_error(partials.p, ntiles, nm, model_counts.p, ctl, pub, stream, nullptr, nullptr, 0u, skipped);
        return;
    }
    const size_t tm = dist_time_begin(1);
    launch_sum_partials(partials.p, ntiles, nm, ctl, d_sums.p, stream);
    comm->allreduce_sum_f32(d_sums.p, (int)nm, stream);
    launch_decide(d_sums.p, nm, model_counts.p, ctl, pub, stream, skipped);
    dist_time_end(tm);
}

// ---- particle creation and removal in a running decomposed world.  Both are COLLECTIVE: every rank calls them between the
// same two steps (with nothing to add / delete where it has nothing), because the per-fluid particle counts the error
// averages divide by are global (compute_divergences / compute_predicted_densities: `max_error / num_particles`,
// dfsph_solver.rs:160,:355) and the new particles' global ids must not collide.
__global__ __launch_bounds__(BLOCK) void k_dist_append(uint32_t n_add, uint32_t at, const float* __restrict__ pos, const float* __restrict__ vel_in,
                                                       float vol, float rho0, uint32_t slot, uint32_t gid0, DistArrays a) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_add) return;
    const uint32_t i = at + k;
    a.posm[i] = make_float4(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2], vol * rho0);  // particle_mass = volume * density0 (fluid.rs:183-185)
    a.vel[i] = vel_in ? make_float4(vel_in[3 * k], vel_in[3 * k + 1], vel_in[3 * k + 2], vol) : make_float4(0.0f, 0.0f, 0.0f, vol);
    a.dv[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    a.model[i] = slot;
    a.gid[i] = gid0 + k;
    a.gtag[i] = 0u;
}

// per-fluid count changes of this rank -> the global counts, on every rank
void World::dist_update_counts(const std::vector<long long>& delta) {
    const uint32_t nm = (uint32_t)std::max<size_t>(fluids.size(), 1);
    std::vector<unsigned long long> buf(2 * (size_t)nm, 0ull);  // [added | removed]
    for (uint32_t f = 0; f < nm && f < delta.size(); ++f) {
        if (delta[f] >= 0) buf[f] = (unsigned long long)delta[f];
        else buf[nm + f] = (unsigned long long)(-delta[f]);
    }
    plane_hist.ensure(std::max<size_t>(2 * (size_t)nm, 64));
    SALVA_HIP_CHECK(hipMemcpyAsync(plane_hist.p, buf.data(), buf.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    comm->allreduce_sum_u64(plane_hist.p, (int)buf.size(), stream);
    SALVA_HIP_CHECK(hipMemcpyAsync(buf.data(), plane_hist.p, buf.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    if (global_counts.size() != nm) global_counts.assign(nm, 0u);
    for (uint32_t f = 0; f < nm; ++f) {
        const unsigned long long c = (unsigned long long)global_counts[f] + buf[f] - std::min<unsigned long long>(buf[nm + f], (unsigned long long)global_counts[f] + buf[f]);
        if (c >= 0xffffffffull) throw HipError(SALVA_HIP_E_CAPACITY, "more than 2^32 particles in one fluid");
        global_counts[f] = (uint32_t)c;
    }
    SALVA_HIP_CHECK(hipMemcpyAsync(model_counts.p, global_counts.data(), nm * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
}

// Fluid::add_particles (fluid.rs:126-150) on the rank that owns the place (a particle may lie up to one slab away: the next
// step's migration hands it over).  Ids continue after the largest id in the run, rank by rank.
void World::dist_add_particles(uint32_t slot, uint64_t n_add, const float* pos, const float* vel_h) {
    const int size = comm->size(), rank = comm->rank();
    // Whatever can fail on ONE rank is tried before the collective and travels in it as a flag: a rank that threw on
    // its own after the all-reduces would leave the others in the next collective, with the global ids and counts already advanced.
    std::string local_error;
    if (n_add && !pos) local_error = "positions are required";
    else if (slot >= fluids.size()) local_error = "fluid slot out of range";
    else if ((uint64_t)n + n_add >= 0xfffffff0ull) local_error = "too many particles in one slab";
    else {
        try {
            ensure_particle_capacity((size_t)n + n_add);
            scratch_f.ensure(6 * std::max<uint64_t>(n_add, 1), stream, false, 1.1f);
        } catch (const HipError& e) {
            local_error = e.what();
        }
    }
    std::vector<unsigned long long> adds((size_t)size + 1, 0ull);
    adds[rank] = local_error.empty() ? n_add : 0ull;
    adds[size] = local_error.empty() ? 0ull : 1ull;  // ranks that cannot take their particles
    plane_hist.ensure(std::max<size_t>((size_t)size + 1, 64));
    SALVA_HIP_CHECK(hipMemcpyAsync(plane_hist.p, adds.data(), adds.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    comm->allreduce_sum_u64(plane_hist.p, size + 1, stream);
    SALVA_HIP_CHECK(hipMemcpyAsync(adds.data(), plane_hist.p, adds.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    if (adds[size] != 0)  // every rank leaves here, together, with nothing changed
        throw HipError(local_error.empty() ? SALVA_HIP_E_CAPACITY : SALVA_HIP_E_INVALID,
                       local_error.empty() ? "add_particles failed on another rank of the run: nothing was added anywhere" : local_error);
    uint64_t before = 0, total = 0;
    for (int r = 0; r < size; ++r) { if (r < rank) before += adds[r]; total += adds[r]; }
    if (gid_next + total >= 0xfffffff0ull) throw HipError(SALVA_HIP_E_CAPACITY, "global particle ids exhausted");
    const uint64_t gid0 = gid_next + before;
    gid_next += total;
    std::vector<long long> delta(std::max<size_t>(fluids.size(), 1), 0);
    delta[slot] = (long long)n_add;
    dist_update_counts(delta);
    if (n_add == 0) return;
    SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p, pos, 3 * n_add * sizeof(float), hipMemcpyHostToDevice, stream));
    if (vel_h) SALVA_HIP_CHECK(hipMemcpyAsync(scratch_f.p + 3 * n_add, vel_h, 3 * n_add * sizeof(float), hipMemcpyHostToDevice, stream));
    const float r = prm.particle_radius, vol = r * r * r * 6.4f;  // Fluid::particle_volume default (fluid.rs:110-120)
    k_dist_append<<<div_up(n_add, BLOCK), BLOCK, 0, stream>>>((uint32_t)n_add, n, scratch_f.p, vel_h ? scratch_f.p + 3 * n_add : nullptr, vol,
                                                              fluids[slot].density0, slot, (uint32_t)gid0, dist_arrays(cur));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));  // (host buffers)
    n += (uint32_t)n_add;
    n_owned += (uint32_t)n_add;
    bbox_known = false;      // the new particles may lie outside last step's box
    have_last_ctx = false;   // lists and tables describe the old set
}

// `Fluid::delete_particle_at_next_timestep` by global id (host indices do not exist in a decomposed run): the particles of
// `gids` this rank owns become ghosts, which the next step's first phase drops like every other ghost.
__global__ __launch_bounds__(BLOCK) void k_dist_mark_deleted(uint32_t n, const uint32_t* __restrict__ gid, uint32_t* __restrict__ gtag,
                                                             const uint32_t* __restrict__ model, const uint32_t* __restrict__ sorted_ids,
                                                             uint32_t n_ids, unsigned long long* __restrict__ removed) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (gtag[i] & GTAG_GHOST)) return;
    const uint32_t g = gid[i];
    uint32_t lo = 0, hi = n_ids;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted_ids[mid] < g) lo = mid + 1; else hi = mid; }
    if (lo < n_ids && sorted_ids[lo] == g) {
        gtag[i] |= GTAG_GHOST;
        atomicAdd(&removed[model[i]], 1ull);
    }
}
uint64_t World::delete_owned(uint32_t n_ids, const uint32_t* gids) {
    use_device();
    if (!comm) throw HipError(SALVA_HIP_E_INVALID, "delete_owned is for multi-GPU worlds (a single domain deletes by host index: salva_hip_delete_particles)");
    if (!dist_started || !sorted_valid) throw HipError(SALVA_HIP_E_INVALID, "no step has run yet");
    if (n_ids && !gids) throw HipError(SALVA_HIP_E_INVALID, "null id list");
    const uint32_t nm = (uint32_t)std::max<size_t>(fluids.size(), 1);
    std::vector<long long> delta(nm, 0);
    if (n_ids && n) {
        std::vector<uint32_t> ids(gids, gids + n_ids);
        std::sort(ids.begin(), ids.end());
        DevBuf<uint32_t> d_ids;
        d_ids.ensure(n_ids);
        d_counters.ensure(std::max<size_t>(4, nm));
        SALVA_HIP_CHECK(hipMemcpyAsync(d_ids.p, ids.data(), (size_t)n_ids * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        SALVA_HIP_CHECK(hipMemsetAsync(d_counters.p, 0, nm * sizeof(unsigned long long), stream));
        k_dist_mark_deleted<<<div_up(n, BLOCK), BLOCK, 0, stream>>>(n, perm[cur].p, gtag[cur].p, model[cur].p, d_ids.p, n_ids, d_counters.p);
        std::vector<unsigned long long> rem(nm, 0ull);
        SALVA_HIP_CHECK(hipMemcpyAsync(rem.data(), d_counters.p, nm * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        SALVA_HIP_CHECK(hipStreamSynchronize(stream));
        for (uint32_t f = 0; f < nm; ++f) { delta[f] = -(long long)rem[f]; n_owned -= (uint32_t)rem[f]; }
        have_last_ctx = false;
    }
    dist_update_counts(delta);
    return n_owned;
}

// Download the particles this rank owns (unordered): global ids, positions, velocities, fluid slot.  Returns the count.
uint64_t World::get_owned(uint32_t cap, uint32_t* gids, float* pos, float* vel_out, uint32_t* models) {
    use_device();
    if (!comm) throw HipError(SALVA_HIP_E_INVALID, "get_owned is for multi-GPU worlds (set_domain)");
    if (!dist_started || !sorted_valid) throw HipError(SALVA_HIP_E_INVALID, "no step has run yet");
    DevBuf<uint32_t> dg, dm;
    DevBuf<float> dp, dvv;
    DevBuf<unsigned int> cnt;
    dg.ensure(std::max(cap, 1u)); dm.ensure(std::max(cap, 1u)); dp.ensure(3 * (size_t)std::max(cap, 1u)); dvv.ensure(3 * (size_t)std::max(cap, 1u));
    cnt.ensure(1);
    SALVA_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(unsigned int), stream));
    if (n) k_pack_owned<<<div_up(n, BLOCK), BLOCK, 0, stream>>>(n, posm[cur].p, vel[cur].p, gtag[cur].p, perm[cur].p, model[cur].p, cnt.p,
                                                               cap, dg.p, dp.p, dvv.p, dm.p);
    unsigned int h = 0;
    SALVA_HIP_CHECK(hipMemcpyAsync(&h, cnt.p, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    SALVA_HIP_CHECK(hipStreamSynchronize(stream));
    const uint32_t m = std::min<uint32_t>(h, cap);
    if (m) {
        if (gids) SALVA_HIP_CHECK(hipMemcpy(gids, dg.p, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (models) SALVA_HIP_CHECK(hipMemcpy(models, dm.p, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (pos) SALVA_HIP_CHECK(hipMemcpy(pos, dp.p, 3 * (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
        if (vel_out) SALVA_HIP_CHECK(hipMemcpy(vel_out, dvv.p, 3 * (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
    }
    return h;
}

}  // namespace salva
