// sched.hip — bank-conflict-aware ordering of the neighbour lists.  KERNEL-DEVELOPMENT BUILD ONLY (`make VARIANT=diag`,
// SALVA_HIP_SCHED=1): correct (the GPU parity, fuzz and decomposition suites pass with it on) and it does what the model says — but
// the pair loops are only ~40 % of a neighbour kernel (the rest are per-tile phases), so a pass gains 6 % while this
// kernel, latency-bound in its LDS hand-shakes, costs ~600 us per step at 10^6 particles (profiles/r03_experiments/r03c_ab_sched.log).
//
// Why.  The pair loops of the neighbour-sum kernels are bound by the CU's LDS pipe, not by VALU issue (round 3: cutting the
// loop of k_pred_density from 24 to 16 VALU per contact moved the kernel from 55.9 to 52.6 us; the per-tile phase stamps show
// the compute phase at ~12.5 k cycles with SQ_LDS_IDX_ACTIVE ~ all of it).  A contact costs one or two ds_read_b128 at a
// data-dependent slot.  The hardware serves a wave's ds_read_b128 in four fixed groups of 16 lanes
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32: MI355X_MICROARCH.md §LDS), one cycle per group when the 16 lanes hit 16
// different bank quads (slot mod 16) or the same address, and one more cycle for every further distinct address on a busy quad.
// With lists in build order (ascending slot) the 16 slots of a group are as good as random: 2.2-2.5 cycles per group instead of 1
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.54; simulated on a jittered lattice: 2.5).
//
// What.  The ORDER of a particle's list is free (the reference's is a hash-iteration order, contacts.rs:222; sums are compared
// at 1e-5).  This pass permutes the first nff[i] entries of every list — padding entries stay where they are — so that at every
// list position k the 16 lanes of a group read distinct quads or a common slot:
//   * each lane buckets its entries by quad (slot mod 16);
//   * per position k, per group, up to ROUNDS times: every unassigned lane proposes the first quad, in an order rotated by
//     (cell rank of the lane + k), in which it still has entries and which it has not tried; if nobody owns that quad yet the
//     proposals race through an LDS atomicMin on (entries left, lane) — the lane with more left wins, takes its first entry of
//     the quad and publishes the slot; a lane whose quad is owned takes the SAME slot if it holds it (a broadcast is free);
//   * lanes still unassigned take the first entry they have (a conflict).
// Simulated (tools/lds_conflict_sim.py, same lattice): 1.4 cycles per group with ROUNDS = 3 against 2.5 in build order.
// Deterministic: the outcome of an atomicMin does not depend on arrival order.  Lists longer than SCHED_MAXE entries, or with
// more than 15 entries in one quad, keep their build order.
//
// One wave per 64-particle slice, launched per tile like every tile kernel; no staging.
#include "../kernels.h"
#include "../tile.h"

namespace salva {

constexpr uint32_t SCHED_MAXE = 64;     // entries per list the LDS scratch holds
constexpr int SCHED_ROUNDS = 3;
constexpr uint32_t SCHED_WAVE_BYTES = SCHED_MAXE * WAVE * 2u + 4u * 16u * 4u * 2u + 4u * 4u;  // entries + claim/pub tables + owned masks

__device__ __forceinline__ uint32_t nib(unsigned long long v, uint32_t c) { return (uint32_t)(v >> (4u * c)) & 15u; }
__device__ __forceinline__ uint32_t byte_of(unsigned long long lo, unsigned long long hi, uint32_t c) {
    return (uint32_t)((c < 8u ? lo : hi) >> (8u * (c & 7u))) & 255u;
}

__global__ __launch_bounds__(TILE_MAX_THREADS) void k_list_schedule(StepCtx c) {
    Tile t;
    t.setup(c);
    if (t.empty()) return;
    const uint32_t wv = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    unsigned char* mine = tile_smem + (size_t)wv * SCHED_WAVE_BYTES;
    uint16_t* ent = reinterpret_cast<uint16_t*>(mine);                                   // entry e of lane l: ent[e * 64 + l]
    // (volatile: other lanes of the wave write these between two reads of one lane)
    volatile uint32_t* claim = reinterpret_cast<volatile uint32_t*>(mine + SCHED_MAXE * WAVE * 2u);  // [group][quad]: smallest key proposed
    volatile uint32_t* pub = claim + 64;                                                              // [group][quad]: slot the owner reads
    volatile uint32_t* owned = pub + 64;                                                              // [group]: quads owned at this position
    // ds_read_b128 lane groups and the lane's rank in its group
    const uint32_t l5 = lane & 31u;
    const bool in_a = (l5 < 4u) || (l5 >= 12u && l5 < 16u) || (l5 >= 20u && l5 < 28u);
    const uint32_t grp = (in_a ? 0u : 1u) + 2u * (lane >> 5);
    const uint32_t rank = in_a ? (l5 < 4u ? l5 : (l5 < 16u ? l5 - 8u : l5 - 12u)) : (l5 < 12u ? l5 - 4u : (l5 < 20u ? l5 - 8u : l5 - 16u));
    const uint32_t rho = rank & ~3u;  // lanes of one cell (four consecutive ranks on the 2r lattice) start from the same quad: they share most slots
    claim[lane] = 0xffffffffu;
    if (lane < 4u) owned[lane] = 0u;
    t.for_own([&](uint32_t i, uint32_t gs, bool active) {
        const uint32_t cnt = active ? c.nff[i] : 0u;
        const uint32_t K = wave_max_u32(cnt);
        uint32_t* __restrict__ p = c.nbr_ff + (size_t)gs * c.cap_ff * WAVE + 4u * lane;
        // ---- pass 1: entries per quad
        unsigned long long cnt64 = 0ull;
        bool too_many = false;
        const uint32_t nqmax = (K + 1u) >> 1;
        for (uint32_t q = 0; q < nqmax; ++q) {
            if (2u * q < cnt) {
                const uint32_t d = p[ellq(q)];
                const uint32_t c0 = d & 15u, c1 = (d >> 16) & 15u;
                if (nib(cnt64, c0) == 15u) too_many = true;
                cnt64 += 1ull << (4u * c0);
                if (2u * q + 1u < cnt) {
                    if (nib(cnt64, c1) == 15u) too_many = true;
                    cnt64 += 1ull << (4u * c1);
                }
            }
        }
        if (K > SCHED_MAXE || K < 2u || __builtin_amdgcn_ballot_w64(too_many) != 0ull) return;  // (wave-uniform)
        // first position of each quad's bucket: 8-bit fields, quads 0-7 in `s_lo`, 8-15 in `s_hi`
        unsigned long long s_lo = 0ull, s_hi = 0ull;
        {
            uint32_t run = 0;
#pragma unroll
            for (uint32_t q4 = 0; q4 < 16u; ++q4) {
                if (q4 < 8u) s_lo |= (unsigned long long)run << (8u * q4);
                else s_hi |= (unsigned long long)run << (8u * (q4 - 8u));
                run += nib(cnt64, q4);
            }
        }
        // ---- pass 2: bucket the entries (ascending slot inside a bucket, as in the list)
        uint32_t self_pad = 0;
        {
            unsigned long long fill = 0ull;
            for (uint32_t q = 0; q < nqmax; ++q) {
                if (2u * q < cnt) {
                    const uint32_t d = p[ellq(q)];
                    const uint32_t x0 = d & 0xffffu, x1 = d >> 16;
                    const uint32_t c0 = x0 & 15u;
                    ent[(byte_of(s_lo, s_hi, c0) + nib(fill, c0)) * WAVE + lane] = (uint16_t)x0;
                    fill += 1ull << (4u * c0);
                    if (2u * q + 1u < cnt) {
                        const uint32_t c1 = x1 & 15u;
                        ent[(byte_of(s_lo, s_hi, c1) + nib(fill, c1)) * WAVE + lane] = (uint16_t)x1;
                        fill += 1ull << (4u * c1);
                    } else {
                        self_pad = x1;  // an odd list ends with the particle's own slot (k_nbr_tile)
                    }
                }
            }
        }
        // ---- the schedule, position by position
        unsigned long long used = 0ull;
        uint32_t avail = 0u;
#pragma unroll
        for (uint32_t q4 = 0; q4 < 16u; ++q4) avail |= (nib(cnt64, q4) ? 1u : 0u) << q4;
        uint32_t rem = cnt, lo_half = 0u;
        for (uint32_t k = 0; k < K; ++k) {
            const bool live = rem != 0u;
            const uint32_t start = (rho + k) & 15u;
            uint32_t tried = 0u;
            bool assigned = false;
            uint32_t my_c = 0u, my_pos = 0u;  // quad and bucket position of the entry this lane reads at position k
            const uint32_t key = ((63u - min(rem, 63u)) << 6) | lane;
#pragma unroll 1
            for (int rnd = 0; rnd < SCHED_ROUNDS; ++rnd) {
                const uint32_t own_mask = owned[grp];
                bool proposed = false, joiner = false;
                uint32_t cq = 0u;
                if (live && !assigned) {
                    const uint32_t cand = avail & ~tried;
                    if (cand) {
                        const uint32_t rot = ((cand >> start) | (cand << (16u - start))) & 0xffffu;
                        cq = (start + (uint32_t)__builtin_ctz(rot)) & 15u;
                        tried |= 1u << cq;
                        if (own_mask & (1u << cq)) joiner = true;
                        else { proposed = true; atomicMin(const_cast<uint32_t*>(&claim[grp * 16u + cq]), key); }
                    }
                }
                bool won = false;
                if (proposed) {
                    won = claim[grp * 16u + cq] == key;
                    if (won) {
                        my_c = cq; my_pos = byte_of(s_lo, s_hi, cq) + nib(used, cq);
                        pub[grp * 16u + cq] = ent[my_pos * WAVE + lane];
                        atomicOr(const_cast<uint32_t*>(&owned[grp]), 1u << cq);
                        assigned = true;
                    } else {
                        joiner = true;  // lost the race: the winner's slot may be one of mine
                    }
                }
                if (joiner) {  // the quad has an owner: read the same slot if this lane holds it (one LDS cycle serves both)
                    const uint32_t x = pub[grp * 16u + cq];
                    const uint32_t b = byte_of(s_lo, s_hi, cq), e0 = b + nib(used, cq), e1 = b + nib(cnt64, cq);
                    for (uint32_t e = e0; e < e1; ++e) {
                        const uint32_t y = ent[e * WAVE + lane];
                        if (y == x) {  // bring it to the front of the unread part of the bucket
                            ent[e * WAVE + lane] = ent[e0 * WAVE + lane];
                            ent[e0 * WAVE + lane] = (uint16_t)x;
                            my_c = cq; my_pos = e0; assigned = true;
                            break;
                        }
                    }
                }
            }
            if (live && !assigned) {  // no free quad found: read the first entry in rotated order (a conflict)
                const uint32_t rot = ((avail >> start) | (avail << (16u - start))) & 0xffffu;
                my_c = (start + (uint32_t)__builtin_ctz(rot)) & 15u;
                my_pos = byte_of(s_lo, s_hi, my_c) + nib(used, my_c);
            }
            // tables back to idle for the next position (every owner clears its own claim; the owned masks are cleared by all)
            if (live && (owned[grp] & (1u << my_c))) claim[grp * 16u + my_c] = 0xffffffffu;
            owned[grp] = 0u;
            if (live) {
                const uint32_t x = ent[my_pos * WAVE + lane];
                used += 1ull << (4u * my_c);
                if (nib(used, my_c) == nib(cnt64, my_c)) avail &= ~(1u << my_c);
                --rem;
                if (k & 1u) p[ellq(k >> 1)] = lo_half | (x << 16);
                else lo_half = x;
            }
        }
        if (cnt & 1u) p[ellq(cnt >> 1)] = lo_half | (self_pad << 16);
    });
}

uint32_t list_schedule_lds_bytes(const TileLds& L) { return (L.threads / WAVE) * SCHED_WAVE_BYTES; }
void launch_list_schedule(const StepCtx& c, const TileLds& L, hipStream_t s) {
    SALVA_LAUNCH_TILE(k_list_schedule, c, L, list_schedule_lds_bytes(L), s, c);
}

}  // namespace salva
