// dist.h — launcher interface of dist.hip (x-slab decomposition kernels).
#pragma once
#include "common.h"
#include "device_types.h"

namespace salva {

// per-particle tag carried through the cell sort in distributed runs
constexpr uint32_t GTAG_GHOST = 0x80000000u;      // copy of a neighbour's particle (read-only here)
constexpr uint32_t GTAG_BORDER_LO = 0x40000000u;  // owned particle mirrored on rank-1 / ghost received from rank-1
constexpr uint32_t GTAG_BORDER_HI = 0x20000000u;  // same towards rank+1
constexpr uint32_t GTAG_SLOT_MASK = 0x1fffffffu;  // position in the exchange buffer of that face

constexpr int GHOST_PLANES = 2;  // cell planes mirrored per face (see dist.hip)

struct DistRec {  // one particle on the wire (64 bytes)
    float4 posm, vel, dv;
    uint32_t model, gid, pad0, pad1;
};
struct DistArrays {
    float4 *posm, *vel, *dv;
    uint32_t *model, *gid, *gtag;
};

void launch_iota_u32(uint32_t n, uint32_t base, uint32_t* out, hipStream_t s);
size_t dist_scan_temp_bytes(uint32_t n);
size_t dist_sel_bytes(uint32_t n);
// per-block counts + their exclusive scan (the pack kernel ranks within blocks); totals_host = {keep, to_lo, to_hi}.  mode 1: migration, mode 2: ghost planes.  Synchronises.
void launch_dist_select(uint32_t n, const float4* posm, const uint32_t* gtag, float h, int lo, int hi, bool has_lo, bool has_hi,
                        int mode, int nbr_lo_lo, int nbr_hi_hi, void* sel, void* pos, void* temp, size_t temp_bytes, uint32_t* flags,
                        uint32_t totals_host[3], hipStream_t s);
void launch_plane_hist(uint32_t n, const float4* posm, const uint32_t* gtag, float h, int base, int len, unsigned long long* hist,
                       hipStream_t s);
void launch_dist_pack(uint32_t n, DistArrays in, DistArrays out, float h, int lo, int hi, bool has_lo, bool has_hi, int mode, int nbr_lo_lo,
                      int nbr_hi_hi, const void* pos, DistRec* send_lo, DistRec* send_hi, hipStream_t s);
void launch_dist_unpack(uint32_t count, uint32_t base, const DistRec* recv, DistArrays out, uint32_t tag_bits, hipStream_t s);
void launch_dist_lists(uint32_t n, const uint32_t* gtag, uint32_t* send_lo_idx, uint32_t* send_hi_idx, uint32_t* ghost_lo_idx,
                       uint32_t* ghost_hi_idx, hipStream_t s);
void launch_gather_f32(uint32_t count, const uint32_t* idx, const float* src, float* dst, hipStream_t s);
void launch_scatter_f32(uint32_t count, const uint32_t* idx, const float* src, float* dst, hipStream_t s);
void launch_gather_idx_f4(uint32_t count, const uint32_t* idx, const float4* src, float4* dst, hipStream_t s);
void launch_scatter_idx_f4(uint32_t count, const uint32_t* idx, const float4* src, float4* dst, hipStream_t s);

}  // namespace salva
