// dcs.h — DynamicContactSampling (dcs.hip): launcher interface used by world.hip.
#pragma once
#include "common.h"
#include "device_types.h"
#include "../../include/salva_hip.h"

namespace salva {

struct DcsParams {
    int kind;
    float p[3];           // radius | half extents | (half height, radius) for capsule and cylinder
    float t[3], q[4];     // collider.position(): translation, unit quaternion (i, j, k, w)
    float lo[3], hi[3];   // shape AABB loosened by h + prediction (fluids_pipeline.rs:196-199)
    int clo[3], chi[3];   // HGrid::key(mins) .. key(maxs) (hgrid.rs:128-129)
    float dt;             // timestep.dt(): the previous substep's length (0 before the first step)
    float margin;         // particle_radius * 0.1 (:194)
    float reach;          // h + prediction (:234)
    float eps;            // f32::default_epsilon() (:223)
    float h;              // cell width (on a folded grid the cell a key names is one of several images: the position says which)
};

// number of parameters of a built-in collider shape (include/salva_hip.h); throws SALVA_HIP_E_INVALID for any other kind
int shape_param_count(int kind);
void shape_world_extent(const SalvaHipShape& shape, const float rotation_ijkw[4], float ext[3]);
DcsParams dcs_params(const SalvaHipShape& shape, const SalvaHipRigidPose& pose, float h, float particle_radius, float dt);
// one candidate (projection xyz, host particle index bits) and one flag per fluid particle; pushes particles inside the
// shape out of it in place (positions and velocities of the sorted working set).  gtag != nullptr (decomposed run): ghosts are
// pushed but emit nothing, and the rows carry the sorted index instead of perm[] (launch_dcs_pack)
void launch_dcs_project(uint32_t n, float4* posm, float4* vel, const uint32_t* keys, const uint32_t* perm, const uint32_t* gtag, TileGrid g,
                        const DcsParams& s, float4* cand, uint8_t* flag, hipStream_t st);
// host-shape arm (salva_hip_set_boundary_dynamic_sampling_host): box tests -> (predicted position, sorted index) per passing
// particle; then, from the host's (projection, is_inside != 0) per compacted candidate, the push-out and the emitted point
DcsParams dcs_params_host(const float mins[3], const float maxs[3], float h, float particle_radius, float dt);
void launch_dcs_gather(uint32_t n, const float4* posm, const float4* vel, const uint32_t* keys, TileGrid g, const DcsParams& s, float4* cand,
                       uint8_t* flag, hipStream_t st);
void launch_dcs_apply(uint32_t cnt, const float4* pred, const float4* proj, float4* posm, float4* vel, const uint32_t* perm,
                      const uint32_t* gtag, const DcsParams& s, float4* cand, uint8_t* flag, hipStream_t st);
// decomposed run: compacted rows (point, sorted index) -> (point, global id of the source particle), fluid of the source
void launch_dcs_pack(uint32_t cnt, const float4* rows, const uint32_t* gid, const uint32_t* model, float4* out_rows, uint32_t* out_models,
                     hipStream_t st);
// compacted candidates -> boundary rows (position, volume 0), (velocity at the point, boundary slot), source particle
void launch_dcs_emit(uint32_t cnt, const float4* cand, const SalvaHipRigidPose& pose, uint32_t slot, float4* pos, float4* vel,
                     uint32_t* src, hipStream_t st);

}  // namespace salva
