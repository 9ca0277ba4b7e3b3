// dev_common.h -- structures shared between the C-ABI host code and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tetsim {

// Per-step dynamic parameters, resident in device memory so that captured HIP graphs stay valid when
// the caller changes dt / physicsParams / grab between frames (kernels read it with scalar loads).
struct DevParams {
    // f32 view -- POLAR_JACOBI: the reference uploads these as f32 uniforms (SoftbodyGPU.js:614-637)
    float dt, gravity, friction, pad0;
    float lo[3], pad1;
    float hi[3], pad2;
    float grab[3];
    int32_t grab_local;  // local vertex index, -1 = none
    // f64 view -- NEOHOOKEAN_GS: JS numbers (Softbody.js:195-240)
    double d_dt, d_gravity, d_friction, d_dev_compliance, d_vol_compliance;
    double d_lo[3], d_hi[3];
};

// ---- POLAR_JACOBI device state (all arrays 16-byte elements: one dwordx4 per lane) ----------------
struct PJDev {
    uint32_t nv_local = 0, nv_owned = 0, nv_boundary = 0, nt = 0;
    uint32_t nt_pad = 0;      // plane stride of `elem`
    uint32_t nv_pad = 0;      // column stride of `slot_tab`
    uint32_t max_valence = 0;
    float4* pos_pred = nullptr;   // [nv_local] predicted positions x* = x + v dt (input of the tet kernel)
    float4* pos_final = nullptr;  // [nv_local] end-of-substep positions (== prevPos of the next substep)
    float4* vel = nullptr;        // [nv_local]
    int4* tet_idx = nullptr;      // [nt] local vertex ids
    float4* elem = nullptr;       // [4][nt_pad] xyz = last rotated rest corner / goal, w = rest volume
    float4* quat = nullptr;       // [nt]
    int32_t* slot_tab = nullptr;  // ELL [max_valence][nv_pad]: index into elem (corner*nt_pad + tet)
    uint32_t* slot_cnt = nullptr; // [nv_pad]
    const DevParams* params = nullptr;
};

// ---- NEOHOOKEAN_GS device state ---------------------------------------------------------------------
struct NHDev {
    uint32_t nv = 0, nt = 0;
    float4* pos = nullptr;     // xyz + invMass in w (one 16-byte gather per corner)
    float4* prev = nullptr;
    float4* vel = nullptr;
    int4* tet_idx = nullptr;   // [nt] in solve order
    float4* irp_a = nullptr;   // invRestPose m0..m3   (column-major, Softbody.js:359-361)
    float4* irp_b = nullptr;   // m4..m7
    float4* irp_c = nullptr;   // m8, invRestVolume, 0, 0
    double* vol_err = nullptr; // [nt] det F - 1 per tet, indexed by the CALLER's tet id
    int32_t* order = nullptr;  // [nt] solve position -> caller's tet id
    const DevParams* params = nullptr;
};

// launchers (one set per arithmetic mode; defined in pj_precise.hip / pj_fast.hip / nh_*.hip)
void pj_launch_tet_precise(hipStream_t s, const PJDev& d);
void pj_launch_tet_fast(hipStream_t s, const PJDev& d);
void pj_launch_vertex_precise(hipStream_t s, const PJDev& d, uint32_t first, uint32_t count);
void pj_launch_vertex_fast(hipStream_t s, const PJDev& d, uint32_t first, uint32_t count);
void pj_launch_repredict_precise(hipStream_t s, const PJDev& d);
void pj_launch_repredict_fast(hipStream_t s, const PJDev& d);

void nh_launch_predict_precise(hipStream_t s, const NHDev& d);
void nh_launch_predict_fast(hipStream_t s, const NHDev& d);
void nh_launch_level_precise(hipStream_t s, const NHDev& d, uint32_t first, uint32_t count);
void nh_launch_level_fast(hipStream_t s, const NHDev& d, uint32_t first, uint32_t count);
void nh_launch_post_precise(hipStream_t s, const NHDev& d);
void nh_launch_post_fast(hipStream_t s, const NHDev& d);

void util_launch_copy(hipStream_t s, const float4* src, float4* dst, uint64_t n);
void util_launch_gather4(hipStream_t s, const float4* src, const int32_t* idx, float4* dst, uint32_t n);

}  // namespace tetsim
