// dev_common.h -- structures shared between the C-ABI host code and the gfx950 kernels.
#pragma once
#include <hip/hip_ext.h>
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
    int32_t grab_local2; // second pinned particle (TETSIM_FLAG_REF_GRAB_TEXEL can select two), -1 = none
    uint32_t epoch;      // persistent frame kernel (pj_blocked.hip): sequence number of this call's first substep (the host advances it by 65536 per parameter push: tetsim_api.hip push_params)
    int32_t poll_delay;  // pj_quad.hip frame kernel: s_sleep units (64 clocks each) between a tile's partial-sum store and its first look at the neighbours'
    int32_t pad3[1];
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
    float rot_exit_w2 = 1.0e-18f; // FAST: squared |omega| that ends a tet's correction iterations (pj_math.inc)
};

// ---- POLAR_JACOBI, blocked formulation (FAST mode) -----------------------
// Tets are tiled into workgroups of <= 256 tets touching <= 256 distinct particles.  The tile's particle
// positions are staged in LDS, the 4 goals of every tet are reduced IN LDS to one partial sum per tile
// particle (fixed order: deterministic), and the per-particle pass adds the few partial sums of the tiles
// that touch it.
struct PJBlk {
    uint32_t nb = 0, nb_interior = 0, nt = 0, nv_local = 0, nv_owned = 0, nv_boundary = 0;  // tiles >= nb_interior touch ghosts
    const uint32_t* blk_tet_off = nullptr;   // [nb+1]
    const uint32_t* blk_vert_off = nullptr;  // [nb+1]
    const int32_t* blk_verts = nullptr;      // particle id of every tile slot
    const uchar4* tet_lidx = nullptr;        // [nt] LDS slot of the 4 corners
    float4* rest_a = nullptr;                // [nt] carried rest corners, 12 floats packed in 3 x 16 B:
    float4* rest_b = nullptr;                //      a = r0.xyz r1.x | b = r1.yz r2.xy | c = r2.z r3.xyz
    float4* rest_c = nullptr;
    float* rest_c1 = nullptr;                // TETSIM_FLAG_LEAN_STATE: r2.z -- the record is r0 r1 r2 (a, b, c1: 36 B), r3 = -(r0 + r1 + r2), rest_c unused
    const float* vol = nullptr;              // [nt] rest volume (the averaging weight)
    float4* quat = nullptr;                  // [nt]
    const uint32_t* lc_range = nullptr;      // per tile slot: first | end << 16 into the tile's entry list
    const uint2* lc_ent = nullptr;           // [nt] 4 x u16 per tet position: (tetLocal*4 + corner), grouped by slot
    float4* partial = nullptr;               // per tile slot: sum V*goal over the slot's entries (w unused)
    const float* wsum = nullptr;             // per owned particle: sum of V over all its entries -- constant, summed on the host in the device's order
    const uint32_t* vp_ell = nullptr;        // ELL [vp_cols][nv_pad]: partial-sum indices of each owned particle,
    uint32_t vp_cols = 0, nv_pad = 0;        //   ascending tile, 0xffffffff = none
    float4* pos_pred = nullptr;
    float4* pos_final = nullptr;
    float4* vel = nullptr;
    // Fused particle pass (pjb_tet_kernel_x<.., TetFused>): the staging of substep s+1 performs the particle update of substep s for the
    // tile's own particles instead of reading predictions a separate kernel wrote.  Inputs of that update are the PREVIOUS tet
    // pass's partial sums and end-of-substep positions, which other tiles of the same launch still read while this launch writes
    // the new ones: both are double buffered (the launcher fills these four per launch; the particle kernel uses fin_in / fin_out
    // too -- a lane only touches its own particle there, so they may be the same array).
    const float4* partial_prev = nullptr;    // fused staging input: the partial sums of the previous tet pass
    const float4* fin_in = nullptr;          // end-of-substep positions to read (prevPos of the update)
    float4* fin_out = nullptr;               // ... and to write
    const uint32_t* slot_src = nullptr;      // ELL [vp_cols][ns_pad]: per tile slot, the partial sums of its particle
    const uint32_t* blk_maxsrc = nullptr;    // [nb] longest such list in the tile
    uint32_t ns_pad = 0;
    const DevParams* params = nullptr;
    // peer-to-peer halo (tetsim_p2p.hip, tetsim_halo.hip): the neighbours store their boundary predictions straight into this rank's ghost range,
    // double buffered by substep parity -- pos_pred's own tail on even substeps, ghost_alt on odd ones (pjb_tet_kernel_x<.., TetAlt>)
    const float4* ghost_alt = nullptr;       // [nv_local - nv_owned]
    // two-layer ghost regions: ghosts [nv_owned, nv_owned + n_ghost1) come from ghost_alt, the second layer from ghost2
    const float4* ghost2 = nullptr;
    uint32_t n_ghost1 = 0xffffffffu;
    bool lean = false;                       // TETSIM_FLAG_CONSTANT_REST_SHAPE: rest_a/b/c hold the centred rest shape, read-only
    bool lean_state = false;                 // TETSIM_FLAG_LEAN_STATE: three carried corners, no quaternion in the substep (pj_blocked.hip: kModeLeanState)
    float rot_exit_w2 = 1.0e-18f;            // squared |omega| that ends a tet's correction iterations 2..9 (pj_math.inc; 1e-18 = the reference's 1e-9)
    uint32_t epoch = 0;                      // persistent frame kernels: first sequence number of this launch; 0 = DevParams::epoch (graph launches: tetsim_step_n)
    unsigned long long* trace = nullptr;     // development: 8 x u64 per tile (phase timestamps), TETSIM_DEBUG_TRACE
    unsigned long long* iter_hist = nullptr; // development (ablation build): rotation-iteration statistics, TETSIM_DEBUG_ITER_HIST (pj_blocked.hip: pjb_log_iterations)
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
    uint32_t* corner_slots = nullptr;  // [nt] TETSIM_ORDER_CLUSTERED: the corners' cluster-local vertex slots, a byte each
    const DevParams* params = nullptr;
};
// One launch of the clustered Gauss-Seidel schedule (host_prep.h ClusterPlan): lane = cluster, step j = tets first[j] + lane
// for lanes < count[j] (count non-increasing), slot_vid = [kNHClusterVerts][clusters] vertex ids (-1 = unused).
constexpr uint32_t kNHClusterVerts = 8, kNHClusterTets = 8;
struct NHClusterLaunch {
    uint32_t nsteps = 0, clusters = 0;
    uint32_t first[kNHClusterTets] = {}, count[kNHClusterTets] = {};
    const int32_t* slot_vid = nullptr;
    // folded particle pass (tetsim_step_n, substeps after the first of a run): bit k of first_mask[cluster] = this cluster is the
    // FIRST of the sweep to touch the particle in its slot k -- its lane finishes the previous substep for that particle (bounds,
    // floor, grab, velocity: Softbody.js:213-239) and predicts this one (:198-202) as it loads it
    const uint8_t* first_mask = nullptr;
};

// The clustered sweep as ONE launch per substep (FAST, four lanes per cluster; nh_kernels.inc: nh_sweep1_kernel): the grid holds the
// clusters of ALL colours, colour by colour -- the dispatcher hands out workgroups in grid order, so whatever a waiting wave waits for has
// been dispatched before it -- and a particle travels from a cluster to the next one that touches it as ONE 16-byte store {x, y, z, stamp}
// into an exchange array, stamp = epoch + substep * colours + colour of the writer + 1 (data and "it is there" in one store, as the polar
// frame kernel's partial sums: pj_blocked.hip).  A cluster polls only its own slots, for the stamp the host worked out per slot (`delta`:
// how many colours back the previous toucher sits; 0 = this cluster is the first of the sweep to touch the particle, which comes from
// NHDev::pos as before -- the previous launch completed it); the sweep's LAST toucher of a particle stores it to NHDev::pos.  Inverse
// masses are constants and travel in the cluster record (slot_im), the fourth float of an exchanged particle being the stamp.
struct NHSweepColour {
    NHClusterLaunch L;
    const float* slot_im = nullptr;      // [kNHClusterVerts][clusters]
    const uint2* delta = nullptr;        // [clusters] a byte per slot
    const uint8_t* last_mask = nullptr;  // [clusters] bit k: no later cluster of the sweep touches the particle in slot k
    const uint2* delta_first = nullptr;  // [clusters] a byte per FIRST-touch slot: colours back to the particle's last toucher of the PREVIOUS sweep (nh_call_kernel)
    uint32_t first_block = 0;            // of this colour in the one-launch grid (4 waves of 16 clusters per block)
    uint32_t pad = 0;
};
struct NHSweep {
    const NHSweepColour* colours = nullptr;   // device, [ncolours]
    uint32_t ncolours = 0, blocks = 0;
    float4* exchange = nullptr;               // [nv]
    uint32_t* error = nullptr;                // raised by a wave whose wait gave up (bounded: timeout_ms)
    uint32_t timeout_ms = 0;
};
// sub_index: substep inside the call (stamps of different substeps never collide); epoch: first stamp of the call, 0 = DevParams::epoch
void nh_launch_sweep1_fast(hipStream_t s, const NHDev& d, const NHSweep& w, bool fold, uint32_t sub_index, uint32_t epoch);
// the sweeps of ALL n substeps of a call in one launch (nh_kernels.inc: nh_call_kernel; bodies with NHSweepColour::delta_first): in front of it
// the prediction kernel, behind it the kernel that ends the call; stamps start at DevParams::epoch
void nh_launch_call_fast(hipStream_t s, const NHDev& d, const NHSweep& w, uint32_t n);

// launchers (one set per arithmetic mode; defined in pj_precise.hip / pj_fast.hip / nh_*.hip)
// e0/e1 (optional): HIP events that timestamp the kernel's own begin and end (hipExtLaunchKernelGGL), for
// tetsim_profile; normal launches pass none.
void pj_launch_tet_precise(hipStream_t s, const PJDev& d, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pj_launch_tet_fast(hipStream_t s, const PJDev& d, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pj_launch_vertex_precise(hipStream_t s, const PJDev& d, uint32_t first, uint32_t count, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pj_launch_vertex_fast(hipStream_t s, const PJDev& d, uint32_t first, uint32_t count, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pj_launch_repredict_precise(hipStream_t s, const PJDev& d);
void pj_launch_repredict_fast(hipStream_t s, const PJDev& d);

void nh_launch_post_predict_precise(hipStream_t s, const NHDev& d);
void nh_launch_post_predict_fast(hipStream_t s, const NHDev& d);
// fold: this sweep's first touchers do the particle pass between the previous substep and this one (L.first_mask)
void nh_launch_cluster_precise(hipStream_t s, const NHDev& d, const NHClusterLaunch& L, bool fold = false);
void nh_launch_cluster_fast(hipStream_t s, const NHDev& d, const NHClusterLaunch& L, bool fold = false);
// the same pass for the particles no cluster touches (list of particle ids), when the sweeps fold the rest
// small bodies: a whole call as ONE workgroup per body with every particle of the body in LDS (nh_kernels.inc: nh_frame_kernel)
struct NHFrameLaunch {
    const uint32_t* seg = nullptr;         // [levels][bodies][2]: per level and body, first and end solve position of the body's tets (8-byte aligned pairs)
    const uint32_t* first_vert = nullptr;  // [bodies + 1]
    uint32_t levels = 0, bodies = 0, block = 512;   // threads per workgroup, <= 512
    uint32_t max_body_particles = 0;       // x 40 bytes of LDS per workgroup
};
void nh_launch_frame_precise(hipStream_t s, const NHDev& d, const NHFrameLaunch& f, uint32_t n, const DevParams& params, DevParams* params_dev);   // (parameters by value: pjb_launch_frame)
void nh_launch_frame_fast(hipStream_t s, const NHDev& d, const NHFrameLaunch& f, uint32_t n, const DevParams& params, DevParams* params_dev);
// levels of at most this many tets are solved on four lanes per tet by the single-workgroup launch of small bodies (and, FAST, by its
// stepwise twin): 128 quads = two waves per SIMD in f32; f64 (PRECISE) runs at half rate, one wave per SIMD
constexpr uint32_t kNHQuadLevelFast = 128, kNHQuadLevelPrecise = 64;
void nh_launch_level4_fast(hipStream_t s, const NHDev& d, uint32_t first, uint32_t count);   // one tet per quad: FAST stepwise twin of the frame kernel's sweep; small levels
void nh_launch_level4_precise(hipStream_t s, const NHDev& d, uint32_t first, uint32_t count);
uint32_t nh_frame_lds_limit_precise();
uint32_t nh_frame_lds_limit_fast();
void nh_launch_post_predict_list_precise(hipStream_t s, const NHDev& d, const uint32_t* list, uint32_t n);
void nh_launch_post_predict_list_fast(hipStream_t s, const NHDev& d, const uint32_t* list, uint32_t n);
// raise_word != nullptr: the kernel sets that hand-over word (PJSync::flag) as it STARTS, i.e. "everything in front of this kernel
// in its queue is done" -- a signal kernel folded into its successor
// clear_word != nullptr: ... and puts that word back to 0 first (a word whose waiters were the waves of the kernel in front: pjb_launch_vertex_await)
void pjb_launch_tet(hipStream_t s, const PJBlk& d, uint32_t tile_first, uint32_t tile_count, hipEvent_t e0 = nullptr,
                    hipEvent_t e1 = nullptr, uint32_t* raise_word = nullptr, uint32_t* clear_word = nullptr);
// n substeps -- per substep every tile, then every owned particle -- as ONE launch (pj_blocked.hip: pjb_call_kernel): partial sums and
// predictions carry the sequence number (d.epoch or DevParams::epoch) + substep + 1, their readers look for it; err: raised by a wave that
// waited in vain.  n * blocks per substep must fit a grid (tetsim_step_n chunks long calls).
void pjb_launch_call(hipStream_t s, const PJBlk& d, uint32_t n, uint32_t* err, uint32_t timeout_ms, const DevParams& params, DevParams* params_dev);
// the same tiles with the previous substep's particle update fused into the staging (d.partial_prev / fin_in / fin_out set)
void pjb_launch_tet_fused(hipStream_t s, const PJBlk& d, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pjb_launch_vertex(hipStream_t s, const PJBlk& d, uint32_t first, uint32_t count, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                       uint32_t* raise_word = nullptr, uint32_t* clear_word = nullptr);
void pjb_launch_repredict(hipStream_t s, const PJBlk& d);
// TETSIM_FLAG_LEAN_STATE: d.quat <- the rotation between the constant centred rest shape (rest0_*) and the carried shape, next to what d.quat held
void pjb_launch_recover_quats(hipStream_t s, const PJBlk& d, const float4* rest0_a, const float4* rest0_b, const float4* rest0_c);
// n substeps of an unpartitioned fused-eligible body in ONE persistent launch (pjb_frame_kernel): every tile's workgroup stays
// resident for the whole call.  block_tile[blocks]: the tile each block works on, -1 = none (the host places the tiles of a body
// on one XCD that way); local: the exchange of partial sums only has to be coherent inside one XCD's L2 (valid with such a
// placement, see pjb_probe_xcd).  pbuf: the two partial-sum buffers; err: a device word raised if a neighbour tile's partial sums
// did not appear within timeout_ms (never in a correct run: all workgroups are co-resident -- pjb_frame_capacity).
// params / params_dev: the call's parameters travel with the launch (by value); its first workgroup leaves them in device memory.
void pjb_launch_frame(hipStream_t s, const PJBlk& d, uint32_t n, const int32_t* block_tile, uint32_t blocks, bool local, float4* pbuf0, float4* pbuf1,
                      uint32_t* err, uint32_t timeout_ms, const DevParams& params, DevParams* params_dev, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// ... and the same three entry points for SMALL bodies tiled into 64-tet tiles, one tet / one particle on four lanes (pj_quad.hip):
// the persistent frame kernel, and its substep as two launches through memory (tetsim_step, tetsim_profile, fallback).  All three
// agree bit for bit.  d.partial: the split kernels' partial sums; pbuf0 / pbuf1: the frame kernel's (sequence-numbered).
void pjq_launch_frame(hipStream_t s, const PJBlk& d, uint32_t n, const int32_t* block_tile, uint32_t blocks, bool local, float4* pbuf0, float4* pbuf1,
                      uint32_t* err, uint32_t timeout_ms, const DevParams& params, DevParams* params_dev, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pjq_launch_tet(hipStream_t s, const PJBlk& d, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pjq_launch_vertex(hipStream_t s, const PJBlk& d, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
uint32_t pjq_frame_capacity(uint32_t* compute_units);              // workgroups of pjq_frame_kernel one CU keeps resident (0 = query failed)
uint32_t pjb_frame_capacity(int mode, uint32_t* compute_units);    // workgroups of the frame kernel one CU keeps resident (0 = query failed); mode: pjb_mode()
// waves of the self-waiting particle kernel / workgroups of the self-waiting halo-side tiles that may wait at once on the current device
void pjb_wait_capacity(int mode, uint32_t* vertex_waves, uint32_t* hwait_blocks);
inline int pjb_mode(const PJBlk& d) { return d.lean ? 1 : d.lean_state ? 2 : 0; }   // 0 carried shape + quaternion, 1 constant rest shape, 2 lean state
uint32_t pjb_probe_xcd(hipStream_t s, uint32_t blocks);            // 8 if block i of a grid runs on XCD i % 8, else 0
// Cross-queue hand-over of partitioned bodies (pj_blocked.hip): `flag` is a binary semaphore in device memory -- signal stores
// 1 behind the producer kernel, wait spins until it is non-zero in front of the consumer kernel and clears it.  No per-launch
// argument changes: the kernels can be replayed from a captured graph.
struct PJSync {
    uint32_t* flag = nullptr;
    uint32_t* error = nullptr;
    uint32_t timeout_ms = 0;   // 0 = unbounded
};
// Peer-to-peer halo: what the boundary-particle kernel needs to store a boundary particle's new prediction straight into the
// ghost ranges of the (<= kMaxPeers) neighbours that read it.  slots: ELL [cols][stride] per boundary particle, entry =
// neighbour index << 24 | position in that neighbour's ghost run for this rank, 0xffffffff = none.
constexpr uint32_t kMaxPeers = 8;
struct PJPeer {
    float4* ghost[kMaxPeers] = {};           // the neighbours' ghost runs for this rank, of the parity being written (peer memory)
    const uint32_t* slots = nullptr;
    uint32_t cols = 0, stride = 0, n = 0;   // n: neighbours in use
    // two-layer ghost regions: `slots` lists the particles that are a neighbour's FIRST layer (prediction -> ghost[k], and their
    // end-of-substep position -> fin[k]: the neighbour advances them itself and needs both), slots2 the ones that are its SECOND
    // layer (prediction -> ghost2[k]).  A null destination = nothing goes there in this substep.
    float4* fin[kMaxPeers] = {};
    float4* ghost2[kMaxPeers] = {};
    const uint32_t* slots2 = nullptr;
    uint32_t cols2 = 0;
};
// ... and the hand-over around it: the wait kernel in front of the halo-side tiles raises `raise` words in the NEIGHBOURS' memory as
// it starts ("my boundary predictions of the previous substep are in your ghost range": the kernel in front of it in the queue, the
// boundary-particle kernel, is complete) and then waits for the words the neighbours raise HERE, clearing them.
struct PJPeerSync {
    uint32_t* raise[kMaxPeers] = {};         // peer memory
    uint32_t* wait[kMaxPeers] = {};          // own memory
    uint32_t n_raise = 0, n_wait = 0;
    uint32_t delay_us = 0;                   // loopback measurements: pretend the data arrived this much later
};
void pjb_launch_wait(hipStream_t s, const PJSync& y);     // one wave: await + clear
void pjb_launch_wait_peers(hipStream_t s, const PJSync& y, const PJPeerSync& w);   // y.flag may be null (nothing local to wait for)
struct PJClear { uint32_t* word[kMaxPeers + 1] = {}; uint32_t n = 0; };   // words a kernel puts back to 0 as it starts
void pjb_launch_vertex_peer(hipStream_t s, const PJBlk& d, uint32_t first, uint32_t count, const PJPeer& peer, uint32_t* raise_word, const PJClear& clr = PJClear());
// the halo-side tiles with the halo queue's hand-overs inside (raise w.raise at the start, await yv.flag and w.wait without clearing them;
// positions from the memory side, ghosts from `ghosts`)
void pjb_launch_tet_hwait(hipStream_t s, const PJBlk& d, uint32_t tile_first, uint32_t tile_count, const PJSync& yv, const PJPeerSync& w, const float4* ghosts);
void pjb_launch_tet_alt(hipStream_t s, const PJBlk& d, uint32_t tile_first, uint32_t tile_count);   // ghosts from d.ghost_alt
void pjb_launch_signal(hipStream_t s, const PJSync& y, uint32_t* clear_word = nullptr);   // one wave: [clear another word,] set
// the particle kernel whose every wave awaits y.flag itself, without clearing it (the queue's next kernel does: clear_word above)
void pjb_launch_vertex_await(hipStream_t s, const PJBlk& d, uint32_t first, uint32_t count, const PJSync& y, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

void nh_launch_predict_precise(hipStream_t s, const NHDev& d);
void nh_launch_predict_fast(hipStream_t s, const NHDev& d);
void nh_launch_predict_value_precise(hipStream_t s, const NHDev& d, const DevParams& params, DevParams* params_dev);   // (the prediction that brings a call's parameters along)
void nh_launch_predict_value_fast(hipStream_t s, const NHDev& d, const DevParams& params, DevParams* params_dev);
void nh_launch_level_precise(hipStream_t s, const NHDev& d, uint32_t first, uint32_t count);
void nh_launch_level_fast(hipStream_t s, const NHDev& d, uint32_t first, uint32_t count);
void nh_launch_post_precise(hipStream_t s, const NHDev& d);
void nh_launch_post_fast(hipStream_t s, const NHDev& d);

// embedded visual mesh skinning (skin_kernels.hip)
struct SkinDev {
    uint32_t nvis = 0;
    const int4* corner = nullptr;    // device particle index of the 4 tet corners of each visual vertex
    const float4* weight = nullptr;  // b0, b1, b2, (unused)
    const int32_t* qidx = nullptr;   // device tet position (quaternion index), polar only
    const float4* normal0 = nullptr; // rest normals (may be null)
    float4* out_pos = nullptr;
    float4* out_nrm = nullptr;
    // computeVertexNormals (tetsim_set_visual_triangles): triangles and, per visual vertex, its triangles in triangle order
    uint32_t ntri = 0;
    const int4* tri = nullptr;          // (a, b, c, 0)
    const uint32_t* vt_off = nullptr;   // [nvis + 1]
    const uint32_t* vt_tri = nullptr;   // triangle ids (a triangle listing a vertex twice appears twice)
    float4* out_vnrm = nullptr;
    // partitions: `tri` holds rows of the caller's GLOBAL visVerts and the positions come from tri_pos [global rows] (the ranks' skins put
    // together: tetsim_visual_vertex_normals_from); null = out_pos (an unpartitioned body's own skin)
    const float4* tri_pos = nullptr;
};
// js_order: Softbody.js:259-277 arithmetic (f64 accumulate, f32 store per step); else SoftbodyGPU.js:431-435 (f32)
void skin_launch(hipStream_t s, const SkinDev& d, const float4* pos, const float4* quat, bool js_order);
void skin_launch_vertex_normals(hipStream_t s, const SkinDev& d);   // three.js computeVertexNormals of d.out_pos -> d.out_vnrm

void util_launch_pack_xyz(hipStream_t s, const float4* src, const uint32_t* map, float* out, uint32_t n);
void util_launch_nearest(hipStream_t s, const float4* pos, const uint32_t* map, uint32_t n, double px, double py, double pz,
                         double* best_d2, uint32_t* best_id);
void util_launch_copy(hipStream_t s, const float4* src, float4* dst, uint64_t n);
// streaming probe (tetsim_measure_stream_bandwidth): kind 0 copy / 1 read only / 2 write only, nt = non-temporal accesses, unroll = 4 or 8
// independent 16-byte accesses per lane, grid = workgroups (0: one per chunk of 256 x unroll float4s)
void util_launch_stream(hipStream_t s, int kind, bool nt, uint32_t unroll, uint32_t grid, const float4* src, float4* dst, uint64_t n);
void util_launch_delay(hipStream_t s, uint32_t us);
// tetsim_halo_p2p_probe: raise[k] = this rank's inbox word at neighbour k (peer memory), wait[k] = neighbour k's inbox word here
struct P2PProbe { uint32_t* raise[kMaxPeers] = {}; uint32_t* wait[kMaxPeers] = {}; uint32_t n = 0; };
void util_launch_set_params(hipStream_t s, DevParams* dst, const DevParams& v);   // the call's parameters into device memory, in stream order
void util_launch_p2p_probe(hipStream_t s, const P2PProbe& p, uint32_t base, uint32_t reps, unsigned long long* ticks, uint32_t* error, uint32_t timeout_ms);   // loopback measurements: a stand-in for wire latency
void util_launch_gather4(hipStream_t s, const float4* src, const int32_t* idx, float4* dst, uint32_t n);


}  // namespace tetsim
