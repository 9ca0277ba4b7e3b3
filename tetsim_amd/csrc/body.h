// body.h -- the handle behind tetsim_handle and the helpers the translation units of the C ABI share.
//
//   tetsim_api.hip     lifecycle, stepping (streams / graphs)
//   tetsim_state.hip   state read-back (copying / pinned), checkpoint and resume, plan getters
//   tetsim_visual.hip  embedded visual mesh (skinning, vertex normals), grab (pin, nearest-particle query)
//   tetsim_measure.hip measurement: per-kernel profile, kernel timing loops, device copy bandwidth
//   tetsim_create.hip  construction of the two solvers' device state (host preprocessing -> uploads)
//   tetsim_halo.hip    multi-GPU: per-substep halo choreography (two queues, flag or event synchronised), in-process group stepping
//   tetsim_comm.hip    multi-GPU set-up: RCCL communicator, self-test and probe, halo plan
//   tetsim_p2p.hip     peer-to-peer halo: export / connect (HIP IPC mappings of the neighbours' ghost ranges)
//   tetsim_host.cpp    host-only entry points: preprocessing, partition plans, the .tetsim container
//
// There is NO CPU fallback: every compute entry point needs a working HIP device and fails with
// TETSIM_ENODEVICE / TETSIM_EHIP otherwise.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/tetsim.h"
#include "dev_common.h"
#include "dev_store.h"
#include "host_prep.h"
#include "mesh_file.h"

namespace tetsim {

// device words of the flag-synchronised halo path (binary semaphores): [0] G, [2] V, [3] re-prediction after a dt change, [4] error, [6] [7] queue probe
constexpr uint32_t kSyncWords = 8;

// ---- RCCL, resolved at run time so single-GPU hosts (and the N-API addon) do not need librccl ----------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool loaded = false, process_wide = false;
    bool load() {
        if (loaded) return true;
        // TETSIM_RCCL_LIB: explicit library path (deployments with several RCCL builds; the test double of tests/mock_rccl)
        if (const char* over = getenv("TETSIM_RCCL_LIB")) {
            lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!lib) { err = std::string("cannot load TETSIM_RCCL_LIB=") + over + ": " + dlerror(); return false; }
        }
        // An RCCL that is ALREADY in the process (PyTorch brings its own) is the one to use: a second copy next to it is two sets of
        // the same symbols, and whichever of the two is loaded later binds some of its own calls to the other (seen on ROCm 7.2
        // boxes, where "librccl.so.1" resolves to /opt/rocm's build and PyTorch's bundled one has another name: `double free or
        // corruption` when the process exits).  So: by name without loading, then whatever the process exports, and only then a
        // fresh copy -- with LOCAL scope and DEEPBIND, so that neither copy ever resolves into the other.
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        }
        if (!lib && dlsym(RTLD_DEFAULT, "ncclCommInitRank") && dlsym(RTLD_DEFAULT, "ncclSend")) process_wide = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (lib || process_wide) break;
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        }
        if (!lib && !process_wide) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(process_wide ? RTLD_DEFAULT : lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        CommUserRank = reinterpret_cast<decltype(CommUserRank)>(sym("ncclCommUserRank"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        loaded = GetUniqueId && CommInitRank && CommDestroy && CommCount && CommUserRank && Send && Recv && GroupStart && GroupEnd && GetErrorString;
        return loaded;
    }
};
// A/B switches of choices that measurement has settled (HISTORY.md) are read by the DEVELOPMENT build only (-DTETSIM_ABLATION); the
// product library does not look at them: TETSIM_QUAD_POLL_DELAY, TETSIM_NH_QUADS, TETSIM_NH_FOLD, TETSIM_NH_FRAME, TETSIM_FRAME_LOCAL,
// TETSIM_HALO_ALIGNED_TILES.  What the product reads is what INTEGRATION.md 4 lists.
inline const char* lab_env(const char* name) {
#ifdef TETSIM_ABLATION
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
extern Rccl g_rccl;
// Bumped whenever this library creates a stream in the process (every handle's main stream, every halo stream): HIP does not pin a
// stream to a hardware queue, so a queue-independence probe (probe_queue_independence) is only as good as the set of streams it
// was taken with -- a body re-probes before it replays its two-chain halo graphs whenever this changed since its last probe.
extern std::atomic<uint64_t> g_stream_generation;

// Peer-to-peer halo: where this rank's boundary predictions go at one neighbour -- its ghost run for this rank and the word that
// says "arrived", per substep parity -- in the neighbour's memory (an IPC mapping, a pointer of the same process, or this rank's
// own buffers in loopback measurements).
struct PeerLink {
    float4* ghost[2] = {nullptr, nullptr};
    uint32_t* arrived[2] = {nullptr, nullptr};
    void* ipc[3] = {nullptr, nullptr, nullptr};   // mappings to close (hipIpcCloseMemHandle)
    // two-layer ghost regions (TETSIM_FLAG_DEEP_GHOSTS): this rank's runs in the neighbour's four buffers, two sets each (the sets
    // alternate from one exchange to the next), and its words [set][0 = the even substep's data, 1 = the early, odd-substep data]
    float4* g1_even[2] = {nullptr, nullptr};    // predictions of the neighbour's first-layer ghosts, for its next even substep
    float4* g1_final[2] = {nullptr, nullptr};   // ... and their end-of-substep positions (the neighbour advances them itself)
    float4* g2_even[2] = {nullptr, nullptr};    // predictions of its second-layer ghosts for the even substep
    float4* g2_odd[2] = {nullptr, nullptr};     // ... and for the odd substep before it (evolves its second-layer ghost tets after the fact)
    uint32_t* arrived2[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
};
struct NeighDev {
    int rank = -1;
    uint32_t send_count = 0, recv_start = 0, recv_count = 0;
    bool contiguous = false;
    uint32_t send_first = 0;       // when contiguous: first local id
    int32_t* send_idx = nullptr;   // device, when not contiguous
    float4* send_buf = nullptr;    // device staging, when not contiguous
    std::vector<int32_t> send_global, recv_global, send_local;
};

}  // namespace tetsim

using namespace tetsim;  // (private header of the ABI's own translation units; the handle type lives in the global namespace)

struct tetsim_body {
    std::string err;
    TetSimOptions opt{};
    TetSimInfo info{};
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_bnd_tet = nullptr;
    uint32_t interior_tets = 0;         // tets of the interior tiles (blocked, partitioned)
    bool queues_probed = false, queues_independent = false;   // flag path: do the two streams run on independent hardware queues?
    uint64_t probe_generation = 0;      // g_stream_generation at the time of that probe
    std::map<uint32_t, std::pair<hipGraphExec_t, hipGraphExec_t>> flag_graphs;   // n substeps -> (main chain, halo chain), replayed side by side
    uint32_t* d_sync = nullptr;         // device counters of the flag-synchronised halo path: G done/taken, V done/taken, error
    bool flag_sync = false;             // this body steps through the flag-synchronised path (blocked + transport)
    bool halo_graph_broken = false;
    bool halo_warm = false;             // RCCL bodies: one eager call has run (connections are set up before any capture)
    bool loopback = false;              // measurement only (TETSIM_DEBUG_LOOPBACK_HALO): every neighbour is this rank itself
    bool vel_dead = false;              // the substep being enqueued is not the last of its call: its particle kernels leave the velocity array alone (nobody reads it before the call's last substep)
    bool needs_halo_refresh = false;    // in-process group: predictions were redone for a new dt
    bool fork_needed = true;            // first substep of a step call: the boundary stream must see the main stream's history
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_halo = nullptr;
    // halo choreography events, double buffered by substep parity: an event is never re-recorded while a wait that
    // other streams enqueued on its previous record may still be pending
    hipEvent_t ev_boundary2[2] = {nullptr, nullptr}, ev_packed2[2] = {nullptr, nullptr}, ev_sent2[2] = {nullptr, nullptr};
    uint32_t halo_parity = 0;
    DevParams* d_params = nullptr;
    DevParams* d_params_halo = nullptr;   // the same parameters, copied on the HALO stream (its boundary-particle pass reads them)
    std::vector<int32_t> tet_colour;  // copy of TetSimOptions.tet_colour (create only)
    int32_t grab_global = -1;
    int32_t grab_ref[2] = {-1, -1};  // particles the reference's indexFromUV pins for grab_global (TETSIM_FLAG_REF_GRAB_TEXEL)
    float grab_pos[3] = {0, 0, 0};
    std::map<uint32_t, hipGraphExec_t> graphs;
    std::vector<void*> allocs;
    std::vector<float> h_verts;
    std::vector<int32_t> h_tets;
    // tetsim_create_batch: first particle / first tet of every body in the concatenation, [bodies + 1]; empty = a single body
    std::vector<uint32_t> batch_first_vert, batch_first_tet;
    bool fast = false;

    // POLAR_JACOBI
    PJDev pj;
    PJBlk blk;             // blocked formulation (FAST unless TETSIM_FLAG_GATHER_FORMULATION)
    bool blocked = false;
    // fused particle pass (unpartitioned blocked bodies): tetsim_step_n runs  tet | fused x (n-1) | particle  instead of n x (tet | particle)
    bool halo_use_flags = true;       // TETSIM_HALO_SYNC (read when the body is created): false = "events", the older event-synchronised halo path
    bool halo_use_graph = true;       // TETSIM_HALO_GRAPH (likewise): false = the halo path stays eager
    bool fused = false;
    // large unpartitioned blocked bodies (two kernels per substep): tetsim_step_n runs a CALL as ONE launch -- per substep the tiles, then the
    // particles, handed on by stamped partial sums and predictions (pj_blocked.hip: pjb_call_kernel); TETSIM_PJ_ONE_LAUNCH=0 at creation keeps
    // two kernels per substep (A/B); tetsim_step and tetsim_profile keep the two kernels (same arithmetic, same bits)
    bool pj_one_launch = false;
    uint32_t* d_substep_err = nullptr;
    // TETSIM_FLAG_LEAN_STATE (pj_blocked.hip: kModeLeanState): the substep neither reads nor writes pj.quat; it is recovered from the carried
    // shape and this constant centred rest shape when somebody asks for it (ensure_quats)
    float4 *rest0_a = nullptr, *rest0_b = nullptr, *rest0_c = nullptr;
    bool quat_stale = false;          // substeps have been enqueued since pj.quat was last recovered
    // persistent frame kernel (pjb_frame_kernel): tetsim_step_n runs ONE launch per call; fused bodies of few enough tiles
    bool frame = false;
    uint32_t frame_epoch = 1;         // sequence number of the next call's first substep (DevParams::epoch), advanced by 65536 per parameter push
    uint32_t fold_wave_limit = 0, fold_tile_limit = 0;   // waves / workgroups that may wait in-kernel on this device (pjb_wait_capacity, at creation)
    bool quad = false;                // SMALL body: 64-tet tiles, one tet / one particle on four lanes (pj_quad.hip) -- tetsim_step runs pjq_tet + pjq_vertex,
                                      // tetsim_step_n the persistent pjq_frame_kernel (while `frame`); the three agree bit for bit
    uint32_t* d_frame_err = nullptr;  // raised by a tile whose neighbour's partial sums never arrived (bounded wait)
    int32_t* d_block_tile = nullptr;  // [frame_blocks] tile of every block of the frame kernel's grid, -1 = none
    uint32_t frame_blocks = 0;
    DevParams params_on_device;       // what d_params holds (valid while params_known)
    bool params_known = false;
    bool frame_local = false;         // every body's tiles share one XCD: the exchange is coherent in that XCD's L2 (pjb_frame_kernel_local)
    bool frame_turn_counted = false;  // this body is in its device's count of exclusive bodies
    bool epoch_block_fresh = false;   // push_params took a block of sequence numbers that no launch has used yet
    bool frame_exclusive = false;     // needs MORE than half the device's resident workgroups: persistent launches of this device take turns (tetsim_api.hip: FrameTurn)
    float4* partial_b = nullptr;      // second buffer of the tile partial sums
    size_t partial_slots = 0;         // float4s in each of the two
    float4* pos_final_b = nullptr;    // second buffer of the end-of-substep positions (a call always ENDS in pj.pos_final)
    bool fold_halo = false;           // peer-to-peer halo: the halo-side tiles do their queue's hand-overs themselves (TETSIM_HALO_FOLD_WAIT=0: a wait kernel in front)
    bool fold_possible = false;       // ... this body could (interior tiles and interior particles exist, not switched off)
    bool fold_wait = false;           // flag path: the interior particle kernel awaits G itself (TETSIM_HALO_FOLD_WAIT=0 at creation: a wait kernel in front of it)
    bool v_pending = false;           // flag path: the interior particles of the last enqueued substep are not signalled yet (flush_v)
    uint32_t fuse_step = 0;           // substep index inside the current run (enqueue_substep)
    bool fin_in_b = false;            // the latest end-of-substep positions are in pos_final_b (only between the kernels of one call)
    std::vector<int32_t> tet_perm;  // blocked: device tet position -> local tet index
    // Particles are renumbered on the device (Morton order inside the interior segment) for locality; the API keeps
    // the caller's / the partition plan's numbering.  api2dev[a] = device index of API-local particle a.
    std::vector<uint32_t> api2dev, dev2api;
    Partition part;
    bool partitioned = false;
    std::vector<int32_t> g2l_owned;  // global vertex -> local id (owned) or -1
    std::vector<NeighDev> neigh;
    bool pred_any_dt = true;  // velocities are all zero: the prediction is valid for every dt
    float dt_pred = 0.0f;
    ncclComm_t comm = nullptr;
    int comm_rank = -1, comm_size = 0;
    // peer-to-peer halo (tetsim_halo_p2p_export / _connect): no transfer kernel -- the boundary-particle kernel stores into the
    // neighbours' ghost ranges, double buffered by substep parity (pos_pred's tail | ghost_alt)
    // two-layer ghost region (TETSIM_FLAG_DEEP_GHOSTS): ghosts [nv_owned, nv_owned + n_ghost1) are advanced here on even substeps,
    // the second-layer ghost tets are tiles [nb_first, nb) -- solved on even substeps, evolved after the fact on odd ones
    bool deep = false;
    uint32_t n_ghost1 = 0, nb_first = 0;
    std::vector<int32_t> g2l_ghost1;      // global vertex -> local id of a first-layer ghost, or -1 (grab)
    bool p2p = false;
    uint32_t timeout_ms = 0;              // bound of the device-side waits of this body (TETSIM_HALO_TIMEOUT_MS when it was created / connected)
    float4* ghost_alt = nullptr;          // [nv_local - nv_owned] the ghost buffer of odd substeps
    uint32_t* d_arrived = nullptr;        // [2][kMaxPeers] words the neighbours raise here
    uint32_t* d_peer_slots = nullptr;     // ELL [p2p_cols][p2p_stride]: where a boundary particle goes at which neighbour
    uint32_t p2p_cols = 0, p2p_stride = 0;
    uint32_t* d_peer_slots2 = nullptr;    // two-layer regions: the same for the neighbours' SECOND layer (disjoint lists)
    uint32_t p2p_cols2 = 0;
    // two-layer regions, own receive buffers: ghost_alt holds [g1_even x2 | g1_final x2 | g2_even x2 | g2_odd x2]
    float4* own_g1_even[2] = {nullptr, nullptr};
    float4* own_g1_final[2] = {nullptr, nullptr};
    float4* own_g2_even[2] = {nullptr, nullptr};
    float4* own_g2_odd[2] = {nullptr, nullptr};
    std::vector<PeerLink> links;          // parallel to `neigh`
    uint32_t p2p_probe_base = 0;          // tetsim_halo_p2p_probe: the inbox words count on from here (every rank calls it the same number of times)
    uint64_t p2p_round = 0;               // substeps enqueued since the connection; its parity selects the buffers
    bool p2p_raise_pending = false;       // the last boundary-particle kernel's "arrived" has not been raised yet (no kernel behind it)
    bool halo_pending = false;            // a halo was started and nobody has waited for it yet
    bool final_ghosts_fresh = false;      // pos_final's ghost range holds the neighbours' END-OF-SUBSTEP positions of the current state (tetsim_halo_refresh_final)
    std::vector<tetsim_body*> group;      // in-process group transport: partition i of the decomposition (or empty)

    SkinDev skin;  // embedded visual mesh
    std::vector<int32_t> vis_global;   // row of the caller's visVerts behind each attached visual vertex (a partition keeps the rows of the tets it owns)
    bool vis_attached = false;
    uint32_t vis_total = 0;            // rows of the caller's visVerts (a partition keeps num_vis_verts of them)
    float4* d_vis_full = nullptr;      // partitions with visual triangles: every rank's skin put together, [vis_total] (tetsim_visual_vertex_normals_from)
    float* pinned_pos = nullptr;   // tetsim_read_positions_pinned: host-pinned xyz
    float* d_packed = nullptr;     //   and its device-side staging
    float* pinned_quat = nullptr;  // tetsim_read_quats_pinned: host-pinned xyzw per local tet
    uint32_t* d_api2dev = nullptr; // device copy of api2dev (pack / nearest kernels), null = identity
    double* d_best = nullptr; uint32_t* d_best_id = nullptr;  // tetsim_start_grab candidates

    // NEOHOOKEAN_GS
    NHDev nh;
    std::vector<NHClusterLaunch> cluster_launch;  // TETSIM_ORDER_CLUSTERED: one per cluster colour
    int32_t* d_slot_vid = nullptr;
    // clustered schedule: the particle pass between two substeps of a run is folded into the sweep's first touchers
    // (NHClusterLaunch::first_mask); nh_untouched lists the particles no cluster touches, which keep a (tiny) pass of their own
    bool nh_fold = false;
    uint8_t* d_first_mask = nullptr;
    uint32_t* d_nh_untouched = nullptr;
    uint32_t nh_untouched = 0;
    // clustered FAST bodies: the sweep as ONE launch per substep, particles handed on with their stamp (dev_common.h: NHSweep);
    // TETSIM_NH_ONE_LAUNCH=0 at creation keeps one launch per colour (A/B); tetsim_profile always does (it times the colour kernels)
    bool nh_one_launch = false;
    bool nh_call = false;             // ... and tetsim_step_n runs the sweeps of ALL its substeps as one launch (nh_call_kernel: every particle touched by some cluster, <= 127 colours)
    NHSweep nh_sweep1;
    uint32_t nh_sub_index = 0;        // substep inside the run being enqueued (enqueue_substep: first -> 0)
    uint32_t nh_epoch_arg = 0;        // tetsim_step: a block of stamps of its own as a kernel argument; 0 = DevParams::epoch (tetsim_step_n)
    std::vector<uint32_t> level_off;
    // small bodies (all particles fit one CU's LDS, level schedules): tetsim_step_n runs a call as ONE single-workgroup launch
    bool nh_frame = false;
    NHFrameLaunch nh_frame_launch;
    std::vector<uint32_t> nh_seg;      // host copy of nh_frame_launch.seg ([levels][bodies][2]): the stepwise FAST twin makes the frame kernel's choice per piece
    std::vector<int32_t> order;
    std::vector<float> h_inv_mass;
};

namespace tetsim {

#define HIPCHK(h, call)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
            return TETSIM_EHIP;                                                                         \
        }                                                                                               \
    } while (0)

int fail(tetsim_body* h, int code, const std::string& msg);  // records the message on the handle (or for tetsim_last_error(NULL))
const char* create_error();
void group_begin(tetsim_body* const* hs, uint32_t count);             // entry points over several handles: clear the members' messages ...
int group_result(tetsim_body* const* hs, uint32_t count, int rc);     // ... and publish the failing member's for tetsim_last_error(NULL)

template <class Tp>
inline int dev_alloc(tetsim_body* h, Tp** p, size_t count) {
    *p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(Tp);
    void* raw = nullptr;
    hipError_t e = hipMalloc(&raw, bytes);
    if (e != hipSuccess) { h->err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e); return TETSIM_ENOMEM; }
    h->allocs.push_back(raw);
    h->info.device_bytes += bytes;
    *p = static_cast<Tp*>(raw);
    return 0;
}
template <class Tp>
inline int upload(tetsim_body* h, Tp* dst, const std::vector<Tp>& src) {
    if (src.empty()) return 0;
    HIPCHK(h, hipMemcpy(dst, src.data(), src.size() * sizeof(Tp), hipMemcpyHostToDevice));
    return 0;
}

// SoftbodyGPU.js:335-338,345: texel (px,py) of the R x R position texture is pinned when

void ref_grab_texels(int32_t grab_id, uint32_t num_elems, uint32_t num_particles, int32_t out[2]);
// reuse_ok: the caller's kernels do not need a fresh block of sequence numbers (tetsim_step) -- if the parameters equal what the device
// already holds, nothing is uploaded
int push_params(tetsim_body* h, double dt, const TetSimParams* params, bool reuse_ok = false);

// Development: TETSIM_DEBUG_HOSTPROF=1 accumulates the host time of every call in the eager halo path, printed at destroy.
struct HostProf {
    bool on = [] { const char* e = getenv("TETSIM_DEBUG_HOSTPROF"); return e && e[0] == '1'; }();
    std::map<std::string, std::pair<double, uint64_t>> acc;
    ~HostProf() { for (auto& kv : acc) fprintf(stderr, "[hostprof] %-28s %8.2f us avg over %llu calls\n", kv.first.c_str(), kv.second.first / kv.second.second, (unsigned long long)kv.second.second); }
};
extern HostProf g_hostprof;
struct HostProfScope {
    const char* label; std::chrono::steady_clock::time_point t0;
    explicit HostProfScope(const char* l) : label(l) { if (g_hostprof.on) t0 = std::chrono::steady_clock::now(); }
    ~HostProfScope() { if (g_hostprof.on) { auto& a = g_hostprof.acc[label]; a.first += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); a.second++; } }
};
#define HP(label) HostProfScope hp_scope_##__LINE__(label)

// ---- kernel sequencing (tetsim_api.hip)
void pj_tet(tetsim_body* h, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pj_vertex(tetsim_body* h, uint32_t first, uint32_t count, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
void pj_fused_substep(tetsim_body* h, bool first, bool last, hipEvent_t* e);   // one substep of a fused body (tet | fused x (n-1) | particle)
void pj_repredict(tetsim_body* h);
void nh_sweep(tetsim_body* h, bool fold = false, bool last = true, bool one_launch = false);   // one_launch: bodies with nh_one_launch take the single-launch sweep (enqueue_substep; tetsim_profile keeps the colour kernels)   // fold: first touchers do the particle pass between two substeps; last: the sweep whose volError the call leaves behind
// first / last: position inside a run of substeps enqueued back to back with one dt
int enqueue_substep(tetsim_body* h, bool first = true, bool last = true);
int ensure_prediction(tetsim_body* h, double dt);
int read_float4_as_xyz(tetsim_body* h, const float4* src, uint32_t n, float* out);
// ---- state read-back helpers (tetsim_state.hip)
const float4* current_positions(tetsim_body* h);   // end-of-substep positions of either solver
int ensure_quats(tetsim_body* h);                  // lean-state bodies: pj.quat brought up to date on h->stream (behind both queues' work); else nothing
int ensure_index_map(tetsim_body* h);              // device copy of api2dev (pack / nearest kernels)

// ---- construction (tetsim_create.hip)
int create_polar(tetsim_body* h, const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt);
int create_neohookean(tetsim_body* h, const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt);

// ---- halo (tetsim_halo.hip)
int create_halo_stream(tetsim_body* h);
int rccl_fail(tetsim_body* h, ncclResult_t r, const char* what);
int halo_start(tetsim_body* h);
int halo_wait(tetsim_body* h, hipStream_t on);
uint32_t halo_timeout_ms(const tetsim_body* h = nullptr);   // the body's bound, or the environment's
bool has_transport(const tetsim_body* h);
bool uses_flag_sync(const tetsim_body* h);
int enqueue_phase_a(tetsim_body* h, hipEvent_t* ev = nullptr);  // tet kernels + particles; ev[0..3]: begin/end of the interior tet and the particle kernel
int flush_v(tetsim_body* h);                                    // flag path: the V hand-over that no following substep will carry
int enqueue_phase_b(tetsim_body* h, bool refresh = false);      // halo start (peer-to-peer bodies: only for the refresh after a dt change)
float4* ghost_buffer(tetsim_body* h, uint32_t parity);          // where ghost particle nv_owned + i of that substep parity lives (+ i)
int probe_queue_independence(tetsim_body* h);                   // flag path: may the two chains be replayed from graphs?
int step_n_flag_graphs(tetsim_body* h, uint32_t n);             // n substeps of an RCCL flag-path body as two captured chains
void frame_turn_enter(tetsim_body* h);                         // an exclusive persistent-launch body joins its device's turn-taking (tetsim_api.hip)
int refresh_final_rccl(tetsim_body* h);                         // RCCL bodies: the ghosts' end-of-substep positions into pos_final's ghost range (collective)
void drop_flag_graphs(tetsim_body* h);                          // destroy the captured chains (queues turned out not to be independent)

}  // namespace tetsim
