// tetsim_api.hip -- C ABI of libtetsim_hip.so (include/tetsim.h): handle lifecycle, host preprocessing,
// stream/graph orchestration of the gfx950 kernels, halo transports (in-process copies, RCCL over xGMI).
//
// There is NO CPU fallback: every compute entry point needs a working HIP device and fails with
// TETSIM_ENODEVICE / TETSIM_EHIP otherwise.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/tetsim.h"
#include "dev_common.h"
#include "dev_store.h"
#include "host_prep.h"
#include "mesh_file.h"

using namespace tetsim;

namespace {

thread_local std::string g_create_error;
constexpr int kRing = 64;  // pinned parameter slots in flight
// device words of the flag-synchronised halo path: [0] G flag, [2] V flag, [4] error
constexpr uint32_t kSyncWords = 8;

// ---- RCCL, resolved at run time so single-GPU hosts (and the N-API addon) do not need librccl ----------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        // TETSIM_RCCL_LIB: explicit library path (deployments with several RCCL builds; the test double of tests/mock_rccl)
        if (const char* over = getenv("TETSIM_RCCL_LIB")) {
            lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!lib) { err = std::string("cannot load TETSIM_RCCL_LIB=") + over + ": " + dlerror(); return false; }
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return GetUniqueId && CommInitRank && CommDestroy && Send && Recv && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;

struct NeighDev {
    int rank = -1;
    uint32_t send_count = 0, recv_start = 0, recv_count = 0;
    bool contiguous = false;
    uint32_t send_first = 0;       // when contiguous: first local id
    int32_t* send_idx = nullptr;   // device, when not contiguous
    float4* send_buf = nullptr;    // device staging, when not contiguous
    std::vector<int32_t> send_global, recv_global, send_local;
};

}  // namespace

struct tetsim_body {
    std::string err;
    TetSimOptions opt{};
    TetSimInfo info{};
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_bnd_tet = nullptr;
    uint32_t interior_tets = 0;         // tets of the interior tiles (blocked, partitioned)
    uint32_t halo_seq = 0;              // substep sequence number of the flag-synchronised path
    uint32_t* d_sync = nullptr;         // device counters of the flag-synchronised halo path: G done/taken, V done/taken, error
    bool flag_sync = false;             // this body steps through the flag-synchronised path (blocked + transport)
    bool halo_graph_broken = false;
    bool halo_warm = false;             // RCCL bodies: one eager call has run (connections are set up before any capture)
    bool loopback = false;              // measurement only (TETSIM_DEBUG_LOOPBACK_HALO): every neighbour is this rank itself
    bool needs_halo_refresh = false;    // in-process group: predictions were redone for a new dt
    bool fork_needed = true;            // first substep of a step call: the boundary stream must see the main stream's history
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_halo = nullptr;
    // halo choreography events, double buffered by substep parity: an event is never re-recorded while a wait that
    // other streams enqueued on its previous record may still be pending
    hipEvent_t ev_boundary2[2] = {nullptr, nullptr}, ev_packed2[2] = {nullptr, nullptr}, ev_sent2[2] = {nullptr, nullptr};
    uint32_t halo_parity = 0;
    DevParams* d_params = nullptr;
    DevParams* h_ring = nullptr;  // pinned [kRing]
    hipEvent_t ring_ev[kRing] = {};
    bool ring_used[kRing] = {};
    int ring_pos = 0;
    std::vector<int32_t> tet_colour;  // copy of TetSimOptions.tet_colour (create only)
    int32_t grab_global = -1;
    int32_t grab_ref[2] = {-1, -1};  // particles the reference's indexFromUV pins for grab_global (TETSIM_FLAG_REF_GRAB_TEXEL)
    float grab_pos[3] = {0, 0, 0};
    std::map<uint32_t, hipGraphExec_t> graphs;
    std::vector<void*> allocs;
    std::vector<float> h_verts;
    std::vector<int32_t> h_tets;
    bool fast = false;

    // POLAR_JACOBI
    PJDev pj;
    PJBlk blk;             // blocked formulation (FAST unless TETSIM_FLAG_GATHER_FORMULATION)
    bool blocked = false;
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
    bool halo_pending = false;            // a halo was started and nobody has waited for it yet
    std::vector<tetsim_body*> group;      // in-process group transport: partition i of the decomposition (or empty)

    SkinDev skin;  // embedded visual mesh
    float* pinned_pos = nullptr;   // tetsim_read_positions_pinned: host-pinned xyz
    float* d_packed = nullptr;     //   and its device-side staging
    uint32_t* d_api2dev = nullptr; // device copy of api2dev (pack / nearest kernels), null = identity
    double* d_best = nullptr; uint32_t* d_best_id = nullptr;  // tetsim_start_grab candidates

    // NEOHOOKEAN_GS
    NHDev nh;
    std::vector<NHClusterLaunch> cluster_launch;  // TETSIM_ORDER_CLUSTERED: one per cluster colour
    int32_t* d_slot_vid = nullptr;
    std::vector<uint32_t> level_off;
    std::vector<int32_t> order;
    std::vector<float> h_inv_mass;
};

namespace {

#define HIPCHK(h, call)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
            return TETSIM_EHIP;                                                                         \
        }                                                                                               \
    } while (0)

int fail(tetsim_body* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

template <class Tp>
int dev_alloc(tetsim_body* h, Tp** p, size_t count) {
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
int upload(tetsim_body* h, Tp* dst, const std::vector<Tp>& src) {
    if (src.empty()) return 0;
    HIPCHK(h, hipMemcpy(dst, src.data(), src.size() * sizeof(Tp), hipMemcpyHostToDevice));
    return 0;
}

// SoftbodyGPU.js:335-338,345: texel (px,py) of the R x R position texture is pinned when
// float(int(uv.x*(R-1)) + int(uv.y*(R-1)*R)) == grabId with uv = (px+.5, py+.5)/R, all in f32.  Rows cannot collide
// (the y term advances by R-1 per row and the x term is below R-1), columns px and px+1 can.
void ref_grab_texels(int32_t grab_id, uint32_t num_elems, uint32_t num_particles, int32_t out[2]) {
    out[0] = out[1] = -1;
    if (grab_id < 0) return;
    const int R = static_cast<int>(std::ceil(std::sqrt(static_cast<double>(num_elems))));
    const float Rf = static_cast<float>(R), Rm1 = Rf - 1.0f;
    int n = 0;
    for (int py = 0; py < R && n < 2; py++) {
        const float uy = (static_cast<float>(py) + 0.5f) / Rf;
        const int row = static_cast<int>((uy * Rm1) * Rf);
        const int a = grab_id - row;
        if (a < 0 || a >= R) continue;
        for (int px = std::max(0, a - 1); px <= std::min(R - 1, a + 1) && n < 2; px++) {
            const float ux = (static_cast<float>(px) + 0.5f) / Rf;
            const int idx = static_cast<int>(ux * Rm1) + row;
            const int64_t particle = static_cast<int64_t>(py) * R + px;
            if (static_cast<float>(idx) == static_cast<float>(grab_id) && particle < static_cast<int64_t>(num_particles)) out[n++] = static_cast<int32_t>(particle);
        }
    }
}

void fill_params(const tetsim_body* h, double dt, const TetSimParams& p, DevParams* o) {
    std::memset(o, 0, sizeof(*o));
    o->dt = static_cast<float>(dt);
    o->gravity = static_cast<float>(p.gravity);
    o->friction = static_cast<float>(p.friction);
    const bool fixed = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI && (h->opt.flags & TETSIM_FLAG_REF_FIXED_BOUNDS);
    const float ref_lo[3] = {-2.5f, -1.0f, -2.5f}, ref_hi[3] = {2.5f, 10.0f, 2.5f};  // SoftbodyGPU.js:347
    for (int c = 0; c < 3; c++) {
        o->lo[c] = fixed ? ref_lo[c] : static_cast<float>(p.worldBounds[c]);
        o->hi[c] = fixed ? ref_hi[c] : static_cast<float>(p.worldBounds[3 + c]);
        o->d_lo[c] = p.worldBounds[c];
        o->d_hi[c] = p.worldBounds[3 + c];
        o->grab[c] = h->grab_pos[c];
    }
    auto to_device = [&](int32_t global) -> int32_t {
        if (global < 0) return -1;
        int32_t a = -1;  // API-local index
        if (!h->partitioned) a = global;
        else if (static_cast<size_t>(global) < h->g2l_owned.size()) a = h->g2l_owned[global];
        if (a < 0) return -1;
        return h->api2dev.empty() ? a : static_cast<int32_t>(h->api2dev[a]);
    };
    o->grab_local = o->grab_local2 = -1;
    if (h->grab_global >= 0) {
        if (h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI && (h->opt.flags & TETSIM_FLAG_REF_GRAB_TEXEL)) {
            o->grab_local = to_device(h->grab_ref[0]);
            o->grab_local2 = to_device(h->grab_ref[1]);
        } else o->grab_local = to_device(h->grab_global);
    }
    o->d_dt = dt;
    o->d_gravity = p.gravity;
    o->d_friction = p.friction;
    o->d_dev_compliance = p.devCompliance;
    o->d_vol_compliance = p.volCompliance;
}

// Stage the parameters of this call into a pinned ring slot and copy them to the device in stream order.
int push_params(tetsim_body* h, double dt, const TetSimParams* params) {
    if (!params) return fail(h, TETSIM_EINVAL, "params is null");
    if (!(dt > 0.0) || !std::isfinite(dt)) return fail(h, TETSIM_EINVAL, "dt must be a positive finite number");
    const int slot = h->ring_pos;
    h->ring_pos = (h->ring_pos + 1) % kRing;
    if (h->ring_used[slot]) HIPCHK(h, hipEventSynchronize(h->ring_ev[slot]));
    fill_params(h, dt, *params, &h->h_ring[slot]);
    HIPCHK(h, hipMemcpyAsync(h->d_params, &h->h_ring[slot], sizeof(DevParams), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(h->ring_ev[slot], h->stream));
    h->ring_used[slot] = true;
    h->fork_needed = true;  // whatever the caller did on the main stream since the last call must be visible to the boundary stream
    return 0;
}

// Development: TETSIM_DEBUG_HOSTPROF=1 accumulates the host time of every call in the eager halo path, printed at destroy.
struct HostProf {
    bool on = [] { const char* e = getenv("TETSIM_DEBUG_HOSTPROF"); return e && e[0] == '1'; }();
    std::map<std::string, std::pair<double, uint64_t>> acc;
    ~HostProf() { for (auto& kv : acc) fprintf(stderr, "[hostprof] %-28s %8.2f us avg over %llu calls\n", kv.first.c_str(), kv.second.first / kv.second.second, (unsigned long long)kv.second.second); }
} g_hostprof;
struct HostProfScope {
    const char* label; std::chrono::steady_clock::time_point t0;
    explicit HostProfScope(const char* l) : label(l) { if (g_hostprof.on) t0 = std::chrono::steady_clock::now(); }
    ~HostProfScope() { if (g_hostprof.on) { auto& a = g_hostprof.acc[label]; a.first += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); a.second++; } }
};
#define HP(label) HostProfScope hp_scope_##__LINE__(label)

// ---- kernel sequencing ---------------------------------------------------------------------------------
void pj_tet(tetsim_body* h, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if (h->blocked) pjb_launch_tet(h->stream, h->blk, 0, h->blk.nb, e0, e1);
    else h->fast ? pj_launch_tet_fast(h->stream, h->pj, e0, e1) : pj_launch_tet_precise(h->stream, h->pj, e0, e1);
}
void pj_vertex(tetsim_body* h, uint32_t first, uint32_t count, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if (h->blocked) pjb_launch_vertex(h->stream, h->blk, first, count, e0, e1);
    else h->fast ? pj_launch_vertex_fast(h->stream, h->pj, first, count, e0, e1) : pj_launch_vertex_precise(h->stream, h->pj, first, count, e0, e1);
}
void pj_repredict(tetsim_body* h) {
    if (h->blocked) pjb_launch_repredict(h->stream, h->blk);
    else h->fast ? pj_launch_repredict_fast(h->stream, h->pj) : pj_launch_repredict_precise(h->stream, h->pj);
}

// The halo stream carries the transfers AND the boundary tiles that consume them; high priority so that its few
// workgroups are dispatched ahead of the interior kernel's backlog.
// NEOHOOKEAN_GS: the Gauss-Seidel sweep over all tets (A3-A5), as dependency levels or as cluster colours
void nh_sweep(tetsim_body* h) {
    if (!h->cluster_launch.empty()) {
        for (const NHClusterLaunch& L : h->cluster_launch) h->fast ? nh_launch_cluster_fast(h->stream, h->nh, L) : nh_launch_cluster_precise(h->stream, h->nh, L);
        return;
    }
    for (size_t l = 0; l + 1 < h->level_off.size(); l++) {
        const uint32_t first = h->level_off[l], count = h->level_off[l + 1] - first;
        h->fast ? nh_launch_level_fast(h->stream, h->nh, first, count) : nh_launch_level_precise(h->stream, h->nh, first, count);
    }
}

int create_halo_stream(tetsim_body* h) {
    if (h->comm_stream) return 0;
    int lo = 0, hi = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(h, hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_bnd_tet, hipEventDisableTiming));
    return 0;
}

int rccl_fail(tetsim_body* h, ncclResult_t r, const char* what) {
    return fail(h, TETSIM_ECOMM, std::string(what) + ": " + g_rccl.GetErrorString(r));
}

// Start this substep's halo: owned interface predictions -> the neighbours' ghost ranges, on the halo stream.
// Two transports share this choreography: RCCL (one process per GPU) and, for partitions living in ONE process
// (tests, "multi-GPU without a cluster"), asynchronous device copies issued by the sender.
//
// Dependencies (p = substep parity; all partitions of a group advance in lock-step on the host, so parities agree):
//   boundary[p]  recorded on the main stream after this substep's boundary-particle pass (hence after its tet kernels)
//   packed[p]    = boundary[p] + the pack kernels of non-contiguous send lists
//   a transfer into partition D's ghosts waits for D's boundary[p]: D's tet kernels of this substep have read them
//   sent[p]      recorded on the halo stream after this partition's transfers (RCCL: sends AND receives)
//   the next substep's first ghost-reading tet kernel waits for every neighbour's sent[p] -- and for OUR sent[p], because
//   our next boundary pass overwrites the very buffer our transfer reads
int halo_start(tetsim_body* h) {
    const uint32_t p = h->halo_parity;
    if (h->flag_sync) {  // the halo stream already waited for this substep's particle pass (wait V): stay in stream order
        for (auto& nb : h->neigh)
            if (!nb.contiguous && nb.send_count) util_launch_gather4(h->comm_stream, h->pj.pos_pred, nb.send_idx, nb.send_buf, nb.send_count);
    } else {
        for (auto& nb : h->neigh)
            if (!nb.contiguous && nb.send_count) util_launch_gather4(h->stream, h->pj.pos_pred, nb.send_idx, nb.send_buf, nb.send_count);
        { HP("record packed"); HIPCHK(h, hipEventRecord(h->ev_packed2[p], h->stream)); }
        { HP("comm wait packed"); HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_packed2[p], 0)); }
    }
    static const bool lb_copy = [] { const char* e = getenv("TETSIM_DEBUG_LOOPBACK_COPY"); return e && e[0] == '1'; }();
    if (h->comm && h->loopback && lb_copy) {  // measurement only: the loopback transfer as a plain copy kernel instead of RCCL
        for (auto& nb : h->neigh)
            if (nb.send_count) util_launch_copy(h->comm_stream, nb.contiguous ? h->pj.pos_pred + nb.send_first : nb.send_buf, h->pj.pos_pred + nb.recv_start, nb.send_count);
    } else if (h->comm) {
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) return rccl_fail(h, r, "ncclGroupStart");
        for (auto& nb : h->neigh) {
            if (nb.send_count) {
                const float4* src = nb.contiguous ? h->pj.pos_pred + nb.send_first : nb.send_buf;
                r = g_rccl.Send(src, 4ull * nb.send_count, ncclFloat, h->loopback ? h->comm_rank : nb.rank, h->comm, h->comm_stream);
                if (r != ncclSuccess) return rccl_fail(h, r, "ncclSend");
            }
            if (nb.recv_count) {
                // posted on OUR halo stream, i.e. after our boundary pass of this substep: the ghosts are overwritten only
                // once this partition's tet kernels (which read them) are done
                r = g_rccl.Recv(h->pj.pos_pred + nb.recv_start, 4ull * nb.recv_count, ncclFloat, h->loopback ? h->comm_rank : nb.rank, h->comm, h->comm_stream);
                if (r != ncclSuccess) return rccl_fail(h, r, "ncclRecv");
            }
        }
        r = g_rccl.GroupEnd();
        if (r != ncclSuccess) return rccl_fail(h, r, "ncclGroupEnd");
    } else {  // in-process group: sender-driven copies with the ordering guarantees a posted receive gives
        for (auto& nb : h->neigh) {
            if (!nb.send_count) continue;
            tetsim_body* dst = h->group[nb.rank];
            const NeighDev* back = nullptr;
            for (auto& r : dst->neigh) if (r.rank == h->opt.part_index) back = &r;
            if (!back || back->recv_count != nb.send_count) return fail(h, TETSIM_ESTATE, "asymmetric halo plan");
            { HP("comm wait dst boundary"); HIPCHK(h, hipStreamWaitEvent(h->comm_stream, dst->ev_boundary2[p], 0)); }  // receiver finished reading its ghosts
            const float4* from = nb.contiguous ? h->pj.pos_pred + nb.send_first : nb.send_buf;
            { HP("memcpyAsync d2d"); HIPCHK(h, hipMemcpyAsync(dst->pj.pos_pred + back->recv_start, from, nb.send_count * sizeof(float4), hipMemcpyDeviceToDevice, h->comm_stream)); }
        }
    }
    if (!(h->flag_sync && h->comm)) { HP("record sent"); HIPCHK(h, hipEventRecord(h->ev_sent2[p], h->comm_stream)); }  // (RCCL + flags: stream order is all there is)
    h->halo_pending = true;
    return 0;
}
// Make this partition's main stream wait until the previous substep's halo is complete (see halo_start).
int halo_wait(tetsim_body* h, hipStream_t on) {
    if (!h->halo_pending) return 0;
    const uint32_t p = h->halo_parity ^ 1u;  // the previous substep's parity
    // our own transfers (RCCL: includes our receives); implied by stream order when the consumer runs on the halo stream
    if (on != h->comm_stream) { HP("wait own sent"); HIPCHK(h, hipStreamWaitEvent(on, h->ev_sent2[p], 0)); }
    if (!h->comm)
        for (auto& nb : h->neigh)
            if (nb.recv_count) { HP("wait peer sent"); HIPCHK(h, hipStreamWaitEvent(on, h->group[nb.rank]->ev_sent2[p], 0)); }
    h->halo_pending = false;
    return 0;
}
// Bound of the device-side waits of the flag path.  `wait G` sits behind a transfer, i.e. behind the PEER's progress: a rank
// that steps this much later than its neighbour is reported as TETSIM_ECOMM at the next synchronisation.  0 = wait for ever.
uint32_t halo_timeout_ms() {
    static const uint32_t ms = [] { const char* e = getenv("TETSIM_HALO_TIMEOUT_MS"); return e ? static_cast<uint32_t>(strtoul(e, nullptr, 10)) : 30000u; }();
    return ms;
}
bool has_transport(const tetsim_body* h) { return !h->neigh.empty() && (h->comm || !h->group.empty()); }
// blocked bodies with a transport and ghost-touching tiles step through the flag-synchronised two-queue path (enqueue_phase_a)
bool uses_flag_sync(const tetsim_body* h) {
    static const bool use_flags = [] { const char* e = getenv("TETSIM_HALO_SYNC"); return !(e && e[0] == 'e'); }();
    static const bool one_stream = [] { const char* e = getenv("TETSIM_DEBUG_ONE_STREAM"); return e && e[0] == '1'; }();
    return use_flags && !one_stream && has_transport(h) && h->blocked && h->blk.nb > h->blk.nb_interior;
}

// Host cost matters here: a substep is ~42 us of GPU work and every launch / event call costs 1.5-4 us, so the eager
// halo path issues as few operations as possible -- 3 kernel launches (interior tiles, boundary tiles, ONE particle pass),
// 1 event record + 1 cross-stream wait to start the transfer, 1 record after it, 1 wait before the next boundary tiles.
// The transfer overlaps the NEXT substep's interior tet kernel (~30 us), which is ample for a 200 KB message.
//
// In-process groups must issue every partition's particle pass before anyone's sends (a send waits for the RECEIVER's
// boundary event of the same substep), so a substep is enqueued in two phases; RCCL bodies run both back to back.
int enqueue_phase_a(tetsim_body* h, hipEvent_t* ev = nullptr) {  // tet kernels + particles; ev[0..3]: begin/end of the interior tet and the particle kernel
    if (h->blocked) {
        const uint32_t nbnd = h->blk.nb - h->blk.nb_interior;
        static const bool one_stream = [] { const char* e = getenv("TETSIM_DEBUG_ONE_STREAM"); return e && e[0] == '1'; }();
        static const bool use_flags = [] { const char* e = getenv("TETSIM_HALO_SYNC"); return !(e && e[0] == 'e'); }();  // "events" = the older path
        if (nbnd && !one_stream && use_flags) {
            // Two queues, synchronised through device counters instead of events (util_kernels.hip: a cross-stream event costs
            // ~15 us eagerly and ~6 us as a graph edge here, and a substep has two on its critical path):
            //   main stream:  interior tiles(s) -> wait G(s) -> particles(s) -> signal V(s)
            //   halo stream:  ghost tiles G(s) -> signal G(s) -> wait V(s) -> transfer(s)              [transfer(s-1) precedes G(s)]
            // signal / wait are one-wave kernels (pj_blocked.hip).  Host submission order follows the dependencies (G, signal G,
            // interior, wait G, particles, signal V, wait V, transfer): every wait is submitted after its signal, so the path
            // stays live even if the runtime maps both streams onto one hardware queue (it then merely serialises).
            if (!h->d_sync) {
                int rc = dev_alloc(h, &h->d_sync, kSyncWords);
                if (rc) return rc;
                HIPCHK(h, hipMemset(h->d_sync, 0, kSyncWords * sizeof(uint32_t)));
                HIPCHK(h, hipDeviceSynchronize());   // once: the halo stream must also see everything create() uploaded
            }
            h->flag_sync = true;
            const uint32_t seq = ++h->halo_seq;
            PJSync yg, yv;   // word 0: "G tiles of substep seq are done"; word 2: "particles of substep seq are done"
            yg.wait = yg.signal = h->d_sync + 0; yg.error = h->d_sync + 4; yg.seq = seq; yg.timeout_ms = halo_timeout_ms();
            yv.wait = yv.signal = h->d_sync + 2; yv.error = h->d_sync + 4; yv.seq = seq; yv.timeout_ms = yg.timeout_ms;
            int rc = halo_wait(h, h->comm_stream);   // in-process groups: the neighbours' transfers of the previous substep (events)
            if (rc) return rc;
            { HP("launch tet ghost"); pjb_launch_tet(h->comm_stream, h->blk, h->blk.nb_interior, nbnd); }
            { HP("signal G"); pjb_launch_signal(h->comm_stream, yg); }
            if (!h->comm) { HP("record boundary"); HIPCHK(h, hipEventRecord(h->ev_boundary2[h->halo_parity], h->comm_stream)); }  // group transport: ghosts are free again
            { HP("launch tet interior"); pjb_launch_tet(h->stream, h->blk, 0, h->blk.nb_interior, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr); }
            { HP("wait G"); pjb_launch_wait(h->stream, yg); }
            { HP("launch vertex"); pj_vertex(h, 0, h->pj.nv_owned, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr); }
            { HP("signal V"); pjb_launch_signal(h->stream, yv); }
            { HP("wait V"); pjb_launch_wait(h->comm_stream, yv); }
            return 0;
        }
        if (nbnd && !one_stream) {
            // (TETSIM_HALO_SYNC=events) Interior tiles read no ghost and start at once on the main stream.  The few boundary tiles (272 of 3984 on a
            // 1 M-tet slab) are launched on the HALO stream, right behind the transfer they depend on: after the interior
            // kernel on the main stream they cost a whole extra kernel latency (10-16 us: load -> 9 rotation iterations ->
            // store, however few tiles); beside it their workgroups slot in as interior ones retire (the halo stream has high
            // priority).  It also saves host work, which matters at ~2-4 us per HIP call against ~42 us of GPU work per
            // substep: no event between the transfer and its consumer.
            // Ordering: the halo stream is behind packed[p-1], recorded after the previous particle pass, so the boundary
            // kernel is behind everything it reads; the first substep of a call forks explicitly.
            if (h->fork_needed || !h->halo_pending) {
                HP("fork");
                HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
                HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_fork, 0));
                h->fork_needed = false;
            }
            { HP("launch tet interior"); pjb_launch_tet(h->stream, h->blk, 0, h->blk.nb_interior, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr); }
            int rc = halo_wait(h, h->comm_stream);
            if (rc) return rc;
            { HP("launch tet boundary"); pjb_launch_tet(h->comm_stream, h->blk, h->blk.nb_interior, nbnd); }
            { HP("record bnd_tet"); HIPCHK(h, hipEventRecord(h->ev_bnd_tet, h->comm_stream)); }
            { HP("main wait bnd_tet"); HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_bnd_tet, 0)); }
        } else {
            pjb_launch_tet(h->stream, h->blk, 0, h->blk.nb_interior);
            int rc = halo_wait(h, h->stream);
            if (rc) return rc;
            pjb_launch_tet(h->stream, h->blk, h->blk.nb_interior, nbnd);
        }
    } else {
        int rc = halo_wait(h, h->stream);
        if (rc) return rc;
        pj_tet(h);
    }
    { HP("launch vertex"); pj_vertex(h, 0, h->pj.nv_owned, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr); }
    if (!h->comm) { HP("record boundary"); HIPCHK(h, hipEventRecord(h->ev_boundary2[h->halo_parity], h->stream)); }  // group transport only
    return 0;
}
int enqueue_phase_b(tetsim_body* h) {  // halo start
    int rc = halo_start(h);
    if (rc) return rc;
    h->halo_parity ^= 1u;
    return 0;
}

// one substep's launches (parameters already on the device)
// first / last: position inside a run of substeps enqueued back to back with one dt (NEOHOOKEAN_GS fuses the particle pass
// that ends a substep with the prediction that starts the next one)
int enqueue_substep(tetsim_body* h, bool first = true, bool last = true) {
    if (h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI) {
        if (has_transport(h)) {
            int rc = enqueue_phase_a(h);
            if (!rc) rc = enqueue_phase_b(h);
            if (rc) return rc;
        } else {
            pj_tet(h);
            pj_vertex(h, 0, h->pj.nv_owned);
        }
    } else {
        if (first) h->fast ? nh_launch_predict_fast(h->stream, h->nh) : nh_launch_predict_precise(h->stream, h->nh);
        nh_sweep(h);
        if (last) h->fast ? nh_launch_post_fast(h->stream, h->nh) : nh_launch_post_precise(h->stream, h->nh);
        else h->fast ? nh_launch_post_predict_fast(h->stream, h->nh) : nh_launch_post_predict_precise(h->stream, h->nh);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, TETSIM_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return 0;
}

// POLAR_JACOBI keeps x* = x + v*dt precomputed by the previous vertex kernel; redo it if dt changed.
int enqueue_phase_b(tetsim_body* h);
int ensure_prediction(tetsim_body* h, double dt) {
    if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return 0;
    const float fdt = static_cast<float>(dt);
    if (!h->pred_any_dt && fdt != h->dt_pred) {
        if (h->partitioned && !h->neigh.empty() && !has_transport(h))
            return fail(h, TETSIM_ESTATE, "dt changed between substeps on a partitioned body without a transport (ghost predictions would be stale): "
                                          "exchange halos through tetsim_comm_init / tetsim_group_step_n, or keep dt fixed");
        pj_repredict(h);
        // the neighbours' ghost copies of our interface predictions are stale now: one extra halo exchange (every rank sees the
        // same dt change, so every rank does this).  RCCL bodies do it here; an in-process group does it for all its members
        // in tetsim_group_step_n (a copy waits for the RECEIVER's event, so all records must precede all copies).
        if (has_transport(h)) {
            if (h->flag_sync) {  // the halo stream continues only after the new predictions exist: publish / await one more sequence number
                PJSync y;
                y.wait = h->d_sync + 2; y.signal = h->d_sync + 2; y.error = h->d_sync + 4; y.seq = ++h->halo_seq; y.timeout_ms = halo_timeout_ms();
                pjb_launch_signal(h->stream, y);
                pjb_launch_wait(h->comm_stream, y);
            }
            if (h->comm) {
                int rc = enqueue_phase_b(h);
                if (rc) return rc;
            } else h->needs_halo_refresh = true;
        }
    }
    h->pred_any_dt = false;
    h->dt_pred = fdt;
    return 0;
}

int build_graph(tetsim_body* h, uint32_t n, hipGraphExec_t* out) {
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    const bool halo = has_transport(h);
    if (halo) {  // the graph is self-contained: it forks the halo stream from the main stream and joins it before it ends
        h->halo_pending = false;
        h->fork_needed = true;
    }
    for (uint32_t i = 0; i < n && !rc; i++) rc = enqueue_substep(h, i == 0, i + 1 == n);
    if (halo && !rc) {
        hipError_t je = hipStreamWaitEvent(h->stream, h->ev_sent2[h->halo_parity ^ 1u], 0);  // join: the last transfer
        if (je != hipSuccess) rc = fail(h, TETSIM_EHIP, std::string("join: ") + hipGetErrorString(je));
        h->halo_pending = false;
    }
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(h, TETSIM_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(h, TETSIM_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return 0;
}

int read_float4_as_xyz(tetsim_body* h, const float4* src, uint32_t n, float* out) {
    if (!out) return fail(h, TETSIM_EINVAL, "output pointer is null");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    std::vector<float4> tmp(n);
    if (n) HIPCHK(h, hipMemcpy(tmp.data(), src, n * sizeof(float4), hipMemcpyDeviceToHost));
    const bool perm = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI && !h->api2dev.empty();
    for (uint32_t i = 0; i < n; i++) {
        const float4& t = tmp[perm ? h->api2dev[i] : i];  // owned particles keep their segment: api2dev[i] < n
        out[3 * i] = t.x; out[3 * i + 1] = t.y; out[3 * i + 2] = t.z;
    }
    return 0;
}

// ---- construction ----------------------------------------------------------------------------------------
int create_polar(tetsim_body* h, const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt) {
    const TetSimOptions& o = h->opt;
    const bool ref_table = (o.flags & TETSIM_FLAG_REF_SLOT_TABLE) != 0;
    std::vector<int32_t> ltets;   // local connectivity
    std::vector<int32_t> l2g_v, l2g_t;
    uint32_t nvl = nv, nvo = nv, nvb = 0, ntl = nt;
    h->partitioned = o.part_count > 1;
    if (h->partitioned) {
        std::string e = build_partition(tets, nt, nv, o.part_count, o.part_index, o.vert_owner, &h->part);
        if (!e.empty()) return fail(h, TETSIM_EINVAL, e);
        const Partition& P = h->part;
        ltets = P.local_tets;
        l2g_v = P.local_to_global_vert;
        l2g_t = P.local_to_global_tet;
        nvl = static_cast<uint32_t>(l2g_v.size());
        nvo = P.n_owned;
        nvb = P.n_boundary;
        ntl = static_cast<uint32_t>(l2g_t.size());
        h->g2l_owned.assign(nv, -1);
        for (uint32_t i = 0; i < nvo; i++) h->g2l_owned[l2g_v[i]] = static_cast<int32_t>(i);
        h->info.owned_elems = P.owned_tets;
    } else {
        ltets.assign(tets, tets + 4ull * nt);
        h->info.owned_elems = nt;
    }
    // device numbering: Morton order inside the interior segment [nvb, nvo); boundary (halo sends stay contiguous
    // runs) and ghosts (receive ranges) keep the plan's order
    {
        std::vector<float> lv(3ull * nvl);
        for (uint32_t i = 0; i < nvl; i++) {
            const uint32_t g = h->partitioned ? static_cast<uint32_t>(l2g_v[i]) : i;
            lv[3 * i] = verts[3 * g]; lv[3 * i + 1] = verts[3 * g + 1]; lv[3 * i + 2] = verts[3 * g + 2];
        }
        h->dev2api = morton_vertex_order(lv.data(), nvl, nvb, nvo - nvb);
        h->api2dev.resize(nvl);
        for (uint32_t dv = 0; dv < nvl; dv++) h->api2dev[h->dev2api[dv]] = dv;
        for (auto& id : ltets) id = static_cast<int32_t>(h->api2dev[id]);
    }
    const bool quirk_here = ref_table && ntl > 0 && (!h->partitioned || l2g_t[0] == 0);
    Incidence inc = build_incidence(ltets.data(), ntl, nvl, quirk_here, ref_table);
    // Only owned vertices are averaged here; the table rows of ghosts are never read.
    uint32_t maxv = 0;
    for (uint32_t v = 0; v < nvo; v++) maxv = std::max(maxv, inc.offset[v + 1] - inc.offset[v]);

    PJDev& d = h->pj;
    d.nv_local = nvl; d.nv_owned = nvo; d.nv_boundary = nvb; d.nt = ntl;
    d.nt_pad = (ntl + 63u) & ~63u;
    d.nv_pad = (nvo + 63u) & ~63u;
    d.max_valence = maxv;
    h->info.owned_particles = nvo;
    h->info.local_particles = nvl;
    h->info.local_elems = ntl;
    h->info.max_valence = maxv;
    h->info.dropped_slots = inc.dropped;

    int rc;
    if ((rc = dev_alloc(h, &d.pos_pred, nvl))) return rc;
    if ((rc = dev_alloc(h, &d.pos_final, nvl))) return rc;
    if ((rc = dev_alloc(h, &d.vel, nvl))) return rc;
    d.params = h->d_params;

    std::vector<float4> pos(nvl);
    std::vector<float> lverts(3ull * nvl);
    for (uint32_t i = 0; i < nvl; i++) {  // i = device index
        const uint32_t a = h->dev2api[i];
        const uint32_t g = h->partitioned ? static_cast<uint32_t>(l2g_v[a]) : a;
        pos[i] = make_float4(verts[3 * g], verts[3 * g + 1], verts[3 * g + 2], 0.0f);
        lverts[3 * i] = verts[3 * g]; lverts[3 * i + 1] = verts[3 * g + 1]; lverts[3 * i + 2] = verts[3 * g + 2];
    }
    if ((rc = upload(h, d.pos_pred, pos))) return rc;
    if ((rc = upload(h, d.pos_final, pos))) return rc;
    HIPCHK(h, hipMemset(d.vel, 0, std::max<size_t>(nvl, 1) * sizeof(float4)));

    // the weight the reference's P4 writes into elems.w: 1.0 / texture(invRestVolume).x, in f32
    // (SoftbodyGPU.js:220,259-262 with invRestVolume = fround(1/V), :582-589)
    auto rest_weight = [&](uint32_t local_tet) {
        const uint32_t ge = h->partitioned ? static_cast<uint32_t>(l2g_t[local_tet]) : local_tet;
        return 1.0f / pj_inv_rest_volume(verts, &tets[4 * ge]);
    };

    h->blocked = h->fast && !(o.flags & TETSIM_FLAG_GATHER_FORMULATION);
    if ((o.flags & TETSIM_FLAG_CONSTANT_REST_SHAPE) && !h->blocked)
        return fail(h, TETSIM_EINVAL, "TETSIM_FLAG_CONSTANT_REST_SHAPE needs POLAR_JACOBI + TETSIM_FAST without TETSIM_FLAG_GATHER_FORMULATION");
    if (h->blocked) {
        BlockPlan B;
        build_blocks(lverts.data(), ltets.data(), ntl, nvl, nvo, inc, &B);
        h->tet_perm = B.tet_perm;
        PJBlk& k = h->blk;
        h->interior_tets = B.blk_tet_off[B.num_interior_blocks];
        k.nb = B.num_blocks; k.nb_interior = B.num_interior_blocks; k.nt = ntl; k.nv_local = nvl; k.nv_owned = nvo; k.nv_boundary = nvb;
        k.pos_pred = d.pos_pred; k.pos_final = d.pos_final; k.vel = d.vel; k.params = h->d_params;
        k.lean = (o.flags & TETSIM_FLAG_CONSTANT_REST_SHAPE) != 0;
        uint32_t *bto, *bvo, *lcr, *vpe;
        int32_t* bv;
        uchar4* lidx;
        float* vol;
        uint2* lce;
        const size_t nslots = B.blk_verts.size();
        if ((rc = dev_alloc(h, &bto, B.blk_tet_off.size()))) return rc;
        if ((rc = dev_alloc(h, &bvo, B.blk_vert_off.size()))) return rc;
        if ((rc = dev_alloc(h, &bv, nslots))) return rc;
        if (ntl >= kStoreWtMaxIndex || nslots >= kStoreWtMaxIndex || nvl >= kStoreWtMaxIndex)
            return fail(h, TETSIM_EINVAL, "body too large for one handle (2^27 tets / particles / partial sums: 32-bit store offsets, dev_store.h); partition it");
        if ((rc = dev_alloc(h, &lidx, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.rest_a, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.rest_b, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.rest_c, ntl))) return rc;
        if ((rc = dev_alloc(h, &vol, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.quat, ntl))) return rc;
        if ((rc = dev_alloc(h, &lcr, nslots))) return rc;
        if ((rc = dev_alloc(h, &lce, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.partial, nslots))) return rc;
        if ((rc = dev_alloc(h, &vpe, B.vp_ell.size()))) return rc;
        std::vector<float4> ra(ntl), rb(ntl), rcv(ntl), quat(ntl, make_float4(0, 0, 0, 1));
        std::vector<float> volh(ntl);
        std::vector<uchar4> lidxh(ntl);
        std::vector<uint2> lceh(ntl);
        for (uint32_t i = 0; i < ntl; i++) {
            const uint32_t lt = static_cast<uint32_t>(B.tet_perm[i]);
            const int32_t* c = &ltets[4 * lt];
            const float4 p0 = pos[c[0]], p1 = pos[c[1]], p2 = pos[c[2]], p3 = pos[c[3]];
            float4 r0 = p0, r1 = p1, r2 = p2, r3 = p3;
            if (k.lean) {  // centred rest shape, with the arithmetic the kernel would use (f32, same association)
                const float cx = (((p0.x + p1.x) + p2.x) + p3.x) * 0.25f, cy = (((p0.y + p1.y) + p2.y) + p3.y) * 0.25f,
                            cz = (((p0.z + p1.z) + p2.z) + p3.z) * 0.25f;
                r0 = make_float4(p0.x - cx, p0.y - cy, p0.z - cz, 0.0f); r1 = make_float4(p1.x - cx, p1.y - cy, p1.z - cz, 0.0f);
                r2 = make_float4(p2.x - cx, p2.y - cy, p2.z - cz, 0.0f); r3 = make_float4(p3.x - cx, p3.y - cy, p3.z - cz, 0.0f);
            }
            ra[i] = make_float4(r0.x, r0.y, r0.z, r1.x);
            rb[i] = make_float4(r1.y, r1.z, r2.x, r2.y);
            rcv[i] = make_float4(r2.z, r3.x, r3.y, r3.z);
            volh[i] = rest_weight(lt);
            lidxh[i] = make_uchar4(B.tet_lidx[4ull * i], B.tet_lidx[4ull * i + 1], B.tet_lidx[4ull * i + 2], B.tet_lidx[4ull * i + 3]);
            const uint16_t* en = &B.lc_ent[4ull * i];
            lceh[i] = make_uint2(en[0] | (static_cast<uint32_t>(en[1]) << 16), en[2] | (static_cast<uint32_t>(en[3]) << 16));
        }
        if ((rc = upload(h, bto, B.blk_tet_off))) return rc;
        if ((rc = upload(h, bvo, B.blk_vert_off))) return rc;
        if ((rc = upload(h, bv, B.blk_verts))) return rc;
        if ((rc = upload(h, lidx, lidxh))) return rc;
        if ((rc = upload(h, k.rest_a, ra))) return rc;
        if ((rc = upload(h, k.rest_b, rb))) return rc;
        if ((rc = upload(h, k.rest_c, rcv))) return rc;
        if ((rc = upload(h, vol, volh))) return rc;
        if ((rc = upload(h, k.quat, quat))) return rc;
        if ((rc = upload(h, lcr, B.lc_range))) return rc;
        if ((rc = upload(h, lce, lceh))) return rc;
        if ((rc = upload(h, vpe, B.vp_ell))) return rc;
        HIPCHK(h, hipMemset(k.partial, 0, std::max<size_t>(nslots, 1) * sizeof(float4)));
        k.blk_tet_off = bto; k.blk_vert_off = bvo; k.blk_verts = bv; k.tet_lidx = lidx; k.vol = vol;
        k.lc_range = lcr; k.lc_ent = lce; k.vp_ell = vpe; k.vp_cols = B.max_partials; k.nv_pad = B.nv_pad;
        d.quat = k.quat;  // tetsim_read_quats
        if (getenv("TETSIM_DEBUG_TRACE")) {  // development: per-tile phase timestamps of the LAST tet-kernel launch
            if ((rc = dev_alloc(h, &k.trace, 8ull * B.num_blocks))) return rc;
            HIPCHK(h, hipMemset(k.trace, 0, 8ull * B.num_blocks * sizeof(unsigned long long)));
        }
    } else {
        if ((rc = dev_alloc(h, &d.tet_idx, ntl))) return rc;
        if ((rc = dev_alloc(h, &d.elem, 4ull * d.nt_pad))) return rc;
        if ((rc = dev_alloc(h, &d.quat, ntl))) return rc;
        if ((rc = dev_alloc(h, &d.slot_tab, static_cast<size_t>(std::max(maxv, 1u)) * d.nv_pad))) return rc;
        if ((rc = dev_alloc(h, &d.slot_cnt, d.nv_pad))) return rc;
        std::vector<int4> idx(ntl);
        std::vector<float4> elem(4ull * d.nt_pad, make_float4(0, 0, 0, 0)), quat(ntl, make_float4(0, 0, 0, 1));
        for (uint32_t e = 0; e < ntl; e++) {
            const int32_t* lt = &ltets[4 * e];
            idx[e] = make_int4(lt[0], lt[1], lt[2], lt[3]);
            const float w = rest_weight(e);
            for (int k = 0; k < 4; k++) {
                const float4 p = pos[lt[k]];
                elem[static_cast<size_t>(k) * d.nt_pad + e] = make_float4(p.x, p.y, p.z, w);
            }
        }
        if ((rc = upload(h, d.tet_idx, idx))) return rc;
        if ((rc = upload(h, d.elem, elem))) return rc;
        if ((rc = upload(h, d.quat, quat))) return rc;

        std::vector<int32_t> tab(static_cast<size_t>(std::max(maxv, 1u)) * d.nv_pad, 0);
        std::vector<uint32_t> cnt(d.nv_pad, 0);
        for (uint32_t v = 0; v < nvo; v++) {
            const uint32_t c = inc.offset[v + 1] - inc.offset[v];
            cnt[v] = c;
            for (uint32_t sl = 0; sl < c; sl++) {
                const int32_t enc = inc.slot[inc.offset[v] + sl];
                tab[static_cast<size_t>(sl) * d.nv_pad + v] = static_cast<int32_t>((enc & 3) * d.nt_pad + (enc >> 2));
            }
        }
        if ((rc = upload(h, d.slot_tab, tab))) return rc;
        if ((rc = upload(h, d.slot_cnt, cnt))) return rc;
    }

    if (h->partitioned) {
        for (const auto& nb : h->part.neigh) {
            NeighDev nd;
            nd.rank = nb.rank;
            nd.send_count = static_cast<uint32_t>(nb.send_local.size());
            nd.recv_start = nb.recv_start;
            nd.recv_count = nb.recv_count;
            nd.contiguous = nb.send_contiguous;
            nd.send_first = nd.send_count ? static_cast<uint32_t>(nb.send_local[0]) : 0;
            nd.send_global = nb.send_global;
            nd.recv_global = nb.recv_global;
            nd.send_local = nb.send_local;
            if (nd.send_count) {  // staging is always available (tetsim_halo_export, non-contiguous sends)
                if ((rc = dev_alloc(h, &nd.send_idx, nd.send_count))) return rc;
                if ((rc = dev_alloc(h, &nd.send_buf, nd.send_count))) return rc;
                if ((rc = upload(h, nd.send_idx, nb.send_local))) return rc;
            }
            h->neigh.push_back(std::move(nd));
        }
        h->info.num_neighbours = static_cast<uint32_t>(h->neigh.size());
    }
    return 0;
}

int create_neohookean(tetsim_body* h, const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt) {
    const TetSimOptions& o = h->opt;
    if (o.part_count > 1) return fail(h, TETSIM_EINVAL, "NEOHOOKEAN_GS does not partition (one halo per colour would be needed); use POLAR_JACOBI");
    // 1. element order
    std::vector<int32_t> pre(nt);
    for (uint32_t e = 0; e < nt; e++) pre[e] = static_cast<int32_t>(e);
    if (o.order == TETSIM_ORDER_COLOURED) {
        std::vector<int32_t> colour(nt);
        if (h->tet_colour.size() == nt) colour = h->tet_colour;  // caller-supplied (TetSimOptions.tet_colour)
        else prep_colours(tets, nt, nv, colour.data());
        std::stable_sort(pre.begin(), pre.end(), [&](int32_t a, int32_t b) { return colour[a] < colour[b]; });
    }
    ClusterPlan plan;
    const bool clustered = o.order == TETSIM_ORDER_CLUSTERED;
    if (clustered) {
        plan = prep_clusters(tets, nt, nv);
        pre = plan.pre;
    }
    std::vector<int32_t> ptets(4ull * nt);
    for (uint32_t i = 0; i < nt; i++) std::memcpy(&ptets[4 * i], &tets[4 * pre[i]], 4 * sizeof(int32_t));
    // 2. rest data in the order the reference would see (mass accumulation is order dependent, Softbody.js:74-78)
    std::vector<float> irp(9ull * nt), irv(nt);
    h->h_inv_mass.assign(nv, 0.0f);
    prep_rest(verts, nv, ptets.data(), nt, o.density, h->h_inv_mass.data(), irp.data(), irv.data());
    // 3. dependency levels of that order; solve order = stable sort by level
    std::vector<int32_t> pos_in(nt);
    uint32_t nl = 0;
    if (clustered) {  // the plan IS the schedule: one launch per cluster colour, storage order = step after step
        nl = static_cast<uint32_t>(plan.launch_off.size() - 1);
        for (uint32_t i = 0; i < nt; i++) pos_in[i] = static_cast<int32_t>(plan.exec_pos[i]);
        h->level_off = plan.launch_off;
    } else {
        std::vector<int32_t> level(nt);
        nl = prep_levels(ptets.data(), nt, nv, level.data());
        for (uint32_t i = 0; i < nt; i++) pos_in[i] = static_cast<int32_t>(i);
        std::stable_sort(pos_in.begin(), pos_in.end(), [&](int32_t a, int32_t b) { return level[a] < level[b]; });
        h->level_off.assign(nl + 1, 0);
        for (uint32_t i = 0; i < nt; i++) h->level_off[level[i] + 1]++;
        for (uint32_t l = 0; l < nl; l++) h->level_off[l + 1] += h->level_off[l];
    }
    h->order.resize(nt);  // solve position -> caller's tet id, for the PERMUTED sequence the reference must be fed
    std::vector<int32_t> seq(nt);
    for (uint32_t i = 0; i < nt; i++) seq[i] = pre[i];
    h->order = seq;  // tetsim_get_tet_order: the sequential order whose result we reproduce
    h->info.num_levels = nl;
    h->info.owned_particles = h->info.local_particles = nv;
    h->info.local_elems = h->info.owned_elems = nt;

    if (nv >= kStoreWtMaxIndex) return fail(h, TETSIM_EINVAL, "more than 2^27 particles (32-bit store offsets, dev_store.h)");
    NHDev& d = h->nh;
    d.nv = nv; d.nt = nt;
    int rc;
    if ((rc = dev_alloc(h, &d.pos, nv))) return rc;
    if ((rc = dev_alloc(h, &d.prev, nv))) return rc;
    if ((rc = dev_alloc(h, &d.vel, nv))) return rc;
    if ((rc = dev_alloc(h, &d.tet_idx, nt))) return rc;
    if ((rc = dev_alloc(h, &d.irp_a, nt))) return rc;
    if ((rc = dev_alloc(h, &d.irp_b, nt))) return rc;
    if ((rc = dev_alloc(h, &d.irp_c, nt))) return rc;
    if ((rc = dev_alloc(h, &d.vol_err, nt))) return rc;
    if ((rc = dev_alloc(h, &d.order, nt))) return rc;
    d.params = h->d_params;

    std::vector<float4> pos(nv);
    for (uint32_t i = 0; i < nv; i++) pos[i] = make_float4(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2], h->h_inv_mass[i]);
    if ((rc = upload(h, d.pos, pos))) return rc;
    if ((rc = upload(h, d.prev, pos))) return rc;
    HIPCHK(h, hipMemset(d.vel, 0, std::max<size_t>(nv, 1) * sizeof(float4)));
    HIPCHK(h, hipMemset(d.vol_err, 0, std::max<size_t>(nt, 1) * sizeof(double)));
    std::vector<int4> idx(nt);
    std::vector<float4> a(nt), b(nt), c(nt);
    std::vector<int32_t> ord(nt);
    for (uint32_t s = 0; s < nt; s++) {
        const uint32_t i = static_cast<uint32_t>(pos_in[s]);  // position in the permuted sequential order
        const int32_t* t = &ptets[4 * i];
        idx[s] = make_int4(t[0], t[1], t[2], t[3]);
        const float* m = &irp[9 * i];
        a[s] = make_float4(m[0], m[1], m[2], m[3]);
        b[s] = make_float4(m[4], m[5], m[6], m[7]);
        c[s] = make_float4(m[8], irv[i], 0.0f, 0.0f);
        ord[s] = static_cast<int32_t>(i);  // vol_err is indexed by sequential position
    }
    if ((rc = upload(h, d.tet_idx, idx))) return rc;
    if ((rc = upload(h, d.irp_a, a))) return rc;
    if ((rc = upload(h, d.irp_b, b))) return rc;
    if ((rc = upload(h, d.irp_c, c))) return rc;
    if ((rc = upload(h, d.order, ord))) return rc;
    if (clustered) {
        if ((rc = dev_alloc(h, &d.corner_slots, nt))) return rc;
        if ((rc = upload(h, d.corner_slots, plan.corner_slots))) return rc;
        if ((rc = dev_alloc(h, &h->d_slot_vid, plan.slot_vid.size()))) return rc;
        if ((rc = upload(h, h->d_slot_vid, plan.slot_vid))) return rc;
        for (uint32_t l = 0; l < nl; l++) {
            NHClusterLaunch L;
            L.nsteps = plan.step_off[l + 1] - plan.step_off[l];
            for (uint32_t j = 0; j < L.nsteps; j++) {
                L.first[j] = plan.step_first[plan.step_off[l] + j];
                L.count[j] = plan.step_count[plan.step_off[l] + j];
            }
            L.clusters = L.nsteps ? L.count[0] : 0;
            L.slot_vid = h->d_slot_vid + plan.vid_off[l];
            h->cluster_launch.push_back(L);
        }
    }
    return 0;
}

}  // namespace

// =============================================================================================================
extern "C" {

int tetsim_abi_version(void) { return TETSIM_ABI_VERSION; }

// ---- .tetsim mesh container ---------------------------------------------------------------------------------------
struct tetsim_mesh_file { tetsim::MeshFile* m; };

int tetsim_mesh_write(const char* path, const TetSimMeshArrays* a) {
    if (!a) return fail(nullptr, TETSIM_EINVAL, "arrays is null");
    const std::string e = mesh_write(path, *a);
    return e.empty() ? TETSIM_OK : fail(nullptr, TETSIM_EINVAL, e);
}
int tetsim_mesh_open(const char* path, tetsim_mesh* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "out is null");
    *out = nullptr;
    tetsim::MeshFile* m = nullptr;
    const std::string e = mesh_open(path, &m);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    *out = new tetsim_mesh_file{m};
    return TETSIM_OK;
}
int tetsim_mesh_arrays(tetsim_mesh m, TetSimMeshArrays* out) {
    if (!m || !out) return fail(nullptr, TETSIM_EINVAL, "null argument");
    *out = mesh_arrays(m->m);
    return TETSIM_OK;
}
int tetsim_mesh_close(tetsim_mesh m) {
    if (!m) return TETSIM_OK;
    mesh_close(m->m);
    delete m;
    return TETSIM_OK;
}
int tetsim_create_from_file(const char* path, const TetSimOptions* opts, tetsim_handle* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "out handle pointer is null");
    *out = nullptr;
    tetsim::MeshFile* m = nullptr;
    const std::string e = mesh_open(path, &m);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    const TetSimMeshArrays& a = mesh_arrays(m);
    TetSimOptions o;
    if (opts) o = *opts; else tetsim_default_options(&o);
    if (!o.tet_colour && a.tet_colour && o.solver == TETSIM_SOLVER_NEOHOOKEAN_GS && o.order == TETSIM_ORDER_COLOURED) o.tet_colour = a.tet_colour;
    if (o.part_count > 1 && !o.vert_owner && a.vert_owner) {
        if (static_cast<uint32_t>(o.part_count) != a.part_count) {
            mesh_close(m);
            return fail(nullptr, TETSIM_EINVAL, std::string(path) + ": stored partition map is for " + std::to_string(a.part_count) + " parts, " + std::to_string(o.part_count) + " requested");
        }
        o.vert_owner = a.vert_owner;
    }
    int rc = tetsim_create(a.verts, a.num_particles, a.tets, a.num_elems, &o, out);
    if (rc == TETSIM_OK && a.vis_verts && a.num_vis_verts && o.part_count <= 1) {
        rc = tetsim_set_visual_mesh(*out, a.vis_verts, a.num_vis_verts, nullptr);
        if (rc != TETSIM_OK) { g_create_error = (*out)->err; tetsim_destroy(*out); *out = nullptr; }
    }
    mesh_close(m);  // create copied what it keeps
    return rc;
}

void tetsim_default_options(TetSimOptions* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->solver = TETSIM_SOLVER_POLAR_JACOBI;
    o->precision = TETSIM_PRECISE;
    o->order = TETSIM_ORDER_ORIGINAL;
    o->flags = TETSIM_FLAG_REF_SLOT_TABLE | TETSIM_FLAG_REF_FIXED_BOUNDS;
    o->device = 0;
    o->density = 1000.0;
    o->part_count = 1;
    o->part_index = 0;
    o->vert_owner = nullptr;
    o->tet_colour = nullptr;
}

void tetsim_default_params(TetSimParams* p) {  // main.js:22-36
    if (!p) return;
    p->gravity = -9.81;
    p->friction = 1000.0;
    p->devCompliance = 1.0 / 100000.0;
    p->volCompliance = 0.0;
    const double wb[6] = {-2.5, -1.0, -2.5, 2.5, 10.0, 2.5};
    std::memcpy(p->worldBounds, wb, sizeof(wb));
}

const char* tetsim_last_error(tetsim_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tetsim_create(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, const TetSimOptions* opts, tetsim_handle* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "out handle pointer is null");
    *out = nullptr;
    TetSimOptions o;
    if (opts) o = *opts; else tetsim_default_options(&o);
    if (o.solver != TETSIM_SOLVER_POLAR_JACOBI && o.solver != TETSIM_SOLVER_NEOHOOKEAN_GS) return fail(nullptr, TETSIM_EINVAL, "unknown solver");
    if (o.precision != TETSIM_PRECISE && o.precision != TETSIM_FAST) return fail(nullptr, TETSIM_EINVAL, "unknown precision");
    if (o.solver == TETSIM_SOLVER_NEOHOOKEAN_GS && (o.order < TETSIM_ORDER_ORIGINAL || o.order > TETSIM_ORDER_CLUSTERED)) return fail(nullptr, TETSIM_EINVAL, "unknown order");
    if (o.part_count < 1) o.part_count = 1;
    std::string merr = validate_mesh(verts, nv, tets, nt, o.solver == TETSIM_SOLVER_NEOHOOKEAN_GS);
    if (!merr.empty()) return fail(nullptr, TETSIM_EINVAL, merr);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, TETSIM_ENODEVICE, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                                                   "); libtetsim_hip has no CPU fallback");
    if (o.device < 0 || o.device >= ndev) return fail(nullptr, TETSIM_ENODEVICE, "device ordinal out of range");
    if ((o.flags & TETSIM_FLAG_CONSTANT_REST_SHAPE) && o.solver != TETSIM_SOLVER_POLAR_JACOBI)
        return fail(nullptr, TETSIM_EINVAL, "TETSIM_FLAG_CONSTANT_REST_SHAPE applies to TETSIM_SOLVER_POLAR_JACOBI only");

    tetsim_body* h = new tetsim_body();
    h->opt = o;
    h->opt.vert_owner = nullptr;  // not retained
    if (o.tet_colour && nt) h->tet_colour.assign(o.tet_colour, o.tet_colour + nt);
    h->opt.tet_colour = nullptr;
    h->fast = o.precision == TETSIM_FAST;
    h->info.num_particles = nv;
    h->info.num_elems = nt;
    h->info.solver = o.solver; h->info.precision = o.precision; h->info.order = o.order; h->info.device = o.device; h->info.flags = o.flags;
    h->h_verts.assign(verts, verts + 3ull * nv);
    h->h_tets.assign(tets, tets + 4ull * nt);

    auto bail = [&](int rc) { g_create_error = h->err; tetsim_destroy(h); return rc; };
    auto hipok = [&](hipError_t er, const char* what) { if (er != hipSuccess) { h->err = std::string(what) + ": " + hipGetErrorString(er); return false; } return true; };
    if (!hipok(hipSetDevice(o.device), "hipSetDevice")) return bail(TETSIM_EHIP);
    if (!hipok(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate")) return bail(TETSIM_EHIP);
    if (!hipok(hipEventCreate(&h->ev_a), "hipEventCreate") || !hipok(hipEventCreate(&h->ev_b), "hipEventCreate")) return bail(TETSIM_EHIP);
    for (int i = 0; i < 2; i++)
        if (!hipok(hipEventCreateWithFlags(&h->ev_boundary2[i], hipEventDisableTiming), "hipEventCreate") ||
            !hipok(hipEventCreateWithFlags(&h->ev_packed2[i], hipEventDisableTiming), "hipEventCreate") ||
            !hipok(hipEventCreateWithFlags(&h->ev_sent2[i], hipEventDisableTiming), "hipEventCreate")) return bail(TETSIM_EHIP);
    if (!hipok(hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming), "hipEventCreate")) return bail(TETSIM_EHIP);
    for (int i = 0; i < kRing; i++)
        if (!hipok(hipEventCreateWithFlags(&h->ring_ev[i], hipEventDisableTiming), "hipEventCreate")) return bail(TETSIM_EHIP);
    if (!hipok(hipHostMalloc(reinterpret_cast<void**>(&h->h_ring), sizeof(DevParams) * kRing, hipHostMallocDefault), "hipHostMalloc")) return bail(TETSIM_EHIP);
    { int rc = dev_alloc(h, &h->d_params, 1); if (rc) return bail(rc); }

    TetSimOptions with_owner = o;  // vert_owner is only read during construction
    h->opt.vert_owner = with_owner.vert_owner;
    int rc = o.solver == TETSIM_SOLVER_POLAR_JACOBI ? create_polar(h, verts, nv, tets, nt) : create_neohookean(h, verts, nv, tets, nt);
    h->opt.vert_owner = nullptr;
    if (rc) return bail(rc);
    if (!hipok(hipDeviceSynchronize(), "hipDeviceSynchronize")) return bail(TETSIM_EHIP);
    *out = h;
    return TETSIM_OK;
}

void tetsim_destroy(tetsim_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->opt.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    if (h->blk.trace && getenv("TETSIM_DEBUG_TRACE")) {
        std::vector<unsigned long long> tr(8ull * h->blk.nb);
        if (hipMemcpy(tr.data(), h->blk.trace, tr.size() * sizeof(tr[0]), hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(getenv("TETSIM_DEBUG_TRACE"), "wb")) { fwrite(tr.data(), sizeof(tr[0]), tr.size(), f); fclose(f); }
    }
    // graphs first: a captured halo graph holds RCCL work, and ncclCommDestroy waits for (hangs on) captured work that still exists
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->h_ring) (void)hipHostFree(h->h_ring);
    if (h->pinned_pos) (void)hipHostFree(h->pinned_pos);
    for (int i = 0; i < kRing; i++) if (h->ring_ev[i]) (void)hipEventDestroy(h->ring_ev[i]);
    for (hipEvent_t ev : {h->ev_a, h->ev_b, h->ev_halo, h->ev_boundary2[0], h->ev_boundary2[1], h->ev_packed2[0], h->ev_packed2[1],
                          h->ev_sent2[0], h->ev_sent2[1]}) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {h->ev_fork, h->ev_bnd_tet}) if (ev) (void)hipEventDestroy(ev);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int tetsim_get_info(tetsim_handle h, TetSimInfo* info) {
    if (!h || !info) return fail(h, TETSIM_EINVAL, "null argument");
    *info = h->info;
    return 0;
}

int tetsim_step(tetsim_handle h, double dt, const TetSimParams* params) {
    if (!h) return TETSIM_EINVAL;
    if (!h->group.empty()) return fail(h, TETSIM_ESTATE, "this body belongs to an in-process group: step it with tetsim_group_step_n");
    HIPCHK(h, hipSetDevice(h->opt.device));
    int rc = push_params(h, dt, params);
    if (rc) return rc;
    if ((rc = ensure_prediction(h, dt))) return rc;
    return enqueue_substep(h);
}

int tetsim_step_n(tetsim_handle h, uint32_t n, double dt, const TetSimParams* params) {
    if (!h) return TETSIM_EINVAL;
    if (!h->group.empty()) return fail(h, TETSIM_ESTATE, "this body belongs to an in-process group: step it with tetsim_group_step_n");
    if (n == 0) return 0;
    HIPCHK(h, hipSetDevice(h->opt.device));
    int rc = push_params(h, dt, params);
    if (rc) return rc;
    if ((rc = ensure_prediction(h, dt))) return rc;
    if (has_transport(h)) {
        // RCCL bodies: the first call runs eagerly (RCCL sets its connections up on first use, which must not happen inside a
        // capture); afterwards the n substeps -- both streams, the grouped send/recv included -- are one captured graph:
        // eager cross-stream dependencies cost ~10 us each on this stack and there are three per substep on the halo's
        // critical path.  TETSIM_HALO_GRAPH=0 keeps everything eager.
        static const bool use_graph = [] { const char* e = getenv("TETSIM_HALO_GRAPH"); return !(e && e[0] == '0'); }();
        if (!h->comm || !use_graph || !h->halo_warm || h->halo_graph_broken || uses_flag_sync(h)) {  // (the flag path is eager by design)
            for (uint32_t i = 0; i < n && !rc; i++) rc = enqueue_substep(h);
            h->halo_warm = true;
            return rc;
        }
        if (h->halo_pending) {  // leftovers of eager calls: join them first
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_sent2[h->halo_parity ^ 1u], 0));
            h->halo_pending = false;
        }
    }
    auto it = h->graphs.find(n);
    if (it == h->graphs.end()) {
        hipGraphExec_t exec = nullptr;
        if ((rc = build_graph(h, n, &exec))) {
            if (!has_transport(h)) return rc;
            // a halo graph that cannot be built (an RCCL build that refuses capture): stay eager for good, loudly
            fprintf(stderr, "[tetsim] halo graph capture failed (%s); falling back to eager halo stepping\n", h->err.c_str());
            h->halo_graph_broken = true;
            (void)hipGetLastError();
            rc = 0;
            for (uint32_t i = 0; i < n && !rc; i++) rc = enqueue_substep(h);
            return rc;
        }
        it = h->graphs.emplace(n, exec).first;
    }
    HIPCHK(h, hipGraphLaunch(it->second, h->stream));
    return 0;
}

int tetsim_sync(tetsim_handle h) {
    if (!h) return TETSIM_EINVAL;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    if (h->d_sync) {  // a bounded device-side wait that gave up (util_kernels.hip): the results since then are not to be trusted
        uint32_t err = 0;
        HIPCHK(h, hipMemcpy(&err, h->d_sync + 4, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(h, TETSIM_ECOMM, "a halo dependency was not signalled within 2 s (device-side wait timed out): a rank or a queue is stuck");
    }
    return 0;
}

int tetsim_read_positions(tetsim_handle h, float* out) {
    if (!h) return TETSIM_EINVAL;
    return h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI ? read_float4_as_xyz(h, h->pj.pos_final, h->pj.nv_owned, out)
                                                       : read_float4_as_xyz(h, h->nh.pos, h->nh.nv, out);
}
namespace {
const float4* current_positions(tetsim_body* h) { return h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI ? h->pj.pos_final : h->nh.pos; }
int ensure_index_map(tetsim_body* h) {  // internal Morton numbering -> API numbering, on the device
    if (h->d_api2dev || h->api2dev.empty()) return 0;
    int rc = dev_alloc(h, &h->d_api2dev, h->api2dev.size());
    if (rc) return rc;
    return upload(h, h->d_api2dev, h->api2dev);
}
}  // namespace

int tetsim_read_positions_pinned(tetsim_handle h, const float** out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const uint32_t n = h->info.owned_particles;
    int rc;
    if (!h->pinned_pos) {
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pinned_pos), std::max<size_t>(3ull * n, 1) * sizeof(float), hipHostMallocDefault));
        if ((rc = dev_alloc(h, &h->d_packed, 3ull * n))) return rc;
        if ((rc = ensure_index_map(h))) return rc;
    }
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    util_launch_pack_xyz(h->stream, current_positions(h), h->d_api2dev, h->d_packed, n);
    if (n) HIPCHK(h, hipMemcpyAsync(h->pinned_pos, h->d_packed, 3ull * n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *out = h->pinned_pos;
    return 0;
}

int tetsim_read_prev_positions(tetsim_handle h, float* out) {
    if (!h) return TETSIM_EINVAL;
    if (h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI)
        return fail(h, TETSIM_ESTATE, "POLAR_JACOBI does not keep prevPos after a substep (it equals the previous read_positions)");
    return read_float4_as_xyz(h, h->nh.prev, h->nh.nv, out);
}
int tetsim_read_velocities(tetsim_handle h, float* out) {
    if (!h) return TETSIM_EINVAL;
    return h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI ? read_float4_as_xyz(h, h->pj.vel, h->pj.nv_owned, out)
                                                       : read_float4_as_xyz(h, h->nh.vel, h->nh.nv, out);
}
int tetsim_read_quats(tetsim_handle h, float* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return fail(h, TETSIM_ESTATE, "quaternions exist only for POLAR_JACOBI");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->pj.nt) HIPCHK(h, hipMemcpy(out, h->pj.quat, h->pj.nt * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
int tetsim_read_vol_error(tetsim_handle h, double* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver != TETSIM_SOLVER_NEOHOOKEAN_GS) return fail(h, TETSIM_ESTATE, "volError exists only for NEOHOOKEAN_GS");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<double> ve(h->nh.nt);
    if (h->nh.nt) HIPCHK(h, hipMemcpy(ve.data(), h->nh.vol_err, h->nh.nt * sizeof(double), hipMemcpyDeviceToHost));
    double s = 0.0;  // Softbody.js:163 accumulates in element order; :209 divides by numElems
    for (double v : ve) s += v;
    *out = s / static_cast<double>(h->nh.nt);
    return 0;
}
int tetsim_write_state(tetsim_handle h, const float* pos, const float* vel) {
    if (!h || !pos || !vel) return fail(h, TETSIM_EINVAL, "null argument");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    const uint32_t n = pjs ? h->pj.nv_owned : h->nh.nv;
    std::vector<float4> p(n), v(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t dv = (pjs && !h->api2dev.empty()) ? h->api2dev[i] : i;
        p[dv] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], pjs ? 0.0f : h->h_inv_mass[i]);
        v[dv] = make_float4(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2], 0.0f);
    }
    if (pjs) {
        if (h->partitioned && !h->neigh.empty()) return fail(h, TETSIM_ESTATE, "write_state is not supported on partitioned bodies");
        if (n) { HIPCHK(h, hipMemcpy(h->pj.pos_final, p.data(), n * sizeof(float4), hipMemcpyHostToDevice));
                 HIPCHK(h, hipMemcpy(h->pj.pos_pred, p.data(), n * sizeof(float4), hipMemcpyHostToDevice));
                 HIPCHK(h, hipMemcpy(h->pj.vel, v.data(), n * sizeof(float4), hipMemcpyHostToDevice)); }
        h->pred_any_dt = false;
        h->dt_pred = std::nanf("");  // forces a re-prediction at the next step
    } else if (n) {
        HIPCHK(h, hipMemcpy(h->nh.pos, p.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->nh.vel, v.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    }
    return 0;
}

int tetsim_get_owned_ids(tetsim_handle h, int32_t* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    const uint32_t n = h->info.owned_particles;
    for (uint32_t i = 0; i < n; i++) out[i] = h->partitioned ? h->part.local_to_global_vert[i] : static_cast<int32_t>(i);
    return 0;
}
int tetsim_get_local_tets(tetsim_handle h, int32_t* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    const uint32_t n = h->info.local_elems;
    for (uint32_t i = 0; i < n; i++) {
        const int32_t lt = h->blocked ? h->tet_perm[i] : static_cast<int32_t>(i);  // blocked: tets live in tile order
        out[i] = h->partitioned ? h->part.local_to_global_tet[lt] : lt;
    }
    return 0;
}
int tetsim_get_tet_order(tetsim_handle h, int32_t* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver != TETSIM_SOLVER_NEOHOOKEAN_GS) return fail(h, TETSIM_ESTATE, "tet order exists only for NEOHOOKEAN_GS");
    std::copy(h->order.begin(), h->order.end(), out);
    return 0;
}
int tetsim_get_level_offsets(tetsim_handle h, int32_t* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver != TETSIM_SOLVER_NEOHOOKEAN_GS) return fail(h, TETSIM_ESTATE, "levels exist only for NEOHOOKEAN_GS");
    for (size_t i = 0; i < h->level_off.size(); i++) out[i] = static_cast<int32_t>(h->level_off[i]);
    return 0;
}
int tetsim_read_inv_mass(tetsim_handle h, float* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver == TETSIM_SOLVER_NEOHOOKEAN_GS) { std::copy(h->h_inv_mass.begin(), h->h_inv_mass.end(), out); return 0; }
    const uint32_t nv = h->info.num_particles, nt = h->info.num_elems;
    std::vector<float> irp(9ull * nt), irv(nt);
    prep_rest(h->h_verts.data(), nv, h->h_tets.data(), nt, h->opt.density, out, irp.data(), irv.data());
    return 0;
}

int tetsim_set_visual_mesh(tetsim_handle h, const float* vis_verts, uint32_t nvis, const float* rest_normals) {
    if (!h || (nvis && !vis_verts)) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->partitioned) return fail(h, TETSIM_ESTATE, "visual meshes are supported on unpartitioned bodies only");
    if (h->skin.nvis) return fail(h, TETSIM_ESTATE, "a visual mesh is already attached");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    const uint32_t nt = h->info.num_elems;
    std::vector<int32_t> tet_pos;  // caller's tet id -> device tet position (quaternion index)
    if (pjs) {
        tet_pos.resize(nt);
        for (uint32_t i = 0; i < nt; i++) tet_pos[h->blocked ? h->tet_perm[i] : i] = static_cast<int32_t>(i);
    }
    std::vector<int4> corner(nvis);
    std::vector<float4> weight(nvis), n0(nvis);
    std::vector<int32_t> qidx(nvis, 0);
    for (uint32_t i = 0; i < nvis; i++) {
        const float tn = vis_verts[4 * i];
        if (!(tn >= 0.0f) || tn >= static_cast<float>(nt) || tn != std::floor(tn)) return fail(h, TETSIM_EINVAL, "visual vertex " + std::to_string(i) + " references a tet outside the mesh");
        const uint32_t e = static_cast<uint32_t>(tn);
        int32_t c[4];
        for (int k = 0; k < 4; k++) {
            const int32_t v = h->h_tets[4 * e + k];
            c[k] = (pjs && !h->api2dev.empty()) ? static_cast<int32_t>(h->api2dev[v]) : v;
        }
        corner[i] = make_int4(c[0], c[1], c[2], c[3]);
        weight[i] = make_float4(vis_verts[4 * i + 1], vis_verts[4 * i + 2], vis_verts[4 * i + 3], 0.0f);
        if (pjs) qidx[i] = tet_pos[e];
        if (rest_normals) n0[i] = make_float4(rest_normals[3 * i], rest_normals[3 * i + 1], rest_normals[3 * i + 2], 0.0f);
    }
    SkinDev& k = h->skin;
    int4* dc; float4 *dw, *dn = nullptr; int32_t* dq;
    int rc;
    if ((rc = dev_alloc(h, &dc, nvis))) return rc;
    if ((rc = dev_alloc(h, &dw, nvis))) return rc;
    if ((rc = dev_alloc(h, &dq, nvis))) return rc;
    if ((rc = dev_alloc(h, &k.out_pos, nvis))) return rc;
    if ((rc = upload(h, dc, corner))) return rc;
    if ((rc = upload(h, dw, weight))) return rc;
    if ((rc = upload(h, dq, qidx))) return rc;
    if (rest_normals && pjs) {
        if ((rc = dev_alloc(h, &dn, nvis))) return rc;
        if ((rc = dev_alloc(h, &k.out_nrm, nvis))) return rc;
        if ((rc = upload(h, dn, n0))) return rc;
    }
    k.corner = dc; k.weight = dw; k.qidx = dq; k.normal0 = dn;
    k.nvis = nvis;
    return 0;
}

int tetsim_read_visual_mesh(tetsim_handle h, float* positions_out, float* normals_out) {
    if (!h || !positions_out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->skin.nvis) return fail(h, TETSIM_ESTATE, "no visual mesh attached (tetsim_set_visual_mesh)");
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    if (normals_out && !h->skin.out_nrm) return fail(h, TETSIM_ESTATE, "normals need POLAR_JACOBI and rest normals at tetsim_set_visual_mesh");
    HIPCHK(h, hipSetDevice(h->opt.device));
    // Softbody.js arithmetic for the solver that mirrors Softbody.js, the vertex-shader arithmetic for the other
    skin_launch(h->stream, h->skin, pjs ? h->pj.pos_final : h->nh.pos, pjs ? h->pj.quat : nullptr, !pjs);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t n = h->skin.nvis;
    std::vector<float4> tmp(n);
    HIPCHK(h, hipMemcpy(tmp.data(), h->skin.out_pos, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) { positions_out[3 * i] = tmp[i].x; positions_out[3 * i + 1] = tmp[i].y; positions_out[3 * i + 2] = tmp[i].z; }
    if (normals_out) {
        HIPCHK(h, hipMemcpy(tmp.data(), h->skin.out_nrm, n * sizeof(float4), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) { normals_out[3 * i] = tmp[i].x; normals_out[3 * i + 1] = tmp[i].y; normals_out[3 * i + 2] = tmp[i].z; }
    }
    return 0;
}

int tetsim_set_grab(tetsim_handle h, int32_t id, const float xyz[3]) {
    if (!h) return TETSIM_EINVAL;
    if (id >= static_cast<int32_t>(h->info.num_particles)) return fail(h, TETSIM_EINVAL, "grab id out of range");
    h->grab_global = id < 0 ? -1 : id;
    ref_grab_texels(h->grab_global, h->info.num_elems, h->info.num_particles, h->grab_ref);
    if (xyz) std::memcpy(h->grab_pos, xyz, 3 * sizeof(float));
    return 0;
}
namespace {
// argmin of Softbody.js:279-291 over this handle's OWNED particles, on the device: one (d2, index) candidate per 256
// particles comes back.  *local receives the API-local index (first minimum), *best its squared distance (f64).
int nearest_owned(tetsim_body* h, const float xyz[3], int32_t* local, double* best_out) {
    HIPCHK(h, hipSetDevice(h->opt.device));
    const uint32_t n = h->info.owned_particles, nblk = (n + 255u) / 256u;
    int rc;
    if (!h->d_best) {
        if ((rc = dev_alloc(h, &h->d_best, nblk))) return rc;
        if ((rc = dev_alloc(h, &h->d_best_id, nblk))) return rc;
        if ((rc = ensure_index_map(h))) return rc;
    }
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    util_launch_nearest(h->stream, current_positions(h), h->d_api2dev, n, static_cast<double>(xyz[0]), static_cast<double>(xyz[1]),
                        static_cast<double>(xyz[2]), h->d_best, h->d_best_id);
    std::vector<double> bd(nblk);
    std::vector<uint32_t> bi(nblk);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (nblk) {
        HIPCHK(h, hipMemcpy(bd.data(), h->d_best, nblk * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(bi.data(), h->d_best_id, nblk * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    double best = 1.7976931348623157e308;
    int32_t id = -1;
    for (uint32_t b = 0; b < nblk; b++)  // blocks are in ascending particle order: `<` keeps the first minimum
        if (bd[b] < best) { best = bd[b]; id = static_cast<int32_t>(bi[b]); }
    *local = id;
    *best_out = best;
    return 0;
}
}  // namespace

int tetsim_start_grab(tetsim_handle h, const float xyz[3], int32_t* id_out) {
    if (!h || !xyz) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->partitioned) return fail(h, TETSIM_ESTATE, "start_grab on a partitioned body: combine tetsim_nearest_particle over the partitions on the host, then tetsim_set_grab on each");
    int32_t id = -1;
    double best = 0.0;
    int rc = nearest_owned(h, xyz, &id, &best);
    if (rc) return rc;
    h->grab_global = id;
    ref_grab_texels(h->grab_global, h->info.num_elems, h->info.num_particles, h->grab_ref);
    std::memcpy(h->grab_pos, xyz, 3 * sizeof(float));
    if (id_out) *id_out = id;
    return 0;
}

int tetsim_nearest_particle(tetsim_handle h, const float xyz[3], int32_t* global_id, double* dist2) {
    if (!h || !xyz || !global_id || !dist2) return fail(h, TETSIM_EINVAL, "null argument");
    int32_t local = -1;
    int rc = nearest_owned(h, xyz, &local, dist2);
    if (rc) return rc;
    *global_id = local < 0 ? -1 : (h->partitioned ? h->part.local_to_global_vert[local] : local);
    return 0;
}

int tetsim_profile(tetsim_handle h, uint32_t n, double dt, const TetSimParams* params, TetSimProfile* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    // a body with an RCCL halo: every rank calls this together (the substeps exchange halos as usual); what is timed is the
    // interior tet kernel and the particle kernel of the two-stream choreography
    const bool halo = has_transport(h);
    if (halo && (!h->comm || !h->blocked || h->blk.nb == h->blk.nb_interior || getenv("TETSIM_DEBUG_ONE_STREAM")))
        return fail(h, TETSIM_ESTATE, "profiling a partitioned body needs the RCCL transport and the blocked formulation (in-process groups: use rocprofv3)");
    if (halo && h->blk.nb_interior == 0)
        return fail(h, TETSIM_ESTATE, "nothing to time: this partition has no interior tiles (every tile is next to the halo)");
    HIPCHK(h, hipSetDevice(h->opt.device));
    std::memset(out, 0, sizeof(*out));
    int rc = push_params(h, dt, params);
    if (rc) return rc;
    if ((rc = ensure_prediction(h, dt))) return rc;
    out->tets_per_tet_launch = halo ? h->interior_tets : h->info.local_elems;
    // POLAR_JACOBI: every kernel carries its own begin/end events (hipExtLaunchKernelGGL), so kernel_ms is the sum of
    // the kernels' OWN durations inside the real tet -> particle -> tet ... sequence (what rocprofv3 reports), not the
    // spacing of event markers.  NEOHOOKEAN_GS: one span per kernel class (hundreds of tiny level launches).
    std::vector<hipEvent_t> ev(4ull * n + 2);
    for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    hipEvent_t first_ev = ev[4ull * n], last_ev = ev[4ull * n + 1];
    HIPCHK(h, hipEventRecord(first_ev, h->stream));
    for (uint32_t i = 0; i < n; i++) {
        if (pjs && halo) {
            if ((rc = enqueue_phase_a(h, &ev[4 * i])) || (rc = enqueue_phase_b(h))) break;
        } else if (pjs) {
            pj_tet(h, ev[4 * i], ev[4 * i + 1]);
            pj_vertex(h, 0, h->pj.nv_owned, ev[4 * i + 2], ev[4 * i + 3]);
        } else {
            HIPCHK(h, hipEventRecord(ev[4 * i], h->stream));
            h->fast ? nh_launch_predict_fast(h->stream, h->nh) : nh_launch_predict_precise(h->stream, h->nh);
            HIPCHK(h, hipEventRecord(ev[4 * i + 1], h->stream));
            nh_sweep(h);
            HIPCHK(h, hipEventRecord(ev[4 * i + 2], h->stream));
            h->fast ? nh_launch_post_fast(h->stream, h->nh) : nh_launch_post_precise(h->stream, h->nh);
            HIPCHK(h, hipEventRecord(ev[4 * i + 3], h->stream));
        }
    }
    HIPCHK(h, hipEventRecord(last_ev, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    if (rc) { for (auto& e : ev) (void)hipEventDestroy(e); return rc; }
    float ms = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        float a = 0, b = 0, c = 0;
        if (pjs) {
            HIPCHK(h, hipEventElapsedTime(&a, ev[4 * i], ev[4 * i + 1]));
            HIPCHK(h, hipEventElapsedTime(&b, ev[4 * i + 2], ev[4 * i + 3]));
            out->kernel_ms[TETSIM_K_TET] += a; out->kernel_ms[TETSIM_K_VERTEX] += b;
            out->launches[TETSIM_K_TET]++; out->launches[TETSIM_K_VERTEX]++;
        } else {
            HIPCHK(h, hipEventElapsedTime(&a, ev[4 * i], ev[4 * i + 1]));
            HIPCHK(h, hipEventElapsedTime(&b, ev[4 * i + 1], ev[4 * i + 2]));
            HIPCHK(h, hipEventElapsedTime(&c, ev[4 * i + 2], ev[4 * i + 3]));
            out->kernel_ms[TETSIM_K_VERTEX] += a + c; out->kernel_ms[TETSIM_K_TET] += b;
            out->launches[TETSIM_K_VERTEX] += 2; out->launches[TETSIM_K_TET] += static_cast<uint32_t>(h->level_off.size() - 1);
        }
    }
    HIPCHK(h, hipEventElapsedTime(&ms, first_ev, last_ev));
    out->total_ms = ms;
    out->substeps = n;
    for (auto& e : ev) (void)hipEventDestroy(e);
    return 0;
}

int tetsim_time_kernels(tetsim_handle h, uint32_t reps, double dt, const TetSimParams* params, TetSimProfile* out) {
    if (!h || !out || reps == 0) return fail(h, TETSIM_EINVAL, "bad argument");
    if (has_transport(h)) return fail(h, TETSIM_ESTATE, "time a partitioned body through rocprofv3 instead");
    HIPCHK(h, hipSetDevice(h->opt.device));
    std::memset(out, 0, sizeof(*out));
    int rc = push_params(h, dt, params);
    if (rc) return rc;
    if ((rc = ensure_prediction(h, dt))) return rc;
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    auto tet_once = [&]() {
        if (pjs) { pj_tet(h); return 1u; }
        nh_sweep(h);
        return static_cast<uint32_t>(h->level_off.size() - 1);
    };
    auto vert_once = [&]() {
        if (pjs) { pj_vertex(h, 0, h->pj.nv_owned); return 1u; }
        h->fast ? nh_launch_predict_fast(h->stream, h->nh) : nh_launch_predict_precise(h->stream, h->nh);
        h->fast ? nh_launch_post_fast(h->stream, h->nh) : nh_launch_post_precise(h->stream, h->nh);
        return 2u;
    };
    float ms = 0.0f;
    for (int which = 0; which < 2; which++) {
        (which == 0 ? tet_once() : vert_once());  // warm
        HIPCHK(h, hipEventRecord(h->ev_a, h->stream));
        uint32_t launches = 0;
        for (uint32_t r = 0; r < reps; r++) launches += which == 0 ? tet_once() : vert_once();
        HIPCHK(h, hipEventRecord(h->ev_b, h->stream));
        HIPCHK(h, hipEventSynchronize(h->ev_b));
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
        const int k = which == 0 ? TETSIM_K_TET : TETSIM_K_VERTEX;
        out->kernel_ms[k] = ms;
        out->launches[k] = launches;
        out->total_ms += ms;
    }
    out->substeps = reps;
    h->pred_any_dt = false;
    h->dt_pred = std::nanf("");  // the prediction no longer matches the state
    return 0;
}

int tetsim_time_step_n(tetsim_handle h, uint32_t n, double dt, const TetSimParams* params, double* ms_out) {
    if (!h || !ms_out) return fail(h, TETSIM_EINVAL, "null argument");
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipEventRecord(h->ev_a, h->stream));
    int rc = tetsim_step_n(h, n, dt, params);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev_b, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev_b));
    float ms = 0.0f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    *ms_out = ms;
    return 0;
}

int tetsim_measure_copy_bandwidth(int32_t device, uint64_t bytes, uint32_t reps, double* gbps_out) {
    if (!gbps_out || bytes < 16 || reps == 0) return fail(nullptr, TETSIM_EINVAL, "bad argument");
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess) { g_create_error = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, TETSIM_ENODEVICE, "no HIP device available");
    if (!chk(hipSetDevice(device), "hipSetDevice")) return TETSIM_EHIP;
    const uint64_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = TETSIM_OK;
    if (!chk(hipMalloc(reinterpret_cast<void**>(&a), n * 16), "hipMalloc") || !chk(hipMalloc(reinterpret_cast<void**>(&b), n * 16), "hipMalloc")) rc = TETSIM_ENOMEM;
    if (!rc && (!chk(hipStreamCreate(&s), "hipStreamCreate") || !chk(hipEventCreate(&e0), "hipEventCreate") || !chk(hipEventCreate(&e1), "hipEventCreate"))) rc = TETSIM_EHIP;
    if (!rc) {
        (void)hipMemsetAsync(a, 0x3c, n * 16, s);
        for (int w = 0; w < 3; w++) util_launch_copy(s, a, b, n);
        (void)hipEventRecord(e0, s);
        for (uint32_t r = 0; r < reps; r++) util_launch_copy(s, (r & 1) ? b : a, (r & 1) ? a : b, n);
        (void)hipEventRecord(e1, s);
        if (!chk(hipEventSynchronize(e1), "hipEventSynchronize")) rc = TETSIM_EHIP;
        float ms = 0.0f;
        if (!rc && chk(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime")) *gbps_out = 2.0 * static_cast<double>(n * 16) * reps / (static_cast<double>(ms) * 1.0e6);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

// ---- multi-GPU -----------------------------------------------------------------------------------------------
int tetsim_comm_unique_id(void* id128) {
    if (!id128) return fail(nullptr, TETSIM_EINVAL, "null id buffer");
    if (!g_rccl.load()) return fail(nullptr, TETSIM_ECOMM, g_rccl.err);
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, TETSIM_ECOMM, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int tetsim_comm_init(tetsim_handle h, const void* id128, int32_t rank, int32_t nranks) {
    if (!h || !id128) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return fail(h, TETSIM_ESTATE, "halo exchange exists only for POLAR_JACOBI");
    // Measurement aid: TETSIM_DEBUG_LOOPBACK_HALO=1 + nranks == 1 on a PARTITIONED body makes every neighbour this rank itself:
    // the real RCCL send/recv kernels then run in the real choreography on one GPU (ghosts receive this rank's own interface
    // values, so the physics is meaningless -- timing and liveness only).
    const char* lb = getenv("TETSIM_DEBUG_LOOPBACK_HALO");
    if (lb && lb[0] == '1' && nranks == 1 && rank == 0 && h->opt.part_count > 1) {
        for (auto& nb : h->neigh)
            if (nb.send_count != nb.recv_count) return fail(h, TETSIM_ESTATE, "loopback halo needs equal send and receive counts per neighbour (use equal slabs)");
        h->loopback = true;
        fprintf(stderr, "[tetsim] WARNING: TETSIM_DEBUG_LOOPBACK_HALO: partition %d exchanges halos with ITSELF; results are not physics\n", h->opt.part_index);
    } else if (nranks != h->opt.part_count || rank != h->opt.part_index) return fail(h, TETSIM_EINVAL, "rank/nranks must equal part_index/part_count");
    if (!g_rccl.load()) return fail(h, TETSIM_ECOMM, g_rccl.err);
    HIPCHK(h, hipSetDevice(h->opt.device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&h->comm, nranks, id, rank);
    if (r != ncclSuccess) { h->comm = nullptr; return rccl_fail(h, r, "ncclCommInitRank"); }
    h->comm_rank = rank;
    h->comm_size = nranks;
    { int rc = create_halo_stream(h); if (rc) return rc; }
    // Connection set-up happens on the first transfer between two ranks and can take seconds; do it here, with the real
    // message sizes on scratch buffers and a host-side wait, so that the stepping path (whose device-side waits are
    // bounded, TETSIM_HALO_TIMEOUT_MS) never sees it.  Collective: every rank of the communicator is inside this call.
    size_t most = 0;
    for (auto& nb : h->neigh) most = std::max<size_t>(most, std::max(nb.send_count, nb.recv_count));
    if (most) {
        float4 *src = nullptr, *dst = nullptr;
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&src), most * sizeof(float4)));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&dst), most * h->neigh.size() * sizeof(float4)));
        int rc = TETSIM_OK;
        if (hipMemsetAsync(src, 0, most * sizeof(float4), h->comm_stream) != hipSuccess) rc = fail(h, TETSIM_EHIP, "halo warm-up memset failed");
        r = rc ? ncclSuccess : g_rccl.GroupStart();
        size_t k = 0;
        for (auto& nb : h->neigh) {
            const int peer = h->loopback ? h->comm_rank : nb.rank;
            if (!rc && r == ncclSuccess && nb.send_count) r = g_rccl.Send(src, 4ull * nb.send_count, ncclFloat, peer, h->comm, h->comm_stream);
            if (!rc && r == ncclSuccess && nb.recv_count) r = g_rccl.Recv(dst + most * k, 4ull * nb.recv_count, ncclFloat, peer, h->comm, h->comm_stream);
            k++;
        }
        if (!rc && r == ncclSuccess) r = g_rccl.GroupEnd();
        if (!rc && r != ncclSuccess) rc = rccl_fail(h, r, "halo warm-up send/recv");
        if (!rc && hipStreamSynchronize(h->comm_stream) != hipSuccess) rc = fail(h, TETSIM_EHIP, "halo warm-up did not complete");
        (void)hipFree(src);
        (void)hipFree(dst);
        if (rc) return rc;
    }
    return 0;
}

int tetsim_comm_selftest(tetsim_handle h) {
    if (!h) return TETSIM_EINVAL;
    if (!h->comm) return fail(h, TETSIM_ESTATE, "no communicator (call tetsim_comm_init first)");
    HIPCHK(h, hipSetDevice(h->opt.device));
    constexpr size_t kN = 256;  // floats
    float *src = nullptr, *dst = nullptr;
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&src), kN * sizeof(float)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&dst), kN * sizeof(float)));
    std::vector<float> host(kN), back(kN, 0.0f);
    for (size_t i = 0; i < kN; i++) host[i] = static_cast<float>(i) * 0.5f + static_cast<float>(h->comm_rank);
    int rc = TETSIM_OK;
    if (hipMemcpy(src, host.data(), kN * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(dst, 0, kN * sizeof(float)) != hipSuccess) rc = fail(h, TETSIM_EHIP, "selftest upload failed");
    if (!rc) {
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.Send(src, kN, ncclFloat, h->comm_rank, h->comm, h->comm_stream);
        if (r == ncclSuccess) r = g_rccl.Recv(dst, kN, ncclFloat, h->comm_rank, h->comm, h->comm_stream);
        if (r == ncclSuccess) r = g_rccl.GroupEnd();
        if (r != ncclSuccess) rc = rccl_fail(h, r, "selftest send/recv");
    }
    if (!rc && (hipStreamSynchronize(h->comm_stream) != hipSuccess ||
                hipMemcpy(back.data(), dst, kN * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(h, TETSIM_EHIP, "selftest download failed");
    if (!rc && back != host) rc = fail(h, TETSIM_ECOMM, "selftest: received bytes differ from the bytes sent");
    (void)hipFree(src);
    (void)hipFree(dst);
    return rc;
}

// Measurement helper (multi-GPU design input): cost of ONE grouped ncclSend+ncclRecv of `bytes` to this rank itself,
// issued `reps` times back to back -- eagerly (use_graph = 0) or captured `per_graph` at a time into a HIP graph and
// replayed (use_graph = 1).  host_us = host time spent issuing, per group; total_us = wall time to completion, per group.
int tetsim_comm_probe(tetsim_handle h, uint64_t bytes, uint32_t reps, int32_t use_graph, uint32_t per_graph, double* host_us, double* total_us) {
    if (!h || !host_us || !total_us || reps == 0 || bytes < 4) return fail(h, TETSIM_EINVAL, "bad argument");
    if (!h->comm) return fail(h, TETSIM_ESTATE, "no communicator (call tetsim_comm_init first)");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const size_t n = bytes / sizeof(float);
    float *src = nullptr, *dst = nullptr;
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&src), n * sizeof(float)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&dst), n * sizeof(float)));
    std::vector<float> host(n), back(n, 0.0f);
    for (size_t i = 0; i < n; i++) host[i] = static_cast<float>(i % 977) + 0.25f;
    int rc = TETSIM_OK;
    if (hipMemcpy(src, host.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess || hipMemset(dst, 0, n * sizeof(float)) != hipSuccess)
        rc = fail(h, TETSIM_EHIP, "probe upload failed");
    auto group = [&]() -> ncclResult_t {
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.Send(src, n, ncclFloat, h->comm_rank, h->comm, h->comm_stream);
        if (r == ncclSuccess) r = g_rccl.Recv(dst, n, ncclFloat, h->comm_rank, h->comm, h->comm_stream);
        if (r == ncclSuccess) r = g_rccl.GroupEnd();
        return r;
    };
    using clk = std::chrono::steady_clock;
    if (!rc) {  // warm-up (connection setup happens on first use)
        ncclResult_t r = group();
        if (r != ncclSuccess) rc = rccl_fail(h, r, "probe warm-up");
        else if (hipStreamSynchronize(h->comm_stream) != hipSuccess) rc = fail(h, TETSIM_EHIP, "probe warm-up sync failed");
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (!rc && use_graph) {
        if (per_graph == 0) per_graph = 1;
        if (hipStreamBeginCapture(h->comm_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = fail(h, TETSIM_EHIP, "probe: begin capture failed");
        for (uint32_t i = 0; !rc && i < per_graph; i++) {
            ncclResult_t r = group();
            if (r != ncclSuccess) rc = rccl_fail(h, r, "probe: send/recv under stream capture");
        }
        hipError_t e = hipStreamEndCapture(h->comm_stream, &graph);
        if (!rc && e != hipSuccess) rc = fail(h, TETSIM_EHIP, std::string("probe: end capture: ") + hipGetErrorString(e));
        if (!rc && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(h, TETSIM_EHIP, "probe: graph instantiate failed");
    }
    if (!rc) {
        (void)hipMemset(dst, 0, n * sizeof(float));
        (void)hipDeviceSynchronize();
        const auto t0 = clk::now();
        uint32_t done = 0;
        if (use_graph) {
            for (; done < reps && !rc; done += per_graph)
                if (hipGraphLaunch(exec, h->comm_stream) != hipSuccess) rc = fail(h, TETSIM_EHIP, "probe: graph launch failed");
        } else {
            for (; done < reps && !rc; done++) {
                ncclResult_t r = group();
                if (r != ncclSuccess) rc = rccl_fail(h, r, "probe send/recv");
            }
        }
        const auto t1 = clk::now();
        if (!rc && hipStreamSynchronize(h->comm_stream) != hipSuccess) rc = fail(h, TETSIM_EHIP, "probe sync failed");
        const auto t2 = clk::now();
        if (!rc) {
            *host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / done;
            *total_us = std::chrono::duration<double, std::micro>(t2 - t0).count() / done;
            if (hipMemcpy(back.data(), dst, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, TETSIM_EHIP, "probe download failed");
            else if (back != host) rc = fail(h, TETSIM_ECOMM, "probe: received bytes differ from the bytes sent");
        }
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipFree(src);
    (void)hipFree(dst);
    return rc;
}

int tetsim_get_halo_plan(tetsim_handle h, int32_t* neigh, int32_t* send_counts, int32_t* recv_counts, int32_t* send_ids, int32_t* recv_ids) {
    if (!h) return TETSIM_EINVAL;
    size_t so = 0, ro = 0;
    for (size_t i = 0; i < h->neigh.size(); i++) {
        const NeighDev& nb = h->neigh[i];
        if (neigh) neigh[i] = nb.rank;
        if (send_counts) send_counts[i] = static_cast<int32_t>(nb.send_count);
        if (recv_counts) recv_counts[i] = static_cast<int32_t>(nb.recv_count);
        if (send_ids) std::copy(nb.send_global.begin(), nb.send_global.end(), send_ids + so);
        if (recv_ids) std::copy(nb.recv_global.begin(), nb.recv_global.end(), recv_ids + ro);
        so += nb.send_global.size();
        ro += nb.recv_global.size();
    }
    return 0;
}

int tetsim_halo_export(tetsim_handle h, uint32_t n, float* out_xyzw) {
    if (!h || !out_xyzw || n >= h->neigh.size()) return fail(h, TETSIM_EINVAL, "bad neighbour slot");
    NeighDev& nb = h->neigh[n];
    if (!nb.send_count) return 0;
    util_launch_gather4(h->stream, h->pj.pos_pred, nb.send_idx, nb.send_buf, nb.send_count);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out_xyzw, nb.send_buf, nb.send_count * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
int tetsim_halo_import(tetsim_handle h, uint32_t n, const float* in_xyzw) {
    if (!h || !in_xyzw || n >= h->neigh.size()) return fail(h, TETSIM_EINVAL, "bad neighbour slot");
    NeighDev& nb = h->neigh[n];
    if (!nb.recv_count) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->pj.pos_pred + nb.recv_start, in_xyzw, nb.recv_count * sizeof(float4), hipMemcpyHostToDevice));
    return 0;
}

int tetsim_group_step_n(tetsim_handle* hs, uint32_t count, uint32_t n, double dt, const TetSimParams* params) {
    if (!hs || count == 0) return TETSIM_EINVAL;
    for (uint32_t i = 0; i < count; i++) {
        tetsim_body* h = hs[i];
        if (!h || h->opt.part_count != static_cast<int32_t>(count) || h->opt.part_index != static_cast<int32_t>(i) || h->comm)
            return fail(h, TETSIM_EINVAL, "handles[i] must be partition i of a count-way decomposition without an RCCL communicator");
        if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return fail(h, TETSIM_ESTATE, "POLAR_JACOBI only");
        if (h->group.empty()) {  // first use: wire the group and give every partition its halo stream
            h->group.assign(hs, hs + count);
            { int rc = create_halo_stream(h); if (rc) return rc; }
        }
    }
    for (uint32_t i = 0; i < count; i++) {
        int rc = push_params(hs[i], dt, params);
        if (!rc) rc = ensure_prediction(hs[i], dt);
        if (rc) return rc;
    }
    bool refresh = false;
    for (uint32_t i = 0; i < count; i++) refresh = refresh || hs[i]->needs_halo_refresh;
    if (refresh) {  // dt changed: every member redid its predictions; re-send them (all "ghosts are free" records, then all copies)
        for (uint32_t i = 0; i < count; i++) {
            hs[i]->needs_halo_refresh = false;
            // "my ghosts may be overwritten": flag bodies record it on the halo stream, which ensure_prediction put behind the re-prediction
            HIPCHK(hs[i], hipEventRecord(hs[i]->ev_boundary2[hs[i]->halo_parity], hs[i]->flag_sync ? hs[i]->comm_stream : hs[i]->stream));
        }
        for (uint32_t i = 0; i < count; i++) { int rc = enqueue_phase_b(hs[i]); if (rc) return rc; }
    }
    static const bool dbg_sync = getenv("TETSIM_DEBUG_GROUP_SYNC") != nullptr;  // development: serialise every phase
    for (uint32_t s = 0; s < n; s++) {
        for (uint32_t i = 0; i < count; i++) { int rc = enqueue_phase_a(hs[i]); if (rc) return rc; }
        if (dbg_sync) (void)hipDeviceSynchronize();
        for (uint32_t i = 0; i < count; i++) { int rc = enqueue_phase_b(hs[i]); if (rc) return rc; }
        if (dbg_sync) (void)hipDeviceSynchronize();
    }
    return 0;
}

int tetsim_halo_exchange_local(tetsim_handle* hs, uint32_t count) {
    if (!hs || count == 0) return TETSIM_EINVAL;
    for (uint32_t i = 0; i < count; i++) {
        if (!hs[i] || hs[i]->opt.part_count != static_cast<int32_t>(count) || hs[i]->opt.part_index != static_cast<int32_t>(i))
            return fail(hs[i], TETSIM_EINVAL, "handles[i] must be partition i of a count-way decomposition");
    }
    // every partition must have finished its vertex kernel before anyone's ghosts are overwritten
    for (uint32_t i = 0; i < count; i++) HIPCHK(hs[i], hipStreamSynchronize(hs[i]->stream));
    for (uint32_t i = 0; i < count; i++) {
        tetsim_body* src = hs[i];
        for (auto& nb : src->neigh) {
            if (!nb.send_count) continue;
            tetsim_body* dst = hs[nb.rank];
            NeighDev* back = nullptr;
            for (auto& r : dst->neigh) if (r.rank == static_cast<int>(i)) back = &r;
            if (!back || back->recv_count != nb.send_count) return fail(src, TETSIM_ESTATE, "asymmetric halo plan");
            const float4* from = nb.contiguous ? src->pj.pos_pred + nb.send_first : nb.send_buf;
            if (!nb.contiguous) util_launch_gather4(src->stream, src->pj.pos_pred, nb.send_idx, nb.send_buf, nb.send_count);
            HIPCHK(src, hipMemcpyAsync(dst->pj.pos_pred + back->recv_start, from, nb.send_count * sizeof(float4), hipMemcpyDeviceToDevice, src->stream));
        }
    }
    for (uint32_t i = 0; i < count; i++) HIPCHK(hs[i], hipStreamSynchronize(hs[i]->stream));
    return 0;
}

// ---- host-only preprocessing ---------------------------------------------------------------------------------
int tetsim_prep_levels(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* level, uint32_t* num_levels) {
    if ((nt && (!tets || !level)) || !num_levels) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    *num_levels = prep_levels(tets, nt, nv, level);
    return 0;
}
int tetsim_prep_colours(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* colour, uint32_t* num_colours) {
    if ((nt && (!tets || !colour)) || !num_colours) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    *num_colours = prep_colours(tets, nt, nv, colour);
    return 0;
}
int tetsim_prep_clusters(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* order, int32_t* launch, int32_t* lane, int32_t* step,
                         uint32_t* num_launches, uint32_t* num_clusters) {
    if ((nt && (!tets || !order || !launch || !lane || !step)) || !num_launches) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    const ClusterPlan P = prep_clusters(tets, nt, nv);
    for (uint32_t i = 0; i < nt; i++) order[i] = P.pre[i];
    const uint32_t nl = static_cast<uint32_t>(P.launch_off.size() - 1);
    for (uint32_t l = 0; l < nl; l++)
        for (uint32_t j = P.step_off[l]; j < P.step_off[l + 1]; j++)
            for (uint32_t i = 0; i < P.step_count[j]; i++) {
                const uint32_t pos = P.exec_pos[P.step_first[j] + i];
                launch[pos] = static_cast<int32_t>(l);
                lane[pos] = static_cast<int32_t>(i);
                step[pos] = static_cast<int32_t>(j - P.step_off[l]);
            }
    *num_launches = nl;
    if (num_clusters) *num_clusters = P.num_clusters;
    return 0;
}
int tetsim_prep_slot_table(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t ref_quirk, int32_t* slots, uint32_t* dropped) {
    if ((nt && !tets) || !slots) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    const uint32_t d = prep_slot_table(tets, nt, nv, ref_quirk != 0, slots);
    if (dropped) *dropped = d;
    return 0;
}
int tetsim_prep_ref_grab_texels(int32_t grab_id, uint32_t num_elems, uint32_t num_particles, int32_t out[2]) {
    if (!out) return TETSIM_EINVAL;
    ref_grab_texels(grab_id, num_elems, num_particles, out);
    return 0;
}
int tetsim_prep_rest(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, double density, float* inv_mass, float* inv_rest_pose, float* inv_rest_volume) {
    if (!inv_mass || (nt && (!inv_rest_pose || !inv_rest_volume))) return TETSIM_EINVAL;
    std::string e = validate_mesh(verts, nv, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    prep_rest(verts, nv, tets, nt, density, inv_mass, inv_rest_pose, inv_rest_volume);
    return 0;
}

// ---- partition plan (host only) ---------------------------------------------------------------------------------
struct tetsim_plan_s { Partition P; };

int tetsim_plan_create(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t part_count, int32_t part_index,
                       const int32_t* vert_owner, tetsim_plan* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "null plan pointer");
    *out = nullptr;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    tetsim_plan_s* p = new tetsim_plan_s();
    e = build_partition(tets, nt, nv, part_count, part_index, vert_owner, &p->P);
    if (!e.empty()) { delete p; return fail(nullptr, TETSIM_EINVAL, e); }
    *out = p;
    return 0;
}
void tetsim_plan_destroy(tetsim_plan p) { delete p; }
int tetsim_plan_sizes(tetsim_plan p, TetSimPlanSizes* out) {
    if (!p || !out) return TETSIM_EINVAL;
    out->owned_particles = p->P.n_owned;
    out->boundary_particles = p->P.n_boundary;
    out->local_particles = static_cast<uint32_t>(p->P.local_to_global_vert.size());
    out->local_elems = static_cast<uint32_t>(p->P.local_to_global_tet.size());
    out->owned_elems = p->P.owned_tets;
    out->num_neighbours = static_cast<uint32_t>(p->P.neigh.size());
    return 0;
}
int tetsim_plan_arrays(tetsim_plan p, int32_t* l2gv, int32_t* l2gt, int32_t* ltets) {
    if (!p) return TETSIM_EINVAL;
    if (l2gv) std::copy(p->P.local_to_global_vert.begin(), p->P.local_to_global_vert.end(), l2gv);
    if (l2gt) std::copy(p->P.local_to_global_tet.begin(), p->P.local_to_global_tet.end(), l2gt);
    if (ltets) std::copy(p->P.local_tets.begin(), p->P.local_tets.end(), ltets);
    return 0;
}
int tetsim_plan_neighbour(tetsim_plan p, uint32_t i, int32_t* rank, uint32_t* send_count, uint32_t* recv_start,
                          uint32_t* recv_count, int32_t* contiguous) {
    if (!p || i >= p->P.neigh.size()) return TETSIM_EINVAL;
    const auto& nb = p->P.neigh[i];
    if (rank) *rank = nb.rank;
    if (send_count) *send_count = static_cast<uint32_t>(nb.send_local.size());
    if (recv_start) *recv_start = nb.recv_start;
    if (recv_count) *recv_count = nb.recv_count;
    if (contiguous) *contiguous = nb.send_contiguous ? 1 : 0;
    return 0;
}
int tetsim_plan_neighbour_ids(tetsim_plan p, uint32_t i, int32_t* send_local, int32_t* send_global, int32_t* recv_global) {
    if (!p || i >= p->P.neigh.size()) return TETSIM_EINVAL;
    const auto& nb = p->P.neigh[i];
    if (send_local) std::copy(nb.send_local.begin(), nb.send_local.end(), send_local);
    if (send_global) std::copy(nb.send_global.begin(), nb.send_global.end(), send_global);
    if (recv_global) std::copy(nb.recv_global.begin(), nb.recv_global.end(), recv_global);
    return 0;
}

}  // extern "C"
