// tetsim_state.hip -- C ABI, part 2 (include/tetsim.h): reading the solver's state back (copying and pinned zero-copy reads),
// checkpoint / resume of the complete state, and the small getters (plans, orders, inverse masses).  See body.h.
#include <thread>

#include "body.h"

using namespace tetsim;

namespace tetsim {
const float4* current_positions(tetsim_body* h) { return h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI ? h->pj.pos_final : h->nh.pos; }
// TETSIM_FLAG_LEAN_STATE: the accumulated quaternions are not part of the substep (pj_blocked.hip); whoever reads pj.quat calls this first.
// The halo queue's tiles write carried shapes too, so both queues are drained (a read-out path: once per frame at most).
int ensure_quats(tetsim_body* h) {
    if (!h->blocked || !h->blk.lean_state || !h->quat_stale) return 0;
    HIPCHK(h, hipSetDevice(h->opt.device));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    pjb_launch_recover_quats(h->stream, h->blk, h->rest0_a, h->rest0_b, h->rest0_c);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, TETSIM_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    h->quat_stale = false;
    return 0;
}
int ensure_index_map(tetsim_body* h) {  // internal Morton numbering -> API numbering, on the device
    if (h->d_api2dev || h->api2dev.empty()) return 0;
    int rc = dev_alloc(h, &h->d_api2dev, h->api2dev.size());
    if (rc) return rc;
    return upload(h, h->d_api2dev, h->api2dev);
}
}  // namespace tetsim

extern "C" {

int tetsim_read_positions(tetsim_handle h, float* out) {
    if (!h) return TETSIM_EINVAL;
    return h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI ? read_float4_as_xyz(h, h->pj.pos_final, h->pj.nv_owned, out)
                                                       : read_float4_as_xyz(h, h->nh.pos, h->nh.nv, out);
}

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
    HIPCHK(h, hipSetDevice(h->opt.device));
    if (int rc = ensure_quats(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    if (h->pj.nt) HIPCHK(h, hipMemcpy(out, h->pj.quat, h->pj.nt * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
int tetsim_read_quats_pinned(tetsim_handle h, const float** out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return fail(h, TETSIM_ESTATE, "quaternions exist only for POLAR_JACOBI");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const size_t n = h->pj.nt;
    if (!h->pinned_quat) HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pinned_quat), std::max<size_t>(n, 1) * sizeof(float4), hipHostMallocDefault));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));  // ghost tiles write their quaternions on the halo stream
    if (int rc = ensure_quats(h)) return rc;
    if (n) HIPCHK(h, hipMemcpyAsync(h->pinned_quat, h->pj.quat, n * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *out = h->pinned_quat;
    return 0;
}

// ---- checkpoint / resume: the complete solver state as one blob (device order: only this library reads it back) -----------
namespace {
struct StateHeader {
    uint32_t magic, abi, solver, precision, flags, blocked, order;
    uint32_t nv, nt, pred_any_dt;
    float dt_pred;
    uint32_t reserved;
    uint64_t payload;
    uint64_t mesh_digest;   // FNV-1a over vertices, tets, density and the batch layout: same counts, other mesh -> rejected
};
uint64_t mesh_digest(const tetsim_body* h) {
    uint64_t d = 0xcbf29ce484222325ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { d ^= b[i]; d *= 0x100000001b3ull; } };
    mix(h->h_verts.data(), h->h_verts.size() * sizeof(float));
    mix(h->h_tets.data(), h->h_tets.size() * sizeof(int32_t));
    mix(&h->opt.density, sizeof(h->opt.density));
    mix(h->batch_first_vert.data(), h->batch_first_vert.size() * sizeof(uint32_t));
    mix(h->batch_first_tet.data(), h->batch_first_tet.size() * sizeof(uint32_t));
    if (h->partitioned) {
        // a partition's blob belongs to THIS cut of the mesh: which part of how many, which particles it holds in which local order
        // (owned boundary | owned interior | ghosts by owner -- i.e. the owner map as far as this part sees it), which tets, how deep
        const int32_t id[3] = {h->opt.part_count, h->opt.part_index, h->part.depth};
        mix(id, sizeof id);
        mix(&h->part.n_owned, sizeof(h->part.n_owned));
        mix(h->part.local_to_global_vert.data(), h->part.local_to_global_vert.size() * sizeof(int32_t));
        mix(h->part.local_to_global_tet.data(), h->part.local_to_global_tet.size() * sizeof(int32_t));
    }
    return d;
}
constexpr uint32_t kStateMagic = 0x54535354u;  // "TSST"
struct StateSection { void* ptr; size_t bytes; };
void state_sections(tetsim_body* h, std::vector<StateSection>& v) {
    if (h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI) {
        const size_t nvl = h->pj.nv_local, nt = h->pj.nt;
        v.push_back({h->pj.pos_final, nvl * sizeof(float4)});
        v.push_back({h->pj.vel, nvl * sizeof(float4)});
        v.push_back({h->pj.pos_pred, nvl * sizeof(float4)});
        v.push_back({h->pj.quat, nt * sizeof(float4)});
        if (h->blocked) {
            if (h->blk.lean_state) {  // three corners (the quaternion section above is brought up to date before a save: ensure_quats)
                v.push_back({h->blk.rest_a, nt * sizeof(float4)});
                v.push_back({h->blk.rest_b, nt * sizeof(float4)});
                v.push_back({h->blk.rest_c1, nt * sizeof(float)});
            } else if (!h->blk.lean) {  // constant-rest-shape bodies carry no shape state
                v.push_back({h->blk.rest_a, nt * sizeof(float4)});
                v.push_back({h->blk.rest_b, nt * sizeof(float4)});
                v.push_back({h->blk.rest_c, nt * sizeof(float4)});
            }
        } else v.push_back({h->pj.elem, 4ull * h->pj.nt_pad * sizeof(float4)});
    } else {
        const size_t nv = h->nh.nv;
        v.push_back({h->nh.pos, nv * sizeof(float4)});
        v.push_back({h->nh.prev, nv * sizeof(float4)});
        v.push_back({h->nh.vel, nv * sizeof(float4)});
        v.push_back({h->nh.vol_err, h->nh.nt * sizeof(double)});
    }
}
StateHeader state_header(tetsim_body* h) {
    StateHeader hd{};
    hd.magic = kStateMagic; hd.abi = TETSIM_ABI_VERSION;
    hd.solver = static_cast<uint32_t>(h->opt.solver); hd.precision = static_cast<uint32_t>(h->opt.precision);
    hd.flags = h->opt.flags; hd.blocked = h->blocked ? 1u : 0u; hd.order = static_cast<uint32_t>(h->opt.order);
    hd.nv = h->info.num_particles; hd.nt = h->info.num_elems;
    hd.pred_any_dt = h->pred_any_dt ? 1u : 0u; hd.dt_pred = h->dt_pred;
    std::vector<StateSection> secs;
    state_sections(h, secs);
    for (const StateSection& sec : secs) hd.payload += sec.bytes;
    hd.mesh_digest = mesh_digest(h);
    return hd;
}
// A PARTITION's blob is its local state in local order: owned particles AND ghosts (the predictions its neighbours sent for the next
// substep are state -- they are what the next substep's ghost tets read), local tets incl. ghost tets.  Every rank saves its own blob at
// the same substep count (after tetsim_sync) and loads its own; nothing crosses ranks.  What is NOT in it is transport state: the
// semaphore words are at rest after a sync, and the peer-to-peer halo's substep parity stays the restored body's own -- so the ghosts
// are read from the buffer the NEXT substep will read and written back into both.  Two-layer ghost regions keep four more receive
// buffers in flight between even substeps: not supported (save / load such a decomposition with one layer).
int state_guard(tetsim_body* h) {
    if (h->partitioned && h->deep) return fail(h, TETSIM_ESTATE, "save/load_state of a body with a two-layer ghost region (TETSIM_FLAG_DEEP_GHOSTS) is not supported");
    if (h->partitioned && h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return fail(h, TETSIM_ESTATE, "partitioned bodies are POLAR_JACOBI bodies");
    return 0;
}
// Everything that may still write into this body's arrays has to be done before they are copied: its own two queues, and -- for the
// partitions of one process (in-process group: sender-driven copies, peer-to-peer stores through plain pointers) -- the queues of
// the other members, whose transfers of the last substep land in THIS body's ghost range.
int quiesce(tetsim_body* h) {
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    for (tetsim_body* g : h->group) {
        if (!g || g == h) continue;
        HIPCHK(h, hipSetDevice(g->opt.device));
        HIPCHK(h, hipStreamSynchronize(g->stream));
        if (g->comm_stream) HIPCHK(h, hipStreamSynchronize(g->comm_stream));
    }
    HIPCHK(h, hipSetDevice(h->opt.device));
    return 0;
}
// peer-to-peer bodies on an odd substep parity read their ghosts from ghost_alt, not from pos_pred's tail: the blob's pos_pred section
// gets them straight from there.  (Round 5 copied ghost_alt into pos_pred's tail on the device first -- but that tail is the LIVE
// receive buffer of the substep after next: a neighbour of another process that has resumed stepping may already have stored there.)
int patch_blob_ghosts(tetsim_body* h, char* pos_pred_section) {
    const size_t ng = h->pj.nv_local - h->pj.nv_owned;
    if (h->partitioned && h->p2p && h->ghost_alt && ng && (h->p2p_round & 1u))
        HIPCHK(h, hipMemcpy(pos_pred_section + static_cast<size_t>(h->pj.nv_owned) * sizeof(float4), h->ghost_alt, ng * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
// One rank per PROCESS with the peer-to-peer halo: this rank's ghosts of the next substep are stored by the NEIGHBOURS' queues, which
// quiesce() cannot drain.  Every rank raises its neighbours' "arrived" words of the next substep's parity when its call ends (flush_v)
// or when its next substep starts, and nothing clears them between calls: wait for them (host-side looks, bounded like every wait).
int await_peer_deliveries(tetsim_body* h) {
    if (!(h->partitioned && h->p2p && h->group.empty() && h->d_arrived && h->p2p_round > 0 && !h->loopback && !h->deep)) return 0;
    const uint32_t par = static_cast<uint32_t>(h->p2p_round & 1u);
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t limit_ms = h->timeout_ms ? h->timeout_ms : 30000u;
    for (;;) {
        uint32_t words[kMaxPeers] = {};
        HIPCHK(h, hipMemcpy(words, h->d_arrived + par * kMaxPeers, sizeof words, hipMemcpyDeviceToHost));
        bool all = true;
        for (size_t i = 0; i < h->neigh.size() && i < kMaxPeers; i++) if (h->neigh[i].recv_count && words[i] == 0u) all = false;
        if (all) return 0;
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > limit_ms)
            return fail(h, TETSIM_ECOMM, "tetsim_save_state: a neighbour's boundary predictions of the last substep did not arrive in time (every rank saves at the same "
                                         "substep count, after its own tetsim_sync)");
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
int ghosts_from_tail(tetsim_body* h) {
    const size_t ng = h->pj.nv_local - h->pj.nv_owned;
    if (h->partitioned && h->ghost_alt && ng)
        HIPCHK(h, hipMemcpy(h->ghost_alt, h->pj.pos_pred + h->pj.nv_owned, ng * sizeof(float4), hipMemcpyDeviceToDevice));
    return 0;
}
}  // namespace

int tetsim_state_size(tetsim_handle h, uint64_t* bytes_out) {
    if (!h || !bytes_out) return fail(h, TETSIM_EINVAL, "null argument");
    if (int rc = state_guard(h)) return rc;
    *bytes_out = sizeof(StateHeader) + state_header(h).payload;
    return 0;
}
int tetsim_save_state(tetsim_handle h, void* blob, uint64_t bytes) {
    if (!h || !blob) return fail(h, TETSIM_EINVAL, "null argument");
    if (int rc = state_guard(h)) return rc;
    const StateHeader hd = state_header(h);
    if (bytes < sizeof(hd) + hd.payload) return fail(h, TETSIM_EINVAL, "state buffer too small (tetsim_state_size)");
    if (int rc = ensure_quats(h)) return rc;
    if (int rc = quiesce(h)) return rc;
    if (int rc = await_peer_deliveries(h)) return rc;
    char* out = static_cast<char*>(blob);
    std::memcpy(out, &hd, sizeof(hd));
    out += sizeof(hd);
    std::vector<StateSection> secs;
    state_sections(h, secs);
    for (const StateSection& sec : secs) {
        if (sec.bytes) HIPCHK(h, hipMemcpy(out, sec.ptr, sec.bytes, hipMemcpyDeviceToHost));
        if (sec.ptr == h->pj.pos_pred && h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI) { if (int rc = patch_blob_ghosts(h, out)) return rc; }
        out += sec.bytes;
    }
    return 0;
}
int tetsim_load_state(tetsim_handle h, const void* blob, uint64_t bytes) {
    if (!h || !blob) return fail(h, TETSIM_EINVAL, "null argument");
    if (int rc = state_guard(h)) return rc;
    StateHeader in{};
    if (bytes < sizeof(in)) return fail(h, TETSIM_EINVAL, "state blob is truncated");
    std::memcpy(&in, blob, sizeof(in));
    const StateHeader want = state_header(h);
    if (in.magic != kStateMagic) return fail(h, TETSIM_EINVAL, "not a tetsim state blob (bad magic)");
    if (in.abi != want.abi) return fail(h, TETSIM_EINVAL, "state blob was written by ABI " + std::to_string(in.abi) + ", this library is ABI " + std::to_string(want.abi));
    if (in.solver != want.solver || in.precision != want.precision || in.flags != want.flags || in.blocked != want.blocked || in.order != want.order ||
        in.nv != want.nv || in.nt != want.nt || in.payload != want.payload)
        return fail(h, TETSIM_EINVAL, "state blob belongs to a body with another mesh or other options");
    if (in.mesh_digest != want.mesh_digest) return fail(h, TETSIM_EINVAL, "state blob belongs to another mesh (same counts, different vertices / tets / density / batch layout)");
    if (bytes < sizeof(in) + in.payload) return fail(h, TETSIM_EINVAL, "state blob is truncated");
    if (int rc = quiesce(h)) return rc;
    const char* src = static_cast<const char*>(blob) + sizeof(in);
    std::vector<StateSection> secs;
    state_sections(h, secs);
    for (const StateSection& sec : secs) {
        if (sec.bytes) HIPCHK(h, hipMemcpy(sec.ptr, src, sec.bytes, hipMemcpyHostToDevice));
        src += sec.bytes;
    }
    if (int rc = ghosts_from_tail(h)) return rc;
    h->pred_any_dt = in.pred_any_dt != 0;
    h->dt_pred = in.dt_pred;
    h->final_ghosts_fresh = false;
    h->quat_stale = false;   // (a lean-state blob holds the quaternions recovered from the very shape it holds)
    return 0;
}

int tetsim_read_vol_error(tetsim_handle h, double* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->opt.solver != TETSIM_SOLVER_NEOHOOKEAN_GS) return fail(h, TETSIM_ESTATE, "volError exists only for NEOHOOKEAN_GS");
    HIPCHK(h, hipSetDevice(h->opt.device));
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
    HIPCHK(h, hipSetDevice(h->opt.device));
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

}  // extern "C"
