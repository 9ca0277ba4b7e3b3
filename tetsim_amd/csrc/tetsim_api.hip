// tetsim_api.hip -- C ABI of libtetsim_hip.so (include/tetsim.h): handle lifecycle, stepping (stream / graph orchestration
// of the gfx950 kernels), state read-back, grab, visual mesh, measurement.  See body.h for the other translation units.
#include "body.h"

using namespace tetsim;

namespace tetsim {

namespace {
thread_local std::string g_create_error;
}
HostProf g_hostprof;

int fail(tetsim_body* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}
const char* create_error() { return g_create_error.c_str(); }

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
    HIPCHK(h, hipSetDevice(h->opt.device));  // group stepping walks over handles that may live on different devices
    const int slot = h->ring_pos;
    h->ring_pos = (h->ring_pos + 1) % kRing;
    if (h->ring_used[slot]) HIPCHK(h, hipEventSynchronize(h->ring_ev[slot]));
    if (h->ring_used_halo[slot]) { HIPCHK(h, hipEventSynchronize(h->ring_ev_halo[slot])); h->ring_used_halo[slot] = false; }
    fill_params(h, dt, *params, &h->h_ring[slot]);
    HIPCHK(h, hipMemcpyAsync(h->d_params, &h->h_ring[slot], sizeof(DevParams), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(h->ring_ev[slot], h->stream));
    h->ring_used[slot] = true;
    if (h->comm_stream && h->d_params_halo) {
        // the halo queue runs the boundary particles itself (enqueue_phase_a) and is ordered against the main queue only through
        // the semaphore words: it gets its own copy of the parameters, in ITS stream order
        if (!h->ring_ev_halo[slot]) HIPCHK(h, hipEventCreateWithFlags(&h->ring_ev_halo[slot], hipEventDisableTiming));
        HIPCHK(h, hipMemcpyAsync(h->d_params_halo, &h->h_ring[slot], sizeof(DevParams), hipMemcpyHostToDevice, h->comm_stream));
        HIPCHK(h, hipEventRecord(h->ring_ev_halo[slot], h->comm_stream));
        h->ring_used_halo[slot] = true;
    }
    h->fork_needed = true;  // whatever the caller did on the main stream since the last call must be visible to the boundary stream
    return 0;
}


// ---- kernel sequencing ---------------------------------------------------------------------------------
void pj_tet(tetsim_body* h, hipEvent_t e0, hipEvent_t e1) {
    if (h->blocked) pjb_launch_tet(h->stream, h->blk, 0, h->blk.nb, e0, e1);
    else h->fast ? pj_launch_tet_fast(h->stream, h->pj, e0, e1) : pj_launch_tet_precise(h->stream, h->pj, e0, e1);
}
void pj_vertex(tetsim_body* h, uint32_t first, uint32_t count, hipEvent_t e0, hipEvent_t e1) {
    if (h->blocked) pjb_launch_vertex(h->stream, h->blk, first, count, e0, e1);
    else h->fast ? pj_launch_vertex_fast(h->stream, h->pj, first, count, e0, e1) : pj_launch_vertex_precise(h->stream, h->pj, first, count, e0, e1);
}
// One substep of a fused body inside a run of substeps with one dt:
//   first substep:  plain tet kernel (predictions of the previous call's particle kernel)          -> partial sums A
//   substep s >= 1: fused kernel = particle update of s-1 (partial sums of s-1, positions in/out double buffered) + tet pass s
//   last substep:   ... followed by the particle kernel, which always leaves the positions in pj.pos_final
// e[0..3]: begin / end events of the tet (or fused) kernel and of the particle kernel (tetsim_profile)
void pj_fused_substep(tetsim_body* h, bool first, bool last, hipEvent_t* e) {
    PJBlk k = h->blk;
    if (first) { h->fuse_step = 0; h->fin_in_b = false; }
    const uint32_t s = h->fuse_step++;
    float4* const pbuf[2] = {h->blk.partial, h->partial_b};
    float4* const fbuf[2] = {h->pj.pos_final, h->pos_final_b};
    k.partial = pbuf[s & 1u];
    if (s == 0) pjb_launch_tet(h->stream, k, 0, k.nb, e ? e[0] : nullptr, e ? e[1] : nullptr);
    else {
        k.partial_prev = pbuf[(s - 1u) & 1u];
        k.fin_in = fbuf[h->fin_in_b ? 1 : 0];
        k.fin_out = fbuf[h->fin_in_b ? 0 : 1];
        pjb_launch_tet_fused(h->stream, k, e ? e[0] : nullptr, e ? e[1] : nullptr);
        h->fin_in_b = !h->fin_in_b;
    }
    if (last) {
        k.fin_in = fbuf[h->fin_in_b ? 1 : 0];
        k.fin_out = h->pj.pos_final;
        pjb_launch_vertex(h->stream, k, 0, h->pj.nv_owned, e ? e[2] : nullptr, e ? e[3] : nullptr);
        h->fin_in_b = false;
    }
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

// one substep's launches (parameters already on the device)
// first / last: position inside a run of substeps enqueued back to back with one dt (NEOHOOKEAN_GS fuses the particle pass
// that ends a substep with the prediction that starts the next one)
int enqueue_substep(tetsim_body* h, bool first, bool last) {
    if (h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI) {
        if (has_transport(h)) {
            int rc = enqueue_phase_a(h);
            if (!rc) rc = enqueue_phase_b(h);
            if (rc) return rc;
        } else if (h->fused) {
            pj_fused_substep(h, first, last, nullptr);
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
int ensure_prediction(tetsim_body* h, double dt) {
    if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return 0;
    const float fdt = static_cast<float>(dt);
    if (!h->pred_any_dt && fdt != h->dt_pred) {
        if (h->partitioned && !h->neigh.empty() && !has_transport(h))
            return fail(h, TETSIM_ESTATE, "dt changed between substeps on a partitioned body without a transport (ghost predictions would be stale): "
                                          "exchange halos through tetsim_comm_init / tetsim_group_step_n, or keep dt fixed");
        if (h->flag_sync && h->comm_stream) {
            // the boundary particles of the last substep were finished by the HALO queue (enqueue_phase_a), and its last transfer
            // still reads their predictions: the re-prediction on the main queue goes behind all of that (an eager cross-queue
            // event, ~15 us -- only when dt changes)
            HIPCHK(h, hipEventRecord(h->ev_bnd_tet, h->comm_stream));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_bnd_tet, 0));
        }
        pj_repredict(h);
        // the neighbours' ghost copies of our interface predictions are stale now: one extra halo exchange (every rank sees the
        // same dt change, so every rank does this).  RCCL bodies do it here; an in-process group does it for all its members
        // in tetsim_group_step_n (a copy waits for the RECEIVER's event, so all records must precede all copies).
        if (has_transport(h)) {
            if (h->flag_sync) {
                // the halo stream continues only after the new predictions exist: one more hand-over, through a word of its OWN.
                // (The regular V word is a binary semaphore whose producer and consumer alternate because the substep's dependency
                // cycle forces them to; this extra signal has no such back-edge -- it could land on a V that the halo stream has
                // not consumed yet, and the two would collapse into one.  Between two uses of this word lies at least one whole
                // substep, whose G hand-over orders them.)
                PJSync y;
                y.flag = h->d_sync + 3; y.error = h->d_sync + 4; y.timeout_ms = halo_timeout_ms();
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
    HIPCHK(h, hipSetDevice(h->opt.device));
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

}  // namespace tetsim


// =============================================================================================================
extern "C" {

int tetsim_abi_version(void) { return TETSIM_ABI_VERSION; }

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

const char* tetsim_last_error(tetsim_handle h) { return h ? h->err.c_str() : create_error(); }

namespace {
// batch_first_vert / batch_first_tet: [bodies + 1] ranges of a concatenation of independent bodies (tetsim_create_batch), or empty
int create_common(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, const TetSimOptions* opts, tetsim_handle* out,
                  const std::vector<uint32_t>& batch_first_vert, const std::vector<uint32_t>& batch_first_tet) {
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
    h->batch_first_vert = batch_first_vert;
    h->batch_first_tet = batch_first_tet;
    h->info.num_bodies = batch_first_vert.empty() ? 1u : static_cast<uint32_t>(batch_first_vert.size() - 1);

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
}  // namespace

int tetsim_create(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, const TetSimOptions* opts, tetsim_handle* out) {
    return create_common(verts, nv, tets, nt, opts, out, {}, {});
}

int tetsim_create_batch(const float* const* verts, const uint32_t* nv, const int32_t* const* tets, const uint32_t* nt, uint32_t count,
                        const TetSimOptions* opts, tetsim_handle* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "out handle pointer is null");
    *out = nullptr;
    if (!verts || !nv || !tets || !nt || count == 0) return fail(nullptr, TETSIM_EINVAL, "tetsim_create_batch: null argument or empty batch");
    if (opts && opts->part_count > 1) return fail(nullptr, TETSIM_EINVAL, "a batch cannot be partitioned (partition the bodies over handles instead)");
    if (opts && opts->tet_colour) return fail(nullptr, TETSIM_EINVAL, "tet_colour is not supported for batches");
    std::vector<uint32_t> fv(count + 1, 0), ft(count + 1, 0);
    for (uint32_t b = 0; b < count; b++) {
        const std::string merr = validate_mesh(verts[b], nv[b], tets[b], nt[b], opts && opts->solver == TETSIM_SOLVER_NEOHOOKEAN_GS);
        if (!merr.empty()) return fail(nullptr, TETSIM_EINVAL, "body " + std::to_string(b) + ": " + merr);
        const uint64_t v = static_cast<uint64_t>(fv[b]) + nv[b], t = static_cast<uint64_t>(ft[b]) + nt[b];
        if (v > 0x3fffffffull || t > 0x1fffffffull) return fail(nullptr, TETSIM_EINVAL, "batch too large for one handle");
        fv[b + 1] = static_cast<uint32_t>(v); ft[b + 1] = static_cast<uint32_t>(t);
    }
    std::vector<float> av(3ull * fv[count]);
    std::vector<int32_t> at(4ull * ft[count]);
    for (uint32_t b = 0; b < count; b++) {
        std::copy(verts[b], verts[b] + 3ull * nv[b], av.begin() + 3ull * fv[b]);
        for (uint64_t i = 0; i < 4ull * nt[b]; i++) at[4ull * ft[b] + i] = tets[b][i] + static_cast<int32_t>(fv[b]);
    }
    return create_common(av.data(), fv[count], at.data(), ft[count], opts, out, fv, ft);
}

int tetsim_get_batch_layout(tetsim_handle h, uint32_t* first_particle, uint32_t* first_elem) {
    if (!h || !first_particle || !first_elem) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->batch_first_vert.empty()) {
        first_particle[0] = 0; first_particle[1] = h->info.num_particles;
        first_elem[0] = 0; first_elem[1] = h->info.num_elems;
        return 0;
    }
    std::copy(h->batch_first_vert.begin(), h->batch_first_vert.end(), first_particle);
    std::copy(h->batch_first_tet.begin(), h->batch_first_tet.end(), first_elem);
    return 0;
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
    for (auto& kv : h->flag_graphs) { (void)hipGraphExecDestroy(kv.second.first); (void)hipGraphExecDestroy(kv.second.second); }
    h->flag_graphs.clear();
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->h_ring) (void)hipHostFree(h->h_ring);
    if (h->pinned_pos) (void)hipHostFree(h->pinned_pos);
    if (h->pinned_quat) (void)hipHostFree(h->pinned_quat);
    for (int i = 0; i < kRing; i++) if (h->ring_ev[i]) (void)hipEventDestroy(h->ring_ev[i]);
    for (int i = 0; i < kRing; i++) if (h->ring_ev_halo[i]) (void)hipEventDestroy(h->ring_ev_halo[i]);
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
    rc = enqueue_substep(h);
    if (!rc) rc = flush_v(h);
    return rc;
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
        if (h->comm && use_graph && h->halo_warm && !h->halo_graph_broken && uses_flag_sync(h)) {
            // flag path: the two streams' chains as two captured linear graphs, replayed side by side -- if the streams are served
            // by independent hardware queues (probed once) and the chains can be captured (else: eager for good, loudly)
            if ((rc = probe_queue_independence(h))) return rc;
            if (h->queues_independent) {
                rc = step_n_flag_graphs(h, n);
                if (!rc) return 0;
                fprintf(stderr, "[tetsim] halo graph capture failed (%s); falling back to eager halo stepping\n", h->err.c_str());
                h->halo_graph_broken = true;
                (void)hipGetLastError();
                rc = 0;
            }
        }
        if (!h->comm || !use_graph || !h->halo_warm || h->halo_graph_broken || uses_flag_sync(h)) {
            for (uint32_t i = 0; i < n && !rc; i++) rc = enqueue_substep(h);
            if (!rc) rc = flush_v(h);
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
            if (!rc) rc = flush_v(h);
            return rc;
        }
        it = h->graphs.emplace(n, exec).first;
    }
    HIPCHK(h, hipGraphLaunch(it->second, h->stream));
    return 0;
}

int tetsim_sync(tetsim_handle h) {
    if (!h) return TETSIM_EINVAL;
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    if (h->d_sync) {  // a bounded device-side wait that gave up (util_kernels.hip): the results since then are not to be trusted
        uint32_t err = 0;
        HIPCHK(h, hipMemcpy(&err, h->d_sync + 4, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(h, TETSIM_ECOMM, "a halo dependency was not signalled in time (device-side wait reached TETSIM_HALO_TIMEOUT_MS): a rank or a queue is stuck");
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
    HIPCHK(h, hipSetDevice(h->opt.device));
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
};
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
            if (!h->blk.lean) {  // constant-rest-shape bodies carry no shape state
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
    return hd;
}
int state_guard(tetsim_body* h) {
    if (h->partitioned && h->opt.part_count > 1) return fail(h, TETSIM_ESTATE, "save/load_state is supported on unpartitioned bodies only");
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
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    char* out = static_cast<char*>(blob);
    std::memcpy(out, &hd, sizeof(hd));
    out += sizeof(hd);
    std::vector<StateSection> secs;
    state_sections(h, secs);
    for (const StateSection& sec : secs) {
        if (sec.bytes) HIPCHK(h, hipMemcpy(out, sec.ptr, sec.bytes, hipMemcpyDeviceToHost));
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
    if (bytes < sizeof(in) + in.payload) return fail(h, TETSIM_EINVAL, "state blob is truncated");
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const char* src = static_cast<const char*>(blob) + sizeof(in);
    std::vector<StateSection> secs;
    state_sections(h, secs);
    for (const StateSection& sec : secs) {
        if (sec.bytes) HIPCHK(h, hipMemcpy(sec.ptr, src, sec.bytes, hipMemcpyHostToDevice));
        src += sec.bytes;
    }
    h->pred_any_dt = in.pred_any_dt != 0;
    h->dt_pred = in.dt_pred;
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
    h->info.num_vis_verts = nvis;
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

int tetsim_set_visual_triangles(tetsim_handle h, const int32_t* tri_ids, uint32_t ntri) {
    if (!h || (ntri && !tri_ids)) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->skin.nvis) return fail(h, TETSIM_ESTATE, "no visual mesh attached (tetsim_set_visual_mesh)");
    if (h->skin.vt_off) return fail(h, TETSIM_ESTATE, "visual triangles are already attached");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const uint32_t nvis = h->skin.nvis;
    std::vector<int4> tri(ntri);
    std::vector<uint32_t> off(nvis + 1, 0);
    for (uint32_t t = 0; t < ntri; t++) {
        for (int k = 0; k < 3; k++) {
            const int32_t v = tri_ids[3 * t + k];
            if (v < 0 || static_cast<uint32_t>(v) >= nvis) return fail(h, TETSIM_EINVAL, "triangle " + std::to_string(t) + " references a visual vertex outside the mesh");
            off[v + 1]++;
        }
        tri[t] = make_int4(tri_ids[3 * t], tri_ids[3 * t + 1], tri_ids[3 * t + 2], 0);
    }
    for (uint32_t v = 0; v < nvis; v++) off[v + 1] += off[v];
    std::vector<uint32_t> ent(3ull * ntri), fill(off.begin(), off.end() - 1);
    for (uint32_t t = 0; t < ntri; t++)   // triangle order, corner order: the order of the reference's accumulation
        for (int k = 0; k < 3; k++) ent[fill[tri_ids[3 * t + k]]++] = t;
    SkinDev& k = h->skin;
    int4* dt; uint32_t *doff, *dent;
    int rc;
    if ((rc = dev_alloc(h, &dt, ntri))) return rc;
    if ((rc = dev_alloc(h, &doff, off.size()))) return rc;
    if ((rc = dev_alloc(h, &dent, ent.size()))) return rc;
    if ((rc = dev_alloc(h, &k.out_vnrm, nvis))) return rc;
    if ((rc = upload(h, dt, tri))) return rc;
    if ((rc = upload(h, doff, off))) return rc;
    if ((rc = upload(h, dent, ent))) return rc;
    k.ntri = ntri; k.tri = dt; k.vt_tri = dent;
    k.vt_off = doff;
    return 0;
}

int tetsim_read_visual_vertex_normals(tetsim_handle h, float* normals_out) {
    if (!h || !normals_out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->skin.vt_off) return fail(h, TETSIM_ESTATE, "no visual triangles attached (tetsim_set_visual_triangles)");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    skin_launch(h->stream, h->skin, pjs ? h->pj.pos_final : h->nh.pos, pjs ? h->pj.quat : nullptr, !pjs);
    skin_launch_vertex_normals(h->stream, h->skin);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t n = h->skin.nvis;
    std::vector<float4> tmp(n);
    HIPCHK(h, hipMemcpy(tmp.data(), h->skin.out_vnrm, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) { normals_out[3 * i] = tmp[i].x; normals_out[3 * i + 1] = tmp[i].y; normals_out[3 * i + 2] = tmp[i].z; }
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
    struct Events : std::vector<hipEvent_t> {  // destroyed on every exit path
        using std::vector<hipEvent_t>::vector;
        ~Events() { for (hipEvent_t e : *this) if (e) (void)hipEventDestroy(e); }
    } ev(4ull * n + 2, nullptr);
    for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    hipEvent_t first_ev = ev[4ull * n], last_ev = ev[4ull * n + 1];
    HIPCHK(h, hipEventRecord(first_ev, h->stream));
    for (uint32_t i = 0; i < n; i++) {
        if (pjs && halo) {
            if ((rc = enqueue_phase_a(h, &ev[4 * i])) || (rc = enqueue_phase_b(h))) break;
        } else if (pjs && h->fused) {   // tet | fused x (n-1) | particle: what tetsim_step_n runs
            pj_fused_substep(h, i == 0, i + 1 == n, &ev[4 * i]);
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
    if (!rc) rc = flush_v(h);
    HIPCHK(h, hipEventRecord(last_ev, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    if (rc) return rc;
    float ms = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        float a = 0, b = 0, c = 0;
        if (pjs && h->fused && !halo) {
            // TETSIM_K_TET = the FUSED kernels (substeps 1..n-1: particle update + tet pass; the plain first tet kernel of the call is
            // not counted), TETSIM_K_VERTEX = the one particle kernel that ends the call
            if (i > 0) { HIPCHK(h, hipEventElapsedTime(&a, ev[4 * i], ev[4 * i + 1])); out->kernel_ms[TETSIM_K_TET] += a; out->launches[TETSIM_K_TET]++; }
            if (i + 1 == n) { HIPCHK(h, hipEventElapsedTime(&b, ev[4 * i + 2], ev[4 * i + 3])); out->kernel_ms[TETSIM_K_VERTEX] += b; out->launches[TETSIM_K_VERTEX]++; }
        } else if (pjs) {
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

}  // extern "C"
