// tetsim_api.hip -- C ABI of libtetsim_hip.so, part 1 (include/tetsim.h): handle lifecycle and stepping (stream / graph
// orchestration of the gfx950 kernels).  State read-back: tetsim_state.hip; visual mesh and grab: tetsim_visual.hip;
// measurement: tetsim_measure.hip.  See body.h for all translation units.
#include <mutex>

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
// Entry points over SEVERAL handles: the caller cannot know whose message to ask for, so the failing member's text is also what
// tetsim_last_error(NULL) returns on this thread.
int group_result(tetsim_body* const* hs, uint32_t count, int rc) {
    if (rc == TETSIM_OK || !hs) return rc;
    for (uint32_t i = 0; i < count; i++)
        if (hs[i] && !hs[i]->err.empty()) { g_create_error = "partition " + std::to_string(i) + ": " + hs[i]->err; return rc; }
    if (g_create_error.empty()) g_create_error = "bad argument";   // (a message fail(nullptr, ...) stored since group_begin -- "handles[i] must be ..." for a null member -- stays)
    return rc;
}
void group_begin(tetsim_body* const* hs, uint32_t count) {
    g_create_error.clear();
    if (hs) for (uint32_t i = 0; i < count; i++) if (hs[i]) hs[i]->err.clear();
}

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
        // (two-layer ghost regions: a first-layer ghost is advanced here too, so a grab of its particle applies to this copy as well)
        if (a < 0 && static_cast<size_t>(global) < h->g2l_ghost1.size()) a = h->g2l_ghost1[global];
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
    o->epoch = h->frame_epoch;
    static const int32_t poll_delay = [] { const char* e = lab_env("TETSIM_QUAD_POLL_DELAY"); return e ? atoi(e) : kQuadPollDelay; }();
    o->poll_delay = poll_delay;
    o->d_dt = dt;
    o->d_gravity = p.gravity;
    o->d_friction = p.friction;
    o->d_dev_compliance = p.devCompliance;
    o->d_vol_compliance = p.volCompliance;
}

// A fresh block of 65,536 sequence numbers for the partial sums of a persistent frame kernel (h->frame_epoch + the substep's index
// inside the launch; tetsim_step_n chunks longer calls): stale sums of an earlier launch never match.  Before the 32-bit counter wraps
// -- 65,535 blocks, minutes at interactive rates -- the numbers left in both buffers are wiped (in stream order, behind every kernel
// that reads them) and the count restarts: a sum of 65,536 launches ago can never pass for a fresh one.
int next_epoch_block(tetsim_body* h) {
    if (h->frame_epoch >= 0xfffe0000u) {
        if (h->partial_b && h->partial_slots) {
            HIPCHK(h, hipMemsetAsync(h->blk.partial, 0, h->partial_slots * sizeof(float4), h->stream));
            HIPCHK(h, hipMemsetAsync(h->partial_b, 0, h->partial_slots * sizeof(float4), h->stream));
        }
        if (h->nh_one_launch) HIPCHK(h, hipMemsetAsync(h->nh_sweep1.exchange, 0, static_cast<size_t>(h->nh.nv) * sizeof(float4), h->stream));
        h->frame_epoch = 1u;
    }
    h->frame_epoch += 65536u;
    return 0;
}

constexpr uint32_t kDirectLaunchTets = 150000u;   // from here on a body's substeps are launched kernel by kernel instead of replayed from a graph (tetsim_step_n)

// The parameters of this call into DevParams, in stream order.
int push_params(tetsim_body* h, double dt, const TetSimParams* params, bool reuse_ok) {
    if (!params) return fail(h, TETSIM_EINVAL, "params is null");
    if (!(dt > 0.0) || !std::isfinite(dt)) return fail(h, TETSIM_EINVAL, "dt must be a positive finite number");
    HIPCHK(h, hipSetDevice(h->opt.device));  // group stepping walks over handles that may live on different devices
    h->final_ghosts_fresh = false;           // (every stepping path comes through here: the ghosts' end-of-substep positions fetched for the visual mesh are stale)
    h->quat_stale = true;                    // (... and a lean-state body's quaternion array: ensure_quats)
    if (reuse_ok && h->params_known && !h->comm_stream && !h->partitioned) {   // (a halo queue keeps a copy of its own: always refreshed)
        // A host that keeps the reference's loop (main.js:79-84: simulate(dt, physicsParams) per substep) sends the same numbers again
        // and again: the copy and its event were most of what such a call cost (22.5 -> 7.8 us per tetsim_step on the Dragon,
        // profiles/r04_step_call_cost.txt).
        DevParams now;
        fill_params(h, dt, *params, &now);
        now.epoch = h->params_on_device.epoch;
        if (std::memcmp(&now, &h->params_on_device, sizeof now) == 0) { h->fork_needed = true; return 0; }
    }
    if (int rc = next_epoch_block(h)) return rc;
    h->epoch_block_fresh = reuse_ok;   // (only tetsim_step passes reuse_ok: its frame launch may use this block instead of taking another;
                                       //  every other caller's launches consume the block the upload names)
    fill_params(h, dt, *params, &h->params_on_device);
    h->params_known = true;
    // (a one-lane kernel whose arguments are the parameters: in stream order like the copy it replaces, without the copy engine)
    util_launch_set_params(h->stream, h->d_params, h->params_on_device);
    // the halo queue runs the boundary particles itself (enqueue_phase_a) and is ordered against the main queue only through
    // the semaphore words: it gets its own copy of the parameters, in ITS stream order
    if (h->comm_stream && h->d_params_halo) util_launch_set_params(h->comm_stream, h->d_params_halo, h->params_on_device);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(h, TETSIM_EHIP, std::string("kernel launch: ") + hipGetErrorString(le));
    h->fork_needed = true;  // whatever the caller did on the main stream since the last call must be visible to the boundary stream
    return 0;
}

// The parameters of a call whose ONE launch takes them by value (pjb_call_kernel: its first workgroup leaves them in DevParams for the
// kernels behind it): the same checks, a fresh block of sequence numbers, the host's record of what the device holds once the launch has
// run -- and no copy, no event, no pinned slot.
int stage_params(tetsim_body* h, double dt, const TetSimParams* params) {
    if (!params) return fail(h, TETSIM_EINVAL, "params is null");
    if (!(dt > 0.0) || !std::isfinite(dt)) return fail(h, TETSIM_EINVAL, "dt must be a positive finite number");
    HIPCHK(h, hipSetDevice(h->opt.device));
    h->final_ghosts_fresh = false;
    h->quat_stale = true;
    if (int rc = next_epoch_block(h)) return rc;
    h->epoch_block_fresh = false;
    fill_params(h, dt, *params, &h->params_on_device);
    h->params_known = true;
    h->fork_needed = true;
    return 0;
}

// ... for such a call: staged, unless its dt differs from the one the predictions on the device were made with -- the re-prediction kernel
// in front of the call reads DevParams, so that call uploads as every other does.
int params_for_launch(tetsim_body* h, double dt, const TetSimParams* params) {
    const bool repredict = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI && !h->pred_any_dt && static_cast<float>(dt) != h->dt_pred;
    if (int rc = repredict ? push_params(h, dt, params) : stage_params(h, dt, params)) return rc;
    return ensure_prediction(h, dt);
}

// ---- kernel sequencing ---------------------------------------------------------------------------------
void pj_tet(tetsim_body* h, hipEvent_t e0, hipEvent_t e1) {
    if (h->quad) pjq_launch_tet(h->stream, h->blk, e0, e1);
    else if (h->blocked) pjb_launch_tet(h->stream, h->blk, 0, h->blk.nb, e0, e1);
    else h->fast ? pj_launch_tet_fast(h->stream, h->pj, e0, e1) : pj_launch_tet_precise(h->stream, h->pj, e0, e1);
}
void pj_vertex(tetsim_body* h, uint32_t first, uint32_t count, hipEvent_t e0, hipEvent_t e1) {
    if (h->quad) pjq_launch_vertex(h->stream, h->blk, e0, e1);   // (all owned particles: quad bodies are unpartitioned)
    else if (h->blocked) pjb_launch_vertex(h->stream, h->blk, first, count, e0, e1);
    else h->fast ? pj_launch_vertex_fast(h->stream, h->pj, first, count, e0, e1) : pj_launch_vertex_precise(h->stream, h->pj, first, count, e0, e1);
}
// One substep of a fused body inside a run of substeps with one dt (HISTORY.md 5.4):
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
        { PJBlk kf = k; kf.vel = nullptr; pjb_launch_tet_fused(h->stream, kf, e ? e[0] : nullptr, e ? e[1] : nullptr); }   // (its particle update is never a call's last: the velocity stays in registers)
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
// last = the call's last sweep: `volError` (Softbody.js:163, reset by every simulate()) is what THAT sweep leaves behind; an earlier
// sweep's per-tet values would be overwritten unread -- 8 B per tet-solve of dead stores, left out (a null pointer)
void nh_sweep(tetsim_body* h, bool fold, bool last, bool one_launch) {
    NHDev nh = h->nh;
    if (!last) nh.vol_err = nullptr;
    if (one_launch && h->nh_one_launch) {   // clustered FAST: every colour in ONE launch, particles handed on with their stamp (nh_kernels.inc: nh_sweep1_kernel)
        nh_launch_sweep1_fast(h->stream, nh, h->nh_sweep1, fold, h->nh_sub_index, h->nh_epoch_arg);
        return;
    }
    if (!h->cluster_launch.empty()) {
        for (const NHClusterLaunch& L : h->cluster_launch) h->fast ? nh_launch_cluster_fast(h->stream, nh, L, fold) : nh_launch_cluster_precise(h->stream, nh, L, fold);
        return;
    }
    if (h->nh_frame && h->fast) {
        // small FAST bodies: the stepwise twin of their single-workgroup launch (nh_kernels.inc) makes ITS choice for every body's
        // piece of a level -- narrow: four lanes per tet, wide: one -- so that the two agree bit for bit
        const uint32_t bodies = h->nh_frame_launch.bodies;
        for (size_t l = 0; l + 1 < h->level_off.size(); l++)
            for (uint32_t b = 0; b < bodies; b++) {
                const uint32_t first = h->nh_seg[2u * (l * bodies + b)], count = h->nh_seg[2u * (l * bodies + b) + 1u] - first;
                if (count <= kNHQuadLevelFast) nh_launch_level4_fast(h->stream, nh, first, count);
                else nh_launch_level_fast(h->stream, nh, first, count);
            }
        return;
    }
    // A level of up to 8,192 tets is a few waves wherever it runs: its time is the dependent chain of ONE tet solve, and one tet on four
    // lanes (nh_level4_kernel: the same bits in PRECISE, tolerance-level re-association in FAST) shortens it -- COLOURED bodies of
    // 25 k-200 k tets: 117-129 -> 102-115 us per substep bit-exact, 88-100 -> 80-93 FAST; the 1 M-tet lattice's levels (32 k tets)
    // fill the chip and keep one lane per tet (190 against 234 us).
    constexpr uint32_t level4_max = 8192u;
    for (size_t l = 0; l + 1 < h->level_off.size(); l++) {
        const uint32_t first = h->level_off[l], count = h->level_off[l + 1] - first;
        if (count <= level4_max) h->fast ? nh_launch_level4_fast(h->stream, nh, first, count) : nh_launch_level4_precise(h->stream, nh, first, count);
        else h->fast ? nh_launch_level_fast(h->stream, nh, first, count) : nh_launch_level_precise(h->stream, nh, first, count);
    }
}

// Persistent frame kernels need every workgroup of their grid resident at once (and the one-launch calls of round 6 keep waiting workgroups
// resident: they take part in the turns below).  Bodies of at most half the device's slots fit side
// by side; once a body that needs MORE lives on a device (`frame_exclusive`), the frame launches of that device's bodies take turns:
// a launch waits for the previous one (of another body) to finish.  Nothing happens, and nothing is paid, without such a body.
// The count is taken when the exclusive body is CREATED (frame_turn_enter, from create_polar) -- not at its first launch: persistent
// launches of other bodies issued before that were untracked, and the exclusive body's first launch could land half resident beside
// one of them (advisor, round 4).  The first launch after the count leaves zero therefore drains the device once; from then on
// every launch records `done`.  The count is given back when the body dies or falls back to one kernel per substep.
struct FrameTurn {
    std::mutex m;
    hipEvent_t done = nullptr;       // recorded behind the last frame launch on this device
    const tetsim_body* last = nullptr;
    int exclusive_bodies = 0;
    bool drain = false;              // launches issued while the count was zero may still be in flight: wait for the device once
};
FrameTurn g_frame_turn[64];
FrameTurn& frame_turn(const tetsim_body* h) { return g_frame_turn[h->opt.device & 63]; }
void frame_turn_enter(tetsim_body* h) {
    if (!h->frame_exclusive || h->frame_turn_counted) return;
    FrameTurn& t = frame_turn(h);
    std::lock_guard<std::mutex> lock(t.m);
    if (t.exclusive_bodies++ == 0) t.drain = true;
    h->frame_turn_counted = true;
}
void frame_turn_leave(tetsim_body* h) {
    FrameTurn& t = frame_turn(h);
    std::lock_guard<std::mutex> lock(t.m);
    if (h->frame_turn_counted && --t.exclusive_bodies == 0) {
        if (t.done) { (void)hipEventDestroy(t.done); t.done = nullptr; }   // (an event destroyed while pending is released when it completes)
        t.last = nullptr;
        t.drain = false;
    }
    h->frame_turn_counted = false;
    if (t.last == h) t.last = nullptr;   // (the caller has drained or abandoned its stream: nothing to wait for)
}
// `launch` puts a persistent frame kernel of body h into h->stream (a graph replay or a direct launch); its turn is taken first if needed
template <class F>
int launch_in_turn(tetsim_body* h, F&& launch) {
    FrameTurn& t = frame_turn(h);
    std::lock_guard<std::mutex> lock(t.m);
    if (t.exclusive_bodies == 0) return launch();
    if (t.drain) { HIPCHK(h, hipDeviceSynchronize()); t.drain = false; }
    else if (t.done && t.last && t.last != h) HIPCHK(h, hipStreamWaitEvent(h->stream, t.done, 0));
    if (int rc = launch()) return rc;
    if (!t.done) HIPCHK(h, hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(t.done, h->stream));
    t.last = h;
    return 0;
}
// the persistent frame kernel of a small polar body for n substeps; epoch 0 = the block DevParams::epoch names (graph capture)
int launch_frame_kernel(tetsim_body* h, uint32_t n, uint32_t epoch) {
    PJBlk k = h->blk;
    k.epoch = epoch;
    if (h->quad) pjq_launch_frame(h->stream, k, n, h->d_block_tile, h->frame_blocks, h->frame_local, h->blk.partial, h->partial_b, h->d_frame_err, halo_timeout_ms(h), h->params_on_device, h->d_params);
    else pjb_launch_frame(h->stream, k, n, h->d_block_tile, h->frame_blocks, h->frame_local, h->blk.partial, h->partial_b, h->d_frame_err, halo_timeout_ms(h), h->params_on_device, h->d_params);
    const hipError_t le = hipGetLastError();
    return le == hipSuccess ? 0 : fail(h, TETSIM_EHIP, std::string("kernel launch: ") + hipGetErrorString(le));
}
// the single-workgroup frame kernel of a small Neo-Hookean body for n substeps, the call's parameters among its arguments
int launch_nh_frame_kernel(tetsim_body* h, uint32_t n) {
    h->fast ? nh_launch_frame_fast(h->stream, h->nh, h->nh_frame_launch, n, h->params_on_device, h->d_params)
            : nh_launch_frame_precise(h->stream, h->nh, h->nh_frame_launch, n, h->params_on_device, h->d_params);
    const hipError_t le = hipGetLastError();
    return le == hipSuccess ? 0 : fail(h, TETSIM_EHIP, std::string("kernel launch: ") + hipGetErrorString(le));
}

// one substep's launches (parameters already on the device)
// first / last: position inside a run of substeps enqueued back to back with one dt (NEOHOOKEAN_GS fuses the particle pass
// that ends a substep with the prediction that starts the next one)
int enqueue_substep(tetsim_body* h, bool first, bool last) {
    if (h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI) {
        if (h->deep && !h->p2p) return fail(h, TETSIM_ESTATE, "a body with a two-layer ghost region steps through the peer-to-peer halo only: call tetsim_halo_p2p_export / _connect first");
        // Between two substeps of one call nobody reads the velocity array: the particle kernel hands the velocity on to the next
        // prediction in registers; re-prediction after a dt change, read-back, checkpoints and the next call all sit behind the call's
        // LAST substep.  Its store -- 16 B per particle and substep -- is left out everywhere but there (blocked kernels: a null pointer).
        if (has_transport(h)) {
            h->vel_dead = !last;
            int rc = enqueue_phase_a(h);
            h->vel_dead = false;
            if (!rc) rc = enqueue_phase_b(h);
            if (rc) return rc;
        } else if (h->fused) {
            pj_fused_substep(h, first, last, nullptr);
        } else {
            pj_tet(h);
            if (h->blocked && !h->quad && !last) {
                PJBlk k = h->blk;
                k.vel = nullptr;   // (see above)
                pjb_launch_vertex(h->stream, k, 0, h->pj.nv_owned);
            } else pj_vertex(h, 0, h->pj.nv_owned);
        }
    } else {
        // clustered schedules fold the particle pass BETWEEN two substeps of a run into the next sweep: the lane of the first cluster
        // to touch a particle finishes the previous substep and predicts the next for it while loading it (nh_kernels.inc:
        // fold_particle) -- one kernel and one launch boundary less per substep, the same operations per particle
        if (first) h->fast ? nh_launch_predict_fast(h->stream, h->nh) : nh_launch_predict_precise(h->stream, h->nh);
        h->nh_sub_index = first ? 0u : h->nh_sub_index + 1u;
        nh_sweep(h, !first && h->nh_fold, last, true);
        if (last) h->fast ? nh_launch_post_fast(h->stream, h->nh) : nh_launch_post_precise(h->stream, h->nh);
        else if (h->nh_fold) h->fast ? nh_launch_post_predict_list_fast(h->stream, h->nh, h->d_nh_untouched, h->nh_untouched)
                                     : nh_launch_post_predict_list_precise(h->stream, h->nh, h->d_nh_untouched, h->nh_untouched);
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
        if (h->deep)
            return fail(h, TETSIM_ESTATE, "dt changed between calls on a body with a two-layer ghost region (TETSIM_FLAG_DEEP_GHOSTS): its neighbours hold predictions "
                                          "made with the old dt for up to two substeps; keep dt fixed");
        if (h->p2p && !h->comm && h->group.empty())
            return fail(h, TETSIM_ESTATE, "dt changed between calls on a body whose only transport is the peer-to-peer halo: nothing refreshes the neighbours' ghost "
                                          "predictions (keep dt fixed, or connect on top of an RCCL communicator, which carries the refresh exchange)");
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
                y.flag = h->d_sync + 3; y.error = h->d_sync + 4; y.timeout_ms = halo_timeout_ms(h);
                pjb_launch_signal(h->stream, y);
                pjb_launch_wait(h->comm_stream, y);
            }
            if (h->comm) {
                int rc = enqueue_phase_b(h, true);
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
    if (rc == TETSIM_OK && a.vis_verts && a.num_vis_verts) {
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
    if ((o.flags & TETSIM_FLAG_LEAN_STATE) && o.solver != TETSIM_SOLVER_POLAR_JACOBI)
        return fail(nullptr, TETSIM_EINVAL, "TETSIM_FLAG_LEAN_STATE applies to TETSIM_SOLVER_POLAR_JACOBI only");
    if ((o.flags & TETSIM_FLAG_LEAN_STATE) && (o.flags & (TETSIM_FLAG_CONSTANT_REST_SHAPE | TETSIM_FLAG_DEEP_GHOSTS)))
        return fail(nullptr, TETSIM_EINVAL, "TETSIM_FLAG_LEAN_STATE excludes TETSIM_FLAG_CONSTANT_REST_SHAPE and TETSIM_FLAG_DEEP_GHOSTS");

    tetsim_body* h = new tetsim_body();
    h->opt = o;
    h->opt.vert_owner = nullptr;  // not retained
    if (o.tet_colour && nt) h->tet_colour.assign(o.tet_colour, o.tet_colour + nt);
    h->opt.tet_colour = nullptr;
    h->fast = o.precision == TETSIM_FAST;
    h->timeout_ms = halo_timeout_ms(nullptr);
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
    g_stream_generation++;
    if (!hipok(hipEventCreate(&h->ev_a), "hipEventCreate") || !hipok(hipEventCreate(&h->ev_b), "hipEventCreate")) return bail(TETSIM_EHIP);
    for (int i = 0; i < 2; i++)
        if (!hipok(hipEventCreateWithFlags(&h->ev_boundary2[i], hipEventDisableTiming), "hipEventCreate") ||
            !hipok(hipEventCreateWithFlags(&h->ev_packed2[i], hipEventDisableTiming), "hipEventCreate") ||
            !hipok(hipEventCreateWithFlags(&h->ev_sent2[i], hipEventDisableTiming), "hipEventCreate")) return bail(TETSIM_EHIP);
    if (!hipok(hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming), "hipEventCreate")) return bail(TETSIM_EHIP);
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
    // an in-process group: the other members' queues may still hold transfers into THIS body's ghost ranges; and they forget this body
    // (tetsim_group_step_n refuses a group with a member gone, tetsim_save_state / _load_state skip it)
    for (tetsim_body* g : h->group) {
        if (!g || g == h) continue;
        (void)hipSetDevice(g->opt.device);
        if (g->stream) (void)hipStreamSynchronize(g->stream);
        if (g->comm_stream) (void)hipStreamSynchronize(g->comm_stream);
        for (tetsim_body*& x : g->group) if (x == h) x = nullptr;
    }
    (void)hipSetDevice(h->opt.device);
    if (h->blk.trace && lab_env("TETSIM_DEBUG_TRACE")) {
        std::vector<unsigned long long> tr(8ull * h->blk.nb);
        if (hipMemcpy(tr.data(), h->blk.trace, tr.size() * sizeof(tr[0]), hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(lab_env("TETSIM_DEBUG_TRACE"), "wb")) { fwrite(tr.data(), sizeof(tr[0]), tr.size(), f); fclose(f); }
    }
#ifdef TETSIM_ABLATION
    if (h->blk.iter_hist && lab_env("TETSIM_DEBUG_ITER_HIST")) {   // one text line per body, appended: "<tets> <particles> <278 counters>"
        std::vector<unsigned long long> hs(278);
        if (hipMemcpy(hs.data(), h->blk.iter_hist, hs.size() * sizeof(hs[0]), hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(lab_env("TETSIM_DEBUG_ITER_HIST"), "a")) {
                fprintf(f, "%u %u", h->info.num_elems, h->info.num_particles);
                for (unsigned long long x : hs) fprintf(f, " %llu", x);
                fprintf(f, "\n");
                fclose(f);
            }
    }
#endif
    frame_turn_leave(h);
    // graphs first: a captured halo graph holds RCCL work, and ncclCommDestroy waits for (hangs on) captured work that still exists
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    drop_flag_graphs(h);
    for (PeerLink& l : h->links) for (void* m : l.ipc) if (m) (void)hipIpcCloseMemHandle(m);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->pinned_pos) (void)hipHostFree(h->pinned_pos);
    if (h->pinned_quat) (void)hipHostFree(h->pinned_quat);
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
    if (h->frame) {
        // small polar bodies: a single substep is ONE launch too -- the persistent frame kernel for n = 1, its parameters (and with them a
        // block of sequence numbers of its own) among its arguments: no upload, moving grab or not
        if (int rc = params_for_launch(h, dt, params)) return rc;
        return launch_in_turn(h, [&]() -> int { return launch_frame_kernel(h, 1u, 0u); });
    }
    if (h->nh_frame) {
        // small Neo-Hookean bodies: also a single substep is ONE single-workgroup launch (a host that keeps the reference's loop,
        // main.js:79-84, pays one enqueue per substep instead of one per level: 34 on the Dragon); tetsim_profile keeps the level kernels
        if (int rc = params_for_launch(h, dt, params)) return rc;
        return launch_nh_frame_kernel(h, 1u);
    }
    int rc = push_params(h, dt, params, true);   // (unchanged parameters stay where they are)
    if (rc) return rc;
    if ((rc = ensure_prediction(h, dt))) return rc;
    if (h->nh_one_launch) {   // (the parameters on the device may be the previous call's: this launch brings its own block of stamps, like the frame kernel above)
        if (!h->epoch_block_fresh && (rc = next_epoch_block(h))) return rc;
        h->epoch_block_fresh = false;
        h->nh_epoch_arg = h->frame_epoch;
    }
    if (h->nh_one_launch) rc = launch_in_turn(h, [&]() -> int { return enqueue_substep(h); });   // (see tetsim_step_n)
    else rc = enqueue_substep(h);
    h->nh_epoch_arg = 0u;
    if (!rc) rc = flush_v(h);
    return rc;
}

int tetsim_step_n(tetsim_handle h, uint32_t n, double dt, const TetSimParams* params) {
    if (!h) return TETSIM_EINVAL;
    if (!h->group.empty()) return fail(h, TETSIM_ESTATE, "this body belongs to an in-process group: step it with tetsim_group_step_n");
    if (n == 0) return 0;
    while (h->frame && n > 32768u) {   // (a persistent launch numbers its substeps inside one block of 65,536 sequence numbers)
        if (int rc = tetsim_step_n(h, 32768u, dt, params)) return rc;
        n -= 32768u;
    }
    if (h->pj_one_launch) {   // (the one-launch call of large polar bodies: n x (tiles + particle workgroups) in one grid, stamps inside one block)
        const uint64_t per_sub = (static_cast<uint64_t>(h->blk.nb) + 7u) / 8u * 8u + (h->blk.nv_owned + kBlockTile - 1u) / kBlockTile + 8u;
        const uint32_t most = static_cast<uint32_t>(std::max<uint64_t>(1u, std::min<uint64_t>(8192u, 0x7fffffffull / per_sub)));
        while (n > most) {
            if (int rc = tetsim_step_n(h, most, dt, params)) return rc;
            n -= most;
        }
    }
    if (h->nh_one_launch) {            // (the one-launch sweep stamps substep x colour inside one block too)
        const uint32_t most = 65000u / h->nh_sweep1.ncolours;
        while (n > most) {
            if (int rc = tetsim_step_n(h, most, dt, params)) return rc;
            n -= most;
        }
    }
    HIPCHK(h, hipSetDevice(h->opt.device));
    if (h->nh_call && h->nh_one_launch && !has_transport(h)) {
        // clustered FAST Neo-Hookean bodies: prediction | the sweeps of all n substeps in ONE launch (nh_kernels.inc: nh_call_kernel) | the kernel
        // that ends the call -- three direct launches, the first of which brings the call's parameters along
        if (int rc = params_for_launch(h, dt, params)) return rc;
        return launch_in_turn(h, [&]() -> int {
            nh_launch_predict_value_fast(h->stream, h->nh, h->params_on_device, h->d_params);
            nh_launch_call_fast(h->stream, h->nh, h->nh_sweep1, n);
            nh_launch_post_fast(h->stream, h->nh);
            const hipError_t le = hipGetLastError();
            return le == hipSuccess ? 0 : fail(h, TETSIM_EHIP, std::string("kernel launch: ") + hipGetErrorString(le));
        });
    }
    if (h->nh_frame) {   // small Neo-Hookean bodies: the whole call is ONE single-workgroup launch with every particle in LDS (nh_kernels.inc: nh_frame_kernel)
        if (int rc = params_for_launch(h, dt, params)) return rc;
        return launch_nh_frame_kernel(h, n);
    }
    if ((h->frame || h->pj_one_launch) && !has_transport(h)) {
        // Calls that are ONE kernel -- small polar bodies: the persistent frame kernel, every tile's workgroup resident for the n substeps
        // (pjb_frame_kernel, pjq_frame_kernel); large ones: the n substeps' tiles and particles in one grid, handed on by stamped data
        // (pjb_call_kernel) -- are launched directly, one kernel needs no graph, with the call's parameters among the arguments.
        if (int rc = params_for_launch(h, dt, params)) return rc;
        // (in turn with an exclusive frame-kernel body, if one lives on this device: see the graph launch below)
        return launch_in_turn(h, [&]() -> int {
            if (h->frame) return launch_frame_kernel(h, n, 0u);
            pjb_launch_call(h->stream, h->blk, n, h->d_substep_err, halo_timeout_ms(h), h->params_on_device, h->d_params);
            const hipError_t le = hipGetLastError();
            return le == hipSuccess ? 0 : fail(h, TETSIM_EHIP, std::string("kernel launch: ") + hipGetErrorString(le));
        });
    }
    int rc = push_params(h, dt, params);
    if (rc) return rc;
    if ((rc = ensure_prediction(h, dt))) return rc;
    if (has_transport(h)) {
        // RCCL bodies: the first call runs eagerly (RCCL sets its connections up on first use, which must not happen inside a
        // capture); afterwards the n substeps -- both streams, the grouped send/recv included -- are one captured graph:
        // eager cross-stream dependencies cost ~10 us each on this stack and there are three per substep on the halo's
        // critical path.  TETSIM_HALO_GRAPH=0 keeps everything eager.
        const bool use_graph = h->halo_use_graph;
        const bool own_rank = h->group.empty() && (h->comm || h->p2p);   // one rank per process: RCCL and / or the peer-to-peer halo
        if (own_rank && use_graph && h->halo_warm && !h->halo_graph_broken && uses_flag_sync(h)) {
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
        if (!own_rank || !h->comm || !use_graph || !h->halo_warm || h->halo_graph_broken || uses_flag_sync(h)) {
            for (uint32_t i = 0; i < n && !rc; i++) rc = enqueue_substep(h, i == 0, i + 1 == n);
            if (!rc) rc = flush_v(h);
            h->halo_warm = true;
            return rc;
        }
        if (h->halo_pending) {  // leftovers of eager calls: join them first
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_sent2[h->halo_parity ^ 1u], 0));
            h->halo_pending = false;
        }
    }
    if (!has_transport(h) && !h->nh_one_launch && h->info.num_elems >= kDirectLaunchTets &&
        (h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI || h->info.num_levels <= 32u)) {   // (a Gauss-Seidel schedule of hundreds of small levels stays a graph)
        // One queue and kernels of tens of microseconds each: the substeps' kernels go into the stream one by one.  A graph's replay ends
        // with a completion signal that the next call's first kernel waits ~9 us for; kernel follows kernel without a gap, and the host
        // (2-3 us per launch) stays ahead.  Small bodies keep their graphs: their kernels are shorter than a launch call.
        for (uint32_t i = 0; i < n && !rc; i++) rc = enqueue_substep(h, i == 0, i + 1 == n);
        return rc;
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
            for (uint32_t i = 0; i < n && !rc; i++) rc = enqueue_substep(h, i == 0, i + 1 == n);
            if (!rc) rc = flush_v(h);
            return rc;
        }
        it = h->graphs.emplace(n, exec).first;
    }
    // (... and so do the calls that run as ONE launch of stamped hand-overs -- pjb_call_kernel, nh_sweep1 / nh_call_kernel: their waiting
    // workgroups hold slots too.  Beside a body whose persistent launch needs most of the device resident AT ONCE the two deadlock until a
    // wait gives up: the frame kernel's resident tiles fill an XCD waiting for tiles that find no slot, the call kernel's workgroups fill the
    // rest waiting for a workgroup that is next in line on THAT XCD -- tools/soak.py, round 6.  Nothing is paid without such a body.)
    if (h->nh_one_launch)
        return launch_in_turn(h, [&]() -> int { HIPCHK(h, hipGraphLaunch(it->second, h->stream)); return 0; });
    HIPCHK(h, hipGraphLaunch(it->second, h->stream));
    return 0;
}

int tetsim_sync(tetsim_handle h) {
    if (!h) return TETSIM_EINVAL;
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    if (h->d_frame_err) {   // persistent frame kernel: a tile's wait for its neighbours' partial sums gave up (never in a correct run)
        uint32_t err = 0;
        HIPCHK(h, hipMemcpy(&err, h->d_frame_err, sizeof err, hipMemcpyDeviceToHost));
        if (err) {
            HIPCHK(h, hipMemset(h->d_frame_err, 0, sizeof err));
            h->frame = false;   // step with one kernel per substep from now on
            frame_turn_leave(h);   // ... and no longer makes the device's persistent launches take turns
            for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            return fail(h, TETSIM_EHIP, "persistent frame kernel: a tile waited in vain for a neighbour tile's partial sums (workgroups not co-resident?); "
                                        "the state since then is invalid; this body falls back to one kernel per substep");
        }
    }
    if (h->pj_one_launch) {   // one-launch substep: a particle wave waited in vain for a tile's partial sums (never in a correct run)
        uint32_t err = 0;
        HIPCHK(h, hipMemcpy(&err, h->d_substep_err, sizeof err, hipMemcpyDeviceToHost));
        if (err) {
            HIPCHK(h, hipMemset(h->d_substep_err, 0, sizeof err));
            h->pj_one_launch = false;   // two kernels per substep from now on
            for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            return fail(h, TETSIM_EHIP, "one-launch substep: a particle wave waited in vain for a tile's partial sums (workgroups not dispatched in grid order?); "
                                        "the state since then is invalid; this body falls back to two kernels per substep");
        }
    }
    if (h->nh_one_launch) {   // one-launch Gauss-Seidel sweep: a cluster waited in vain for a particle of an earlier colour (never in a correct run)
        uint32_t err = 0;
        HIPCHK(h, hipMemcpy(&err, h->nh_sweep1.error, sizeof err, hipMemcpyDeviceToHost));
        if (err) {
            HIPCHK(h, hipMemset(h->nh_sweep1.error, 0, sizeof err));
            h->nh_one_launch = false;   // one launch per colour from now on
            for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            return fail(h, TETSIM_EHIP, "one-launch Gauss-Seidel sweep: a cluster waited in vain for a particle of an earlier colour (workgroups not dispatched in grid order?); "
                                        "the state since then is invalid; this body falls back to one launch per colour");
        }
    }
    if (h->d_sync) {  // a bounded device-side wait that gave up (util_kernels.hip): the results since then are not to be trusted
        uint32_t err = 0;
        HIPCHK(h, hipMemcpy(&err, h->d_sync + 4, sizeof err, hipMemcpyDeviceToHost));
        if (err) {
            // Either a peer is stuck, or the two chains of a graph replay sat in ONE hardware queue after all (a wait kernel at its
            // head blocks the kernel that would raise its word; HIP may re-map streams when other streams appear in the process).
            // The substeps since the time-out used stale data: the caller restores a checkpoint (tetsim_load_state) or gives up.
            // What this library can do is not walk into it again: no more graph replay for this body, the eager path orders every
            // wait behind its signal in submission order and stays live on a shared queue.  The error word is cleared so that the
            // NEXT synchronisation judges the eager substeps on their own.
            const bool replayed = !h->flag_graphs.empty();
            drop_flag_graphs(h);
            h->halo_graph_broken = true;
            h->queues_independent = false;
            HIPCHK(h, hipMemset(h->d_sync, 0, kSyncWords * sizeof(uint32_t)));
            h->v_pending = false;
            return fail(h, TETSIM_ECOMM, std::string("a halo dependency was not signalled in time (device-side wait reached TETSIM_HALO_TIMEOUT_MS): a rank or a queue is stuck; "
                                                     "the state since then is invalid") + (replayed ? " -- graph replay of the halo chains is disabled for this body from now on (eager stepping)" : ""));
        }
    }
    return 0;
}

}  // extern "C"
