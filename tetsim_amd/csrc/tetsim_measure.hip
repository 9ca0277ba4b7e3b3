// tetsim_measure.hip -- C ABI, part 4 (include/tetsim.h): measurement entry points (per-kernel profile with the kernels' own
// begin / end events, kernel timing loops, device copy bandwidth).  Nothing here is on the stepping path.  See body.h.
#include "body.h"

using namespace tetsim;

extern "C" {

int tetsim_profile(tetsim_handle h, uint32_t n, double dt, const TetSimParams* params, TetSimProfile* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    // a body with an RCCL halo: every rank calls this together (the substeps exchange halos as usual); what is timed is the
    // interior tet kernel and the particle kernel of the two-stream choreography
    const bool halo = has_transport(h);
    if (halo && (!h->comm || !h->blocked || h->blk.nb == h->blk.nb_interior || getenv("TETSIM_DEBUG_ONE_STREAM")))
        return fail(h, TETSIM_ESTATE, "profiling a partitioned body needs the RCCL transport and the blocked formulation (in-process groups: use rocprofv3)");
    if (halo && (h->blk.nb_interior == 0 || h->pj.nv_owned == h->pj.nv_boundary))
        // (checked BEFORE anything is stepped: enqueue_phase_a elides a launch of zero tiles / zero particles, its events would
        // never be recorded and the elapsed-time read would fail after the body had already advanced n substeps)
        return fail(h, TETSIM_ESTATE, "nothing to time: this partition has no interior tiles or no interior particles (everything is next to the halo)");
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

// The measured memory peak.  The probe is tuned ONCE per (device, kind, size class) at its first use -- plain or non-temporal accesses, 4 or 8
// of them in flight per lane, 2 / 4 / 8 / 16 / 32 workgroups per CU or one per chunk: 24 candidates, three launches each -- and the winner
// is what is timed.
int tetsim_measure_stream_bandwidth(int32_t device, uint64_t bytes, uint32_t reps, int32_t kind, double* gbps_out) {
    if (!gbps_out || bytes < 16 || reps == 0 || kind < 0 || kind > 2) return fail(nullptr, TETSIM_EINVAL, "bad argument");
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess) { (void)fail(nullptr, TETSIM_EHIP, std::string(what) + ": " + hipGetErrorString(e)); return false; } return true; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, TETSIM_ENODEVICE, "no HIP device available");
    if (!chk(hipSetDevice(device), "hipSetDevice")) return TETSIM_EHIP;
    hipDeviceProp_t prop;
    if (!chk(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return TETSIM_EHIP;
    const uint32_t cus = static_cast<uint32_t>(std::max(prop.multiProcessorCount, 1));
    const uint64_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = TETSIM_OK;
    if (!chk(hipMalloc(reinterpret_cast<void**>(&a), n * 16), "hipMalloc") || !chk(hipMalloc(reinterpret_cast<void**>(&b), n * 16), "hipMalloc")) rc = TETSIM_ENOMEM;
    if (!rc && (!chk(hipStreamCreate(&s), "hipStreamCreate") || !chk(hipEventCreate(&e0), "hipEventCreate") || !chk(hipEventCreate(&e1), "hipEventCreate"))) rc = TETSIM_EHIP;
    auto timed = [&](bool nt, uint32_t unroll, uint32_t grid, uint32_t launches, float* ms) {
        (void)hipEventRecord(e0, s);
        for (uint32_t r = 0; r < launches; r++) util_launch_stream(s, kind, nt, unroll, grid, (r & 1) ? b : a, (r & 1) ? a : b, n);
        (void)hipEventRecord(e1, s);
        return chk(hipEventSynchronize(e1), "hipEventSynchronize") && chk(hipEventElapsedTime(ms, e0, e1), "hipEventElapsedTime");
    };
    if (!rc) {
        (void)hipMemsetAsync(a, 0x3c, n * 16, s);
        (void)hipMemsetAsync(b, 0x3c, n * 16, s);
        struct Choice { bool nt; uint32_t unroll, grid; };
        static std::map<uint64_t, Choice> tuned;   // (device, kind, log2 size) -> what won
        uint32_t lg = 0;
        while ((bytes >> lg) > 1) lg++;
        const uint64_t key = (static_cast<uint64_t>(device) << 16) | (static_cast<uint64_t>(kind) << 8) | lg;
        auto it = tuned.find(key);
        if (it == tuned.end()) {
            Choice best = {false, 4u, 0u};
            float best_ms = 1e30f;
            for (int nt = 0; nt < 2 && !rc; nt++)
                for (uint32_t unroll : {4u, 8u})
                    for (uint32_t per_cu : {2u, 4u, 8u, 16u, 32u, 0u}) {
                        float ms = 0.0f;
                        if (!timed(nt != 0, unroll, per_cu * cus, 1, &ms) || !timed(nt != 0, unroll, per_cu * cus, 4, &ms)) { rc = TETSIM_EHIP; break; }
                        if (ms < best_ms) { best_ms = ms; best = {nt != 0, unroll, per_cu * cus}; }
                    }
            if (!rc) it = tuned.emplace(key, best).first;
        }
        float ms = 0.0f;
        if (getenv("TETSIM_DEBUG_STREAM_PROBE")) fprintf(stderr, "[tetsim] stream probe kind %d, %llu bytes: %s accesses, %u per lane, grid %u\n", kind, static_cast<unsigned long long>(bytes), it->second.nt ? "non-temporal" : "plain", it->second.unroll, it->second.grid);
        // three batches of `reps` launches, the fastest batch counts: the yardstick is what the memory system sustains at its best
        float best = 1e30f;
        if (!rc && !timed(it->second.nt, it->second.unroll, it->second.grid, 3, &ms)) rc = TETSIM_EHIP;
        for (int batch = 0; batch < 3 && !rc; batch++) {
            if (timed(it->second.nt, it->second.unroll, it->second.grid, reps, &ms)) best = std::min(best, ms); else rc = TETSIM_EHIP;
        }
        if (!rc) *gbps_out = (kind == 0 ? 2.0 : 1.0) * static_cast<double>(n * 16) * reps / (static_cast<double>(best) * 1.0e6);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}
int tetsim_measure_copy_bandwidth(int32_t device, uint64_t bytes, uint32_t reps, double* gbps_out) {
    return tetsim_measure_stream_bandwidth(device, bytes, reps, 0, gbps_out);
}


}  // extern "C"
