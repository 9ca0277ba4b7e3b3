// tetsim_comm.hip -- C ABI, multi-GPU set-up (include/tetsim.h): the RCCL communicator of a partitioned body, its self-test and probe, the halo
// plan and the host-side halo export / import used by tests.  The per-substep choreography is tetsim_halo.hip.
#include "body.h"

using namespace tetsim;

extern "C" {

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

int tetsim_comm_info(tetsim_handle h, TetSimCommInfo* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->comm) return fail(h, TETSIM_ESTATE, "no communicator (call tetsim_comm_init first)");
    std::memset(out, 0, sizeof(*out));
    int n = 0, r = -1;
    ncclResult_t e = g_rccl.CommCount(h->comm, &n);
    if (e == ncclSuccess) e = g_rccl.CommUserRank(h->comm, &r);
    if (e != ncclSuccess) return rccl_fail(h, e, "ncclCommCount / ncclCommUserRank");
    out->rccl_ranks = n;
    out->rccl_rank = r;
    out->neighbours = static_cast<uint32_t>(h->neigh.size());
    for (const NeighDev& nb : h->neigh) {
        out->send_bytes_per_substep += 16ull * nb.send_count;
        out->recv_bytes_per_substep += 16ull * nb.recv_count;
        out->max_message_bytes = std::max<uint64_t>(out->max_message_bytes, 16ull * std::max(nb.send_count, nb.recv_count));
    }
    out->loopback = h->loopback ? 1 : 0;
    out->p2p = h->p2p ? 1 : 0;
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
    HIPCHK(h, hipSetDevice(h->opt.device));
    NeighDev& nb = h->neigh[n];
    if (!nb.send_count) return 0;
    util_launch_gather4(h->stream, h->pj.pos_pred, nb.send_idx, nb.send_buf, nb.send_count);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out_xyzw, nb.send_buf, nb.send_count * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
int tetsim_halo_import(tetsim_handle h, uint32_t n, const float* in_xyzw) {
    if (!h || !in_xyzw || n >= h->neigh.size()) return fail(h, TETSIM_EINVAL, "bad neighbour slot");
    HIPCHK(h, hipSetDevice(h->opt.device));
    NeighDev& nb = h->neigh[n];
    if (!nb.recv_count) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->pj.pos_pred + nb.recv_start, in_xyzw, nb.recv_count * sizeof(float4), hipMemcpyHostToDevice));
    return 0;
}

}  // extern "C"

// ---- end-of-substep positions of the ghost particles, on request -------------------------------------------------------------
// The per-substep halo carries PREDICTIONS (what the next substep's ghost tets read); a ghost particle's end-of-substep position is
// not needed by the solver and never travels.  The embedded visual mesh needs it: a visual vertex inside a tet that straddles a cut
// blends four corners of which some belong to a neighbour (SoftbodyGPU.js:429-435).  So it is fetched when asked for -- once per
// frame, not per substep: every partition sends the pos_final of its send lists into its neighbours' pos_final ghost ranges.
namespace tetsim {
int refresh_final_rccl(tetsim_body* h) {
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    for (auto& nb : h->neigh)
        if (!nb.contiguous && nb.send_count) util_launch_gather4(h->stream, h->pj.pos_final, nb.send_idx, nb.send_buf, nb.send_count);
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) return rccl_fail(h, r, "ncclGroupStart");
    for (auto& nb : h->neigh) {
        const int peer = h->loopback ? h->comm_rank : nb.rank;
        if (nb.send_count) {
            r = g_rccl.Send(nb.contiguous ? h->pj.pos_final + nb.send_first : nb.send_buf, 4ull * nb.send_count, ncclFloat, peer, h->comm, h->stream);
            if (r != ncclSuccess) return rccl_fail(h, r, "ncclSend");
        }
        if (nb.recv_count) {
            r = g_rccl.Recv(h->pj.pos_final + nb.recv_start, 4ull * nb.recv_count, ncclFloat, peer, h->comm, h->stream);
            if (r != ncclSuccess) return rccl_fail(h, r, "ncclRecv");
        }
    }
    r = g_rccl.GroupEnd();
    if (r != ncclSuccess) return rccl_fail(h, r, "ncclGroupEnd");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->final_ghosts_fresh = true;
    return 0;
}
}  // namespace tetsim

extern "C" {
// What one halo exchange costs THIS rank with ITS neighbours and ITS message sizes -- the transfer term of the substep's halo chain
// (wait V + halo-side tiles + boundary particles + transfer) -- measured, not assumed: `reps` times the grouped send /
// recv of the current predictions into the neighbours' ghost ranges (idempotent: the ghosts already hold these values), each bracketed
// by events on the halo stream.  A collective: every rank calls it with the same `reps`, between steps.  bench.py --gpus N reports it
// per rank so that the first run on real xGMI says which term of the chain the wire is.
int tetsim_halo_probe(tetsim_handle h, uint32_t reps, double* min_us, double* median_us, double* max_us) {
    if (!h || !min_us || !median_us || !max_us || reps == 0 || reps > 4096) return fail(h, TETSIM_EINVAL, "bad argument");
    if (!h->comm || !h->comm_stream) return fail(h, TETSIM_ESTATE, "no RCCL communicator on this body (tetsim_comm_init)");
    if (h->deep) return fail(h, TETSIM_ESTATE, "bodies with a two-layer ghost region exchange their ghosts through the peer-to-peer halo only");
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    struct Events : std::vector<hipEvent_t> { using std::vector<hipEvent_t>::vector; ~Events() { for (hipEvent_t e : *this) if (e) (void)hipEventDestroy(e); } } ev(2ull * reps, nullptr);
    for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
    for (uint32_t i = 0; i < reps; i++) {
        HIPCHK(h, hipEventRecord(ev[2 * i], h->comm_stream));
        for (auto& nb : h->neigh)
            if (!nb.contiguous && nb.send_count) util_launch_gather4(h->comm_stream, h->pj.pos_pred, nb.send_idx, nb.send_buf, nb.send_count);
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) return rccl_fail(h, r, "ncclGroupStart");
        for (auto& nb : h->neigh) {
            const int peer = h->loopback ? h->comm_rank : nb.rank;
            if (nb.send_count && (r = g_rccl.Send(nb.contiguous ? h->pj.pos_pred + nb.send_first : nb.send_buf, 4ull * nb.send_count, ncclFloat, peer, h->comm, h->comm_stream)) != ncclSuccess)
                return rccl_fail(h, r, "ncclSend");
            if (nb.recv_count && (r = g_rccl.Recv(ghost_buffer(h, static_cast<uint32_t>(h->p2p_round)) + (nb.recv_start - h->pj.nv_owned), 4ull * nb.recv_count, ncclFloat, peer, h->comm, h->comm_stream)) != ncclSuccess)
                return rccl_fail(h, r, "ncclRecv");
        }
        if ((r = g_rccl.GroupEnd()) != ncclSuccess) return rccl_fail(h, r, "ncclGroupEnd");
        HIPCHK(h, hipEventRecord(ev[2 * i + 1], h->comm_stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    std::vector<double> us(reps);
    for (uint32_t i = 0; i < reps; i++) { float ms = 0.0f; HIPCHK(h, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); us[i] = 1e3 * ms; }
    std::sort(us.begin(), us.end());
    *min_us = us.front(); *median_us = us[reps / 2]; *max_us = us.back();
    return 0;
}
int tetsim_halo_refresh_final(tetsim_handle h) {
    if (!h) return TETSIM_EINVAL;
    if (!h->partitioned || h->neigh.empty()) { h->final_ghosts_fresh = true; return 0; }
    if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return fail(h, TETSIM_ESTATE, "POLAR_JACOBI only");
    if (!h->comm) return fail(h, TETSIM_ESTATE, "no RCCL communicator on this body (in-process groups: tetsim_group_refresh_final)");
    return refresh_final_rccl(h);
}
static int tetsim_group_refresh_final_impl(tetsim_handle* hs, uint32_t count);
int tetsim_group_refresh_final(tetsim_handle* hs, uint32_t count) {
    group_begin(hs, count);
    return group_result(hs, count, tetsim_group_refresh_final_impl(hs, count));
}
static int tetsim_group_refresh_final_impl(tetsim_handle* hs, uint32_t count) {
    if (!hs || count == 0) return TETSIM_EINVAL;
    for (uint32_t i = 0; i < count; i++)
        if (!hs[i] || hs[i]->opt.part_count != static_cast<int32_t>(count) || hs[i]->opt.part_index != static_cast<int32_t>(i) || hs[i]->opt.solver != TETSIM_SOLVER_POLAR_JACOBI)
            return fail(hs[i], TETSIM_EINVAL, "handles[i] must be partition i of a count-way POLAR_JACOBI decomposition");
    for (uint32_t i = 0; i < count; i++) {   // every partition has finished its last substep before anyone's ghost range is written
        HIPCHK(hs[i], hipSetDevice(hs[i]->opt.device));
        HIPCHK(hs[i], hipStreamSynchronize(hs[i]->stream));
        if (hs[i]->comm_stream) HIPCHK(hs[i], hipStreamSynchronize(hs[i]->comm_stream));
    }
    for (uint32_t i = 0; i < count; i++) {
        tetsim_body* src = hs[i];
        HIPCHK(src, hipSetDevice(src->opt.device));
        for (auto& nb : src->neigh) {
            if (!nb.send_count) continue;
            tetsim_body* dst = hs[nb.rank];
            NeighDev* back = nullptr;
            for (auto& r : dst->neigh) if (r.rank == static_cast<int>(i)) back = &r;
            if (!back || back->recv_count != nb.send_count) return fail(src, TETSIM_ESTATE, "asymmetric halo plan");
            if (!nb.contiguous) util_launch_gather4(src->stream, src->pj.pos_final, nb.send_idx, nb.send_buf, nb.send_count);
            HIPCHK(src, hipMemcpyAsync(dst->pj.pos_final + back->recv_start, nb.contiguous ? src->pj.pos_final + nb.send_first : nb.send_buf,
                                       nb.send_count * sizeof(float4), hipMemcpyDeviceToDevice, src->stream));
        }
    }
    for (uint32_t i = 0; i < count; i++) { HIPCHK(hs[i], hipSetDevice(hs[i]->opt.device)); HIPCHK(hs[i], hipStreamSynchronize(hs[i]->stream)); hs[i]->final_ghosts_fresh = true; }
    return 0;
}
}  // extern "C"
