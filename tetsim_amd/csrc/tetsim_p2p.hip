// tetsim_p2p.hip -- C ABI, peer-to-peer halo (include/tetsim.h: tetsim_halo_p2p_export / _connect): a rank describes its ghost
// buffers and "arrived" words in a blob, the caller gathers the blobs, every rank maps its neighbours' (HIP IPC, or plain pointers for
// ranks of one process).  What the boundary-particle kernel does with the mappings is in tetsim_halo.hip / pj_blocked.hip.
#include "body.h"

#include <unistd.h>

using namespace tetsim;

extern "C" {

// ---- peer-to-peer halo ---------------------------------------------------------------------------------------------------
namespace {
struct P2PBlob {   // what a rank tells the others about its buffers (TETSIM_P2P_BLOB_BYTES)
    uint32_t magic, rank, part_count, device;
    uint64_t pid;
    uint32_t nv_owned, nv_local, n_neigh, n_ghost1;             // n_ghost1: 0xffffffff = one ghost layer
    hipIpcMemHandle_t h_pred, h_alt, h_arrived;                 // IPC handles of pos_pred / ghost_alt / the "arrived" words
    uint64_t p_pred, p_alt, p_arrived;                          // ... and the plain pointers (ranks of the same process)
    struct { int32_t rank; uint32_t recv_start, recv_count, recv2_start, recv2_count; } neigh[kMaxPeers];
};
constexpr uint32_t kArrivedWords = 4 * kMaxPeers;               // [set][parity][neighbour]; one-layer bodies use the first 2 * kMaxPeers as [parity][neighbour]
static_assert(sizeof(P2PBlob) <= TETSIM_P2P_BLOB_BYTES, "blob too large");
constexpr uint32_t kP2PMagic = 0x50325054u;
}  // namespace

int tetsim_halo_p2p_export(tetsim_handle h, void* blob) {
    if (!h || !blob) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->partitioned || !h->blocked || h->blk.nb == h->blk.nb_interior)
        return fail(h, TETSIM_ESTATE, "the peer-to-peer halo needs a partitioned POLAR_JACOBI body in the blocked FAST formulation with halo-side tiles");
    if (h->neigh.size() > kMaxPeers) return fail(h, TETSIM_ESTATE, "the peer-to-peer halo supports at most 8 neighbours per partition");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const uint32_t nvo = h->pj.nv_owned, ng = h->pj.nv_local - nvo;
    const uint32_t ng1 = h->deep ? h->n_ghost1 : ng, ng2 = ng - ng1;
    if (!h->ghost_alt) {
        int rc;
        // one ghost layer: the second (odd-substep) ghost buffer.  Two layers: the eight receive buffers in one allocation,
        // [g1_even x2 | g1_final x2 | g2_even x2 | g2_odd x2]
        if ((rc = dev_alloc(h, &h->ghost_alt, h->deep ? 4ull * ng1 + 4ull * ng2 : ng))) return rc;
        if (h->deep)
            for (uint32_t st = 0; st < 2; st++) {
                h->own_g1_even[st] = h->ghost_alt + static_cast<size_t>(st) * ng1;
                h->own_g1_final[st] = h->ghost_alt + (2ull + st) * ng1;
                h->own_g2_even[st] = h->ghost_alt + 4ull * ng1 + static_cast<size_t>(st) * ng2;
                h->own_g2_odd[st] = h->ghost_alt + 4ull * ng1 + (2ull + st) * ng2;
            }
        if ((rc = dev_alloc(h, &h->d_arrived, kArrivedWords))) return rc;
        HIPCHK(h, hipMemset(h->d_arrived, 0, kArrivedWords * sizeof(uint32_t)));
        // where each boundary particle goes: (neighbour, position in that neighbour's ghost run for this rank), ELL by particle
        const uint32_t nvb = h->pj.nv_boundary;
        std::vector<std::vector<uint32_t>> per(nvb);
        for (size_t k = 0; k < h->neigh.size(); k++)
            for (size_t j = 0; j < h->neigh[k].send_local.size(); j++) {
                const uint32_t api = static_cast<uint32_t>(h->neigh[k].send_local[j]);
                const uint32_t dv = h->api2dev.empty() ? api : h->api2dev[api];
                if (dv >= nvb) return fail(h, TETSIM_ESTATE, "internal error: a sent particle is not a boundary particle");
                per[dv].push_back((static_cast<uint32_t>(k) << 24) | static_cast<uint32_t>(j));
            }
        uint32_t cols = 1;
        for (auto& v : per) cols = std::max<uint32_t>(cols, static_cast<uint32_t>(v.size()));
        h->p2p_cols = cols; h->p2p_stride = std::max(nvb, 1u);
        std::vector<uint32_t> ell(static_cast<size_t>(cols) * h->p2p_stride, 0xffffffffu);
        for (uint32_t v = 0; v < nvb; v++) for (size_t c = 0; c < per[v].size(); c++) ell[c * h->p2p_stride + v] = per[v][c];
        if ((rc = dev_alloc(h, &h->d_peer_slots, ell.size()))) return rc;
        if ((rc = upload(h, h->d_peer_slots, ell))) return rc;
        if (h->deep) {   // the same table for the neighbours' SECOND layer
            std::vector<std::vector<uint32_t>> per2(nvb);
            for (size_t k = 0; k < h->part.neigh.size(); k++)
                for (size_t j = 0; j < h->part.neigh[k].send2_local.size(); j++) {
                    const uint32_t api = static_cast<uint32_t>(h->part.neigh[k].send2_local[j]);
                    const uint32_t dv = h->api2dev.empty() ? api : h->api2dev[api];
                    if (dv >= nvb) return fail(h, TETSIM_ESTATE, "internal error: a sent particle is not a boundary particle");
                    per2[dv].push_back((static_cast<uint32_t>(k) << 24) | static_cast<uint32_t>(j));
                }
            uint32_t cols2 = 1;
            for (auto& v : per2) cols2 = std::max<uint32_t>(cols2, static_cast<uint32_t>(v.size()));
            h->p2p_cols2 = cols2;
            std::vector<uint32_t> ell2(static_cast<size_t>(cols2) * h->p2p_stride, 0xffffffffu);
            for (uint32_t v = 0; v < nvb; v++) for (size_t c = 0; c < per2[v].size(); c++) ell2[c * h->p2p_stride + v] = per2[v][c];
            if ((rc = dev_alloc(h, &h->d_peer_slots2, ell2.size()))) return rc;
            if ((rc = upload(h, h->d_peer_slots2, ell2))) return rc;
        }
    }
    P2PBlob b;
    std::memset(&b, 0, sizeof b);
    b.magic = kP2PMagic; b.rank = static_cast<uint32_t>(h->opt.part_index); b.part_count = static_cast<uint32_t>(h->opt.part_count);
    b.device = static_cast<uint32_t>(h->opt.device); b.pid = static_cast<uint64_t>(getpid());
    b.nv_owned = nvo; b.nv_local = h->pj.nv_local; b.n_neigh = static_cast<uint32_t>(h->neigh.size());
    b.n_ghost1 = h->deep ? ng1 : 0xffffffffu;
    b.p_pred = reinterpret_cast<uint64_t>(h->pj.pos_pred); b.p_alt = reinterpret_cast<uint64_t>(h->ghost_alt); b.p_arrived = reinterpret_cast<uint64_t>(h->d_arrived);
    // (a handle can only be opened by ANOTHER process; failing to make one is not an error for ranks of one process)
    (void)hipIpcGetMemHandle(&b.h_pred, h->pj.pos_pred);
    (void)hipIpcGetMemHandle(&b.h_alt, h->ghost_alt);
    (void)hipIpcGetMemHandle(&b.h_arrived, h->d_arrived);
    (void)hipGetLastError();
    for (size_t k = 0; k < h->neigh.size(); k++) {
        b.neigh[k].rank = h->neigh[k].rank; b.neigh[k].recv_start = h->neigh[k].recv_start; b.neigh[k].recv_count = h->neigh[k].recv_count;
        if (h->deep) { b.neigh[k].recv2_start = h->part.neigh[k].recv2_start; b.neigh[k].recv2_count = h->part.neigh[k].recv2_count; }
    }
    std::memset(blob, 0, TETSIM_P2P_BLOB_BYTES);
    std::memcpy(blob, &b, sizeof b);
    return 0;
}

int tetsim_halo_p2p_connect(tetsim_handle h, const void* blobs, uint32_t count) {
    if (!h || !blobs) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->ghost_alt) return fail(h, TETSIM_ESTATE, "call tetsim_halo_p2p_export first");
    // Transports it can follow: RCCL (connect after tetsim_comm_init; RCCL keeps carrying the refresh after a dt change), an
    // in-process group (after its first tetsim_group_step_n), or none at all -- ranks in different processes that exchange the
    // blobs themselves; such a body cannot change dt between calls (nothing would refresh the ghosts).
    {
        const char* sy = getenv("TETSIM_HALO_SYNC");
        const char* os = getenv("TETSIM_DEBUG_ONE_STREAM");
        if ((sy && sy[0] == 'e') || (os && os[0] == '1')) return fail(h, TETSIM_ESTATE, "the peer-to-peer halo rides on the two-queue flag path (TETSIM_HALO_SYNC=events / TETSIM_DEBUG_ONE_STREAM exclude it)");
    }
    if (!h->comm_stream) { int rc = create_halo_stream(h); if (rc) return rc; }
    h->timeout_ms = 0;
    h->timeout_ms = halo_timeout_ms(h);
    if (!(h->loopback && count == 1) && count != static_cast<uint32_t>(h->opt.part_count)) return fail(h, TETSIM_EINVAL, "one blob per partition, in rank order");
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    const uint32_t nvo = h->pj.nv_owned, ng = h->pj.nv_local - nvo;
    const uint32_t ng1 = h->deep ? h->n_ghost1 : ng, ng2 = ng - ng1;
    if (h->deep) {
        // the first substep after the connection is an EVEN one of exchange set 0: its ghosts are the ghosts as they are now
        if (ng1) HIPCHK(h, hipMemcpy(h->own_g1_even[0], h->pj.pos_pred + nvo, ng1 * sizeof(float4), hipMemcpyDeviceToDevice));
        if (ng1) HIPCHK(h, hipMemcpy(h->own_g1_final[0], h->pj.pos_final + nvo, ng1 * sizeof(float4), hipMemcpyDeviceToDevice));
        if (ng2) HIPCHK(h, hipMemcpy(h->own_g2_even[0], h->pj.pos_pred + nvo + ng1, ng2 * sizeof(float4), hipMemcpyDeviceToDevice));
    } else if (ng) {
        // both ghost buffers start from the ghosts as they are now (the last exchange of the previous transport, or the rest pose)
        HIPCHK(h, hipMemcpy(h->ghost_alt, h->pj.pos_pred + nvo, ng * sizeof(float4), hipMemcpyDeviceToDevice));
    }
    HIPCHK(h, hipMemset(h->d_arrived, 0, kArrivedWords * sizeof(uint32_t)));
    const char* all = static_cast<const char*>(blobs);
    // a re-connect replaces the links: the IPC mappings of the previous connection are closed first (opening a handle that is still
    // open can fail), and every error return below closes what this call has opened so far
    for (PeerLink& l : h->links) for (void*& m : l.ipc) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
    h->links.clear();
    h->p2p = false;
    drop_flag_graphs(h);   // the captured chains hold kernel nodes with the closed peer pointers baked in: gone with the mappings, also on an error return below
    std::vector<PeerLink> links(h->neigh.size());
    struct CloseOnError {
        std::vector<PeerLink>& v; bool armed = true;
        ~CloseOnError() { if (armed) for (PeerLink& l : v) for (void*& m : l.ipc) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; } }
    } guard{links};
    for (size_t k = 0; k < h->neigh.size(); k++) {
        const NeighDev& nb = h->neigh[k];
        P2PBlob pb;
        std::memcpy(&pb, all + static_cast<size_t>(h->loopback ? 0 : nb.rank) * TETSIM_P2P_BLOB_BYTES, sizeof pb);
        if (pb.magic != kP2PMagic || (!h->loopback && static_cast<int>(pb.rank) != nb.rank)) return fail(h, TETSIM_EINVAL, "bad peer blob for rank " + std::to_string(nb.rank));
        const uint32_t send2 = h->deep ? static_cast<uint32_t>(h->part.neigh[k].send2_local.size()) : 0u;
        if (!nb.send_count && !send2) continue;
        if ((pb.n_ghost1 != 0xffffffffu) != h->deep) return fail(h, TETSIM_ESTATE, "ranks " + std::to_string(h->opt.part_index) + " and " + std::to_string(nb.rank) + " disagree on the depth of the ghost region");
        // this rank's run in the neighbour's ghost range, and this rank's slot among the neighbour's neighbours
        uint32_t start = 0, cnt = 0, start2 = 0, cnt2 = 0, slot = kMaxPeers;
        if (h->loopback) {
            start = nb.recv_start; cnt = nb.recv_count; slot = static_cast<uint32_t>(k);
            if (h->deep) { start2 = h->part.neigh[k].recv2_start; cnt2 = h->part.neigh[k].recv2_count; }
        } else
            for (uint32_t j = 0; j < pb.n_neigh && j < kMaxPeers; j++)
                if (pb.neigh[j].rank == h->opt.part_index) { start = pb.neigh[j].recv_start; cnt = pb.neigh[j].recv_count; start2 = pb.neigh[j].recv2_start; cnt2 = pb.neigh[j].recv2_count; slot = j; }
        if (slot == kMaxPeers || cnt != nb.send_count || cnt2 != send2) return fail(h, TETSIM_ESTATE, "asymmetric halo plan between ranks " + std::to_string(h->opt.part_index) + " and " + std::to_string(nb.rank));
        // The blob is another process's word: before the boundary-particle kernel stores into peer memory at start + i, the runs must lie
        // inside the peer's ghost range as the peer itself describes it (a blob of another mesh or partition map with equal counts would
        // otherwise write out of bounds), and a send-list position must fit the 24 bits PJPeer's slot words give it.
        if (!h->loopback) {
            const uint64_t pg = static_cast<uint64_t>(pb.nv_local) - pb.nv_owned, pg1 = h->deep ? pb.n_ghost1 : pg;
            const bool ok1 = pb.nv_local >= pb.nv_owned && pg1 <= pg && (cnt == 0u || (start >= pb.nv_owned && static_cast<uint64_t>(start) + cnt <= pb.nv_owned + pg1));
            const bool ok2 = cnt2 == 0u || (start2 >= pb.nv_owned + pg1 && static_cast<uint64_t>(start2) + cnt2 <= pb.nv_local);
            if (!ok1 || !ok2) return fail(h, TETSIM_EINVAL, "bad peer blob for rank " + std::to_string(nb.rank) + ": this rank's run lies outside the peer's ghost range");
        }
        if (cnt >= (1u << 24) || cnt2 >= (1u << 24)) return fail(h, TETSIM_EINVAL, "a neighbour's send list has 2^24 particles or more: the peer-to-peer slot words cannot address it");
        float4 *pred = nullptr, *alt = nullptr;
        uint32_t* arr = nullptr;
        if (pb.pid == static_cast<uint64_t>(getpid())) {   // same process: plain pointers (peer access if the devices differ)
            pred = reinterpret_cast<float4*>(pb.p_pred); alt = reinterpret_cast<float4*>(pb.p_alt); arr = reinterpret_cast<uint32_t*>(pb.p_arrived);
            if (static_cast<int>(pb.device) != h->opt.device) {
                const hipError_t e = hipDeviceEnablePeerAccess(static_cast<int>(pb.device), 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(h, TETSIM_EHIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                (void)hipGetLastError();
            }
        } else {
            PeerLink& l = links[k];
            HIPCHK(h, hipIpcOpenMemHandle(&l.ipc[0], pb.h_pred, hipIpcMemLazyEnablePeerAccess));
            HIPCHK(h, hipIpcOpenMemHandle(&l.ipc[1], pb.h_alt, hipIpcMemLazyEnablePeerAccess));
            HIPCHK(h, hipIpcOpenMemHandle(&l.ipc[2], pb.h_arrived, hipIpcMemLazyEnablePeerAccess));
            pred = static_cast<float4*>(l.ipc[0]); alt = static_cast<float4*>(l.ipc[1]); arr = static_cast<uint32_t*>(l.ipc[2]);
        }
        if (h->deep) {
            const size_t pg1 = pb.n_ghost1, pg2 = pb.nv_local - pb.nv_owned - pb.n_ghost1;
            const size_t r1 = start - pb.nv_owned, r2 = start2 - pb.nv_owned - pb.n_ghost1;   // this rank's runs in the neighbour's layers
            for (uint32_t st = 0; st < 2; st++) {
                links[k].g1_even[st] = alt + st * pg1 + r1;
                links[k].g1_final[st] = alt + (2 + st) * pg1 + r1;
                links[k].g2_even[st] = alt + 4 * pg1 + st * pg2 + r2;
                links[k].g2_odd[st] = alt + 4 * pg1 + (2 + st) * pg2 + r2;
                for (uint32_t par = 0; par < 2; par++) links[k].arrived2[st][par] = arr + (st * 2 + par) * kMaxPeers + slot;
            }
            continue;
        }
        links[k].ghost[0] = pred + start;                       // even substeps read pos_pred's tail
        links[k].ghost[1] = alt + (start - pb.nv_owned);        // odd ones the second buffer
        links[k].arrived[0] = arr + slot;
        links[k].arrived[1] = arr + kMaxPeers + slot;
    }
    guard.armed = false;
    h->links = std::move(links);
    h->p2p = true;
    h->p2p_round = 0;
    h->p2p_probe_base = 0;
    h->p2p_raise_pending = false;
    h->fold_wait = h->fold_possible;   // (between calls: nothing is pending, the captured chains are dropped below)
    {   // ... and the halo queue's wait kernel goes into the halo-side tiles (tetsim_halo.hip: enqueue_phase_a), one-layer ghost regions only
        const char* fw = getenv("TETSIM_HALO_FOLD_WAIT");
        h->fold_halo = !(fw && fw[0] == '0') && !h->deep;
    }
    h->halo_warm = false;     // the first call after the connection runs eagerly (its first substep has no "arrived" to wait for)
    drop_flag_graphs(h);
    return 0;
}

// The peer-to-peer halo's transfer term on THIS wire, measured like tetsim_halo_probe measures RCCL's: `reps` hand-overs with all
// neighbours at once (a store into each neighbour's inbox word, a wait on the own ones -- the words behind the "arrived" words of a
// one-layer body, unused otherwise), one wave on the halo stream, stamped by the device's 100 MHz clock.  A repetition costs one one-way
// signal latency; the payload (the boundary predictions, tetsim_comm_info's bytes) rides with the same stores in a real substep.
// A collective: every rank calls it with the same reps, between steps, the same number of times.
int tetsim_halo_p2p_probe(tetsim_handle h, uint32_t reps, double* min_us, double* median_us, double* max_us) {
    if (!h || !min_us || !median_us || !max_us || reps == 0 || reps > 4096) return fail(h, TETSIM_EINVAL, "bad argument");
    if (!h->p2p || !h->d_arrived || !h->comm_stream) return fail(h, TETSIM_ESTATE, "no peer-to-peer halo on this body (tetsim_halo_p2p_connect)");
    if (h->deep) return fail(h, TETSIM_ESTATE, "bodies with a two-layer ghost region use all their hand-over words");
    HIPCHK(h, hipSetDevice(h->opt.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    P2PProbe p;
    for (size_t k = 0; k < h->neigh.size() && k < kMaxPeers; k++) {
        if (!h->links[k].arrived[0] || !h->neigh[k].recv_count) continue;
        p.raise[p.n] = h->links[k].arrived[0] + 2u * kMaxPeers;     // (arrived[0] = the neighbour's word array + this rank's slot there)
        p.wait[p.n] = h->d_arrived + 2u * kMaxPeers + k;
        p.n++;
    }
    *min_us = *median_us = *max_us = 0.0;
    if (p.n == 0) return 0;
    unsigned long long* d_ticks = nullptr;
    uint32_t* d_err = nullptr;
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&d_ticks), reps * sizeof(unsigned long long)));
    if (hipMalloc(reinterpret_cast<void**>(&d_err), sizeof(uint32_t)) != hipSuccess) { (void)hipFree(d_ticks); return fail(h, TETSIM_ENOMEM, "hipMalloc"); }
    (void)hipMemset(d_err, 0, sizeof(uint32_t));
    util_launch_p2p_probe(h->comm_stream, p, h->p2p_probe_base, reps, d_ticks, d_err, halo_timeout_ms(h));
    h->p2p_probe_base += reps;
    std::vector<unsigned long long> ticks(reps);
    uint32_t err = 0;
    const bool ok = hipStreamSynchronize(h->comm_stream) == hipSuccess && hipMemcpy(ticks.data(), d_ticks, reps * sizeof(ticks[0]), hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(&err, d_err, sizeof err, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d_ticks); (void)hipFree(d_err);
    if (!ok) return fail(h, TETSIM_EHIP, "tetsim_halo_p2p_probe: the probe kernel failed");
    if (err) return fail(h, TETSIM_ECOMM, "tetsim_halo_p2p_probe: a neighbour's hand-over did not arrive in time (every rank calls the probe together, with the same reps)");
    std::sort(ticks.begin(), ticks.end());
    *min_us = ticks.front() / 100.0; *median_us = ticks[reps / 2] / 100.0; *max_us = ticks.back() / 100.0;
    return 0;
}

}  // extern "C"
