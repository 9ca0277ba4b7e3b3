// tetsim_halo.hip -- multi-GPU: the per-substep halo choreography of partitioned POLAR_JACOBI bodies and the stepping of
// in-process groups.  Communicator set-up and probes: tetsim_comm.hip; peer-to-peer halo (export / connect): tetsim_p2p.hip.
#include "body.h"

#include <unistd.h>

using namespace tetsim;

namespace tetsim {

Rccl g_rccl;
std::atomic<uint64_t> g_stream_generation{0};

int create_halo_stream(tetsim_body* h) {
    if (h->comm_stream) return 0;
    int lo = 0, hi = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(h, hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi));
    g_stream_generation++;
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_bnd_tet, hipEventDisableTiming));
    // the halo stream's own copy of the substep parameters (push_params fills it in THIS stream's order)
    return dev_alloc(h, &h->d_params_halo, 1);
}

// Ghost particle nv_owned + i of a substep with that parity: pos_pred's own tail, or -- peer-to-peer bodies, odd substeps -- the
// second ghost buffer.
float4* ghost_buffer(tetsim_body* h, uint32_t parity) {
    return (h->p2p && (parity & 1u)) ? h->ghost_alt : h->pj.pos_pred + h->pj.nv_owned;
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
    if (h->deep) return fail(h, TETSIM_ESTATE, "bodies with a two-layer ghost region exchange their ghosts through the peer-to-peer halo only");
    const uint32_t p = h->halo_parity;
    if (h->flag_sync) {  // the halo stream ran this substep's boundary particles itself: stay in stream order
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
            if (nb.send_count) util_launch_copy(h->comm_stream, nb.contiguous ? h->pj.pos_pred + nb.send_first : nb.send_buf, ghost_buffer(h, static_cast<uint32_t>(h->p2p_round)) + (nb.recv_start - h->pj.nv_owned), nb.send_count);
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
                r = g_rccl.Recv(ghost_buffer(h, static_cast<uint32_t>(h->p2p_round)) + (nb.recv_start - h->pj.nv_owned), 4ull * nb.recv_count, ncclFloat, h->loopback ? h->comm_rank : nb.rank, h->comm, h->comm_stream);
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
            { HP("memcpyAsync d2d"); HIPCHK(h, hipMemcpyAsync(ghost_buffer(dst, static_cast<uint32_t>(dst->p2p_round)) + (back->recv_start - dst->pj.nv_owned), from, nb.send_count * sizeof(float4), hipMemcpyDeviceToDevice, h->comm_stream)); }
        }
    }
    if (h->loopback) {   // measurement only: how much wire latency the choreography hides (tools/loopback_rank.py)
        static const uint32_t delay_us = [] { const char* e = getenv("TETSIM_DEBUG_LOOPBACK_DELAY_US"); return e ? static_cast<uint32_t>(strtoul(e, nullptr, 10)) : 0u; }();
        util_launch_delay(h->comm_stream, delay_us);
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
    if (!h->comm && !h->group.empty())
        for (auto& nb : h->neigh)
            if (nb.recv_count) { HP("wait peer sent"); HIPCHK(h, hipStreamWaitEvent(on, h->group[nb.rank]->ev_sent2[p], 0)); }
    h->halo_pending = false;
    return 0;
}
// Bound of the device-side waits of the flag path.  `wait G` sits behind a transfer, i.e. behind the PEER's progress: a rank
// that steps this much later than its neighbour is reported as TETSIM_ECOMM at the next synchronisation.  0 = wait for ever.
uint32_t halo_timeout_ms(const tetsim_body* h) {
    if (h && h->timeout_ms) return h->timeout_ms;
    const char* e = getenv("TETSIM_HALO_TIMEOUT_MS");   // (read when a body is created / connected, not per call)
    return e ? static_cast<uint32_t>(strtoul(e, nullptr, 10)) : 30000u;
}
// (a connected peer-to-peer halo is a transport of its own: ranks in different processes need no RCCL communicator for it)
bool has_transport(const tetsim_body* h) { return !h->neigh.empty() && (h->comm || !h->group.empty() || h->p2p); }
// blocked bodies with a transport and halo-side tiles step through the flag-synchronised two-queue path (enqueue_phase_a)
bool uses_flag_sync(const tetsim_body* h) {
    static const bool one_stream = [] { const char* e = getenv("TETSIM_DEBUG_ONE_STREAM"); return e && e[0] == '1'; }();
    return h->halo_use_flags && !one_stream && has_transport(h) && h->blocked && h->blk.nb > h->blk.nb_interior;
}

// Host cost matters here: a substep is ~42 us of GPU work and every launch / event call costs 1.5-4 us, so the eager
// halo path issues as few operations as possible -- 3 kernel launches (interior tiles, boundary tiles, ONE particle pass),
// 1 event record + 1 cross-stream wait to start the transfer, 1 record after it, 1 wait before the next boundary tiles.
// The transfer overlaps the NEXT substep's interior tet kernel (~30 us), which is ample for a 200 KB message.
//
// In-process groups must issue every partition's particle pass before anyone's sends (a send waits for the RECEIVER's
// boundary event of the same substep), so a substep is enqueued in two phases; RCCL bodies run both back to back.
// The main queue's second half of a substep: wait for G ("the halo-side tiles of this substep are done": the interior particles add
// up their partial sums too), then the interior particles.  Folded (h->fold_wait): the particle kernel's waves look at G themselves and
// the queue's NEXT kernel puts the word back as it starts (the interior tiles of the next substep, or flush_v's signal) -- two kernels per
// substep on this queue, like a monolithic body, instead of three.
static void interior_particles(tetsim_body* h, const PJSync& yg, hipEvent_t* ev) {
    const uint32_t nvb = h->pj.nv_boundary, cnt = h->pj.nv_owned - nvb;
    // Waves that look at a word hold their slots while they wait, and the kernel that raises the word needs slots too: only one rank
    // per process (partitions of one process share the device: eight 1 M-tet slabs' particle kernels are 22,000 waves on 8,192 slots
    // -- they starved the boundary-particle kernels until the time-out), and only while the kernel is at most half the device's waves.
    if (h->fold_wait && h->group.empty() && (cnt + 63u) / 64u <= h->fold_wave_limit) {
        HP("launch vertex interior (awaits G)");
        pjb_launch_vertex_await(h->stream, h->blk, nvb, cnt, yg, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr);
        return;
    }
    { HP("wait G"); pjb_launch_wait(h->stream, yg); }
    { HP("launch vertex interior"); pj_vertex(h, nvb, cnt, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr); }
}

int enqueue_phase_a(tetsim_body* h, hipEvent_t* ev) {  // tet kernels + particles; ev[0..3]: begin/end of the interior tet and the particle kernel
    if (h->deep && !h->p2p) return fail(h, TETSIM_ESTATE, "a body with a two-layer ghost region steps through the peer-to-peer halo only: call tetsim_halo_p2p_export / _connect first");
    if (!h->group.empty()) HIPCHK(h, hipSetDevice(h->opt.device));  // in-process groups may span devices: streams, events and lazy allocations below are per device
    // (a substep that is not its call's last: every particle kernel below gets a null velocity array -- tetsim_api.hip: enqueue_substep)
    struct VelGuard { PJBlk& b; float4* keep; ~VelGuard() { b.vel = keep; } } vel_guard{h->blk, h->blk.vel};
    if (h->vel_dead) h->blk.vel = nullptr;
    if (h->blocked) {
        const uint32_t nbnd = h->blk.nb - h->blk.nb_interior;
        static const bool one_stream = [] { const char* e = getenv("TETSIM_DEBUG_ONE_STREAM"); return e && e[0] == '1'; }();
        if (nbnd && !one_stream && h->halo_use_flags) {   // (TETSIM_HALO_SYNC=events at creation: the older path below)
            // Two queues, synchronised through device words instead of events (a cross-stream event costs ~15 us eagerly and ~6 us
            // as a graph edge here, and a substep has two hand-overs on its critical path):
            //   main stream:  interior tiles(s) [raises V(s-1)] -> wait G(s) -> interior particles(s)
            //   halo stream:  wait V(s-1) -> H tiles(s) -> boundary particles(s) [raises G(s)] -> transfer(s)     [enqueue_phase_b]
            // H tiles = the tiles that touch a ghost OR a boundary particle (host_prep.cpp): every contribution to a boundary
            // particle comes from an H tile, so the halo queue finishes the boundary particles itself and starts the transfer
            // while the main queue is still in its tet kernel.  The H tiles wait for V: they also read interior particles.
            // A wait is a one-wave kernel on a binary semaphore (pj_blocked.hip); a signal is ONE STORE at the start of the kernel
            // that follows the producer in its queue (an in-order queue starts a kernel when its predecessor is complete):
            // "H tiles done" is raised by the boundary-particle kernel, "interior particles done" by the next substep's interior tet
            // kernel -- or by a signal kernel where no such kernel follows (end of a call: flush_v).  A kernel of its own costs
            // ~2.7 us of queue time here, and the substep had a signal kernel on each of its two critical chains.
            // Host submission order follows the dependencies, every wait behind the kernel that raises its word, so the eager path
            // stays live even if the runtime maps both streams onto one hardware queue (it then merely serialises).
            if (!h->d_sync) {
                int rc = dev_alloc(h, &h->d_sync, kSyncWords);
                if (rc) return rc;
                HIPCHK(h, hipMemset(h->d_sync, 0, kSyncWords * sizeof(uint32_t)));
                HIPCHK(h, hipDeviceSynchronize());   // once: the halo stream must also see everything create() uploaded
            }
            h->flag_sync = true;
            PJSync yg, yv;   // word 0: "the H tiles of this substep are done"; word 2: "the interior particles of this substep are done"
            yg.flag = h->d_sync + 0; yg.error = h->d_sync + 4; yg.timeout_ms = halo_timeout_ms(h);
            yv.flag = h->d_sync + 2; yv.error = h->d_sync + 4; yv.timeout_ms = yg.timeout_ms;
            const uint32_t nvb = h->pj.nv_boundary;
            int rc;
            const bool v_open = h->v_pending;   // the previous substep's "interior particles done": raised here, consumed below
            if (v_open && h->blk.nb_interior) h->v_pending = false;
            else if ((rc = flush_v(h))) return rc;
            // (fold_wait: whoever raises V also puts G back -- the waves that looked at it belong to the kernel in front of this one)
            const bool g_folded = h->fold_wait && h->group.empty() && (h->pj.nv_owned - nvb + 63u) / 64u <= h->fold_wave_limit;   // (interior_particles' rule)
            { HP("launch tet interior"); pjb_launch_tet(h->stream, h->blk, 0, h->blk.nb_interior, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr,
                                                       v_open && h->blk.nb_interior ? yv.flag : nullptr,
                                                       v_open && h->blk.nb_interior && g_folded ? yg.flag : nullptr); }
            PJBlk kb = h->blk;   // kernels of the halo queue read the halo queue's copy of the parameters
            if (h->d_params_halo) kb.params = h->d_params_halo;
            if (h->p2p && h->deep) {
                // Two-layer ghost region: ghosts cross only every other substep (the algorithm is
                // tests/test_partition_gloo.py's).  r = substeps since the connection, set = (r / 2) & 1 the exchange's buffer set.
                //   EVEN r:  wait [V, arrived(set, even)] - tiles: halo-side + second-layer ghost tets, ghosts from the set's EVEN buffers -
                //            boundary particles [their second-layer share -> the neighbours' ODD buffer of this set: the early message] -
                //            first ghost layer, advanced here (previous positions: the set's g1_final)
                //   ODD r:   wait [V; raises the early message's word] - halo-side tiles (first-layer ghosts: the local predictions) -
                //            boundary particles [both layers -> the neighbours' EVEN buffers of the NEXT set, + end-of-substep positions of
                //            their first layer: the one message on the critical chain] - wait [raises its word at once; the neighbours'
                //            early message] - second-layer ghost tets evolved after the fact (ghosts: local first layer, ODD buffer)
                // Nothing a neighbour stores can hit a buffer still being read: its next store into a set needs this rank's next
                // message, which this rank's queue sends behind the reads.
                const uint64_t r = h->p2p_round;
                const uint32_t par = static_cast<uint32_t>(r & 1u), set = static_cast<uint32_t>((r >> 1) & 1u);
                const uint32_t nvo = h->pj.nv_owned, ng1 = h->n_ghost1;
                const bool group = !h->group.empty();
                auto own_word = [&](uint32_t st, uint32_t pr, size_t i) { return h->d_arrived + (st * 2u + pr) * kMaxPeers + i; };
                auto raise_list = [&](PJPeerSync& w, uint32_t st, uint32_t pr) { for (const PeerLink& l : h->links) if (l.arrived2[st][pr]) w.raise[w.n_raise++] = l.arrived2[st][pr]; };
                const uint32_t delay_us = h->loopback ? [] { const char* e = getenv("TETSIM_DEBUG_LOOPBACK_DELAY_US"); return e ? static_cast<uint32_t>(strtoul(e, nullptr, 10)) : 0u; }() : 0u;
                PJPeerSync w;
                if (h->p2p_raise_pending) raise_list(w, set, 1u);   // (only an even substep leaves a raise behind: its early message)
                h->p2p_raise_pending = false;
                if (par == 0u && r > 0)
                    for (size_t i = 0; i < h->neigh.size(); i++)
                        if (h->neigh[i].recv_count || h->part.neigh[i].recv2_count) w.wait[w.n_wait++] = own_word(set, 0u, i);
                if (par == 0u) w.delay_us = delay_us;   // (loopback measurements: the message on the critical chain arrives later)
                PJSync yw = yv;
                if (!(v_open && h->blk.nb_interior)) yw.flag = nullptr;
                if (yw.flag || w.n_raise || w.n_wait) { HP("wait V + peers"); pjb_launch_wait_peers(h->comm_stream, yw, w); }
                PJPeer pr;
                pr.slots = h->d_peer_slots; pr.cols = h->p2p_cols; pr.stride = h->p2p_stride; pr.n = static_cast<uint32_t>(h->links.size());
                pr.slots2 = h->d_peer_slots2; pr.cols2 = h->p2p_cols2;
                kb.n_ghost1 = ng1;
                if (par == 0u) {
                    kb.ghost_alt = h->own_g1_even[set]; kb.ghost2 = h->own_g2_even[set];
                    { HP("launch tet halo-side + second layer"); pjb_launch_tet_alt(h->comm_stream, kb, h->blk.nb_interior, h->blk.nb - h->blk.nb_interior); }
                    for (size_t i = 0; i < h->links.size(); i++) pr.ghost2[i] = h->links[i].g2_odd[set];
                    if (!nvb) { HP("signal G"); pjb_launch_signal(h->comm_stream, yg); }
                    else { HP("launch vertex boundary"); pjb_launch_vertex_peer(h->comm_stream, kb, 0, nvb, pr, yg.flag); }
                    PJBlk kg = kb;   // the first ghost layer: previous positions from the set's g1_final (indexed by particle id)
                    kg.fin_in = h->own_g1_final[set] - nvo;
                    { HP("launch vertex first ghost layer"); pjb_launch_vertex(h->comm_stream, kg, nvo, ng1); }
                    h->p2p_raise_pending = true;
                    if (group) {   // (ranks of one process: the raise gets a kernel of its own, see the one-layer branch)
                        PJPeerSync sg;
                        raise_list(sg, set, 1u);
                        PJSync none;
                        none.error = yg.error;
                        pjb_launch_wait_peers(h->comm_stream, none, sg);
                        h->p2p_raise_pending = false;
                    }
                } else {
                    { HP("launch tet halo-side"); pjb_launch_tet(h->comm_stream, kb, h->blk.nb_interior, h->nb_first - h->blk.nb_interior); }
                    for (size_t i = 0; i < h->links.size(); i++) { pr.ghost[i] = h->links[i].g1_even[set ^ 1u]; pr.fin[i] = h->links[i].g1_final[set ^ 1u]; pr.ghost2[i] = h->links[i].g2_even[set ^ 1u]; }
                    if (!nvb) { HP("signal G"); pjb_launch_signal(h->comm_stream, yg); }
                    else { HP("launch vertex boundary"); pjb_launch_vertex_peer(h->comm_stream, kb, 0, nvb, pr, yg.flag); }
                    PJPeerSync wl;   // its first thread tells the neighbours "your even buffers of the next set are full"; then the early message
                    raise_list(wl, set ^ 1u, 0u);
                    for (size_t i = 0; i < h->neigh.size(); i++) if (h->part.neigh[i].recv2_count) wl.wait[wl.n_wait++] = own_word(set, 1u, i);
                    PJSync ye;
                    ye.error = yg.error; ye.timeout_ms = yg.timeout_ms;
                    { HP("raise + wait early message"); pjb_launch_wait_peers(h->comm_stream, ye, wl); }
                    kb.ghost_alt = h->pj.pos_pred + nvo; kb.ghost2 = h->own_g2_odd[set];
                    { HP("evolve second-layer tets"); pjb_launch_tet_alt(h->comm_stream, kb, h->nb_first, h->blk.nb - h->nb_first); }
                }
                h->p2p_round++;
                interior_particles(h, yg, ev);
                h->v_pending = true;
                return 0;
            }
            if (h->p2p) {
                // Peer-to-peer halo: no transfer.  Substep r reads its ghosts from buffer r & 1, which the neighbours' boundary-particle
                // kernels of substep r - 1 filled; this rank's boundary-particle kernel fills THEIR buffers (r + 1) & 1.  One wave in
                // front of the halo-side tiles does all the hand-overs of the queue: it raises the neighbours' "arrived" words of
                // parity r & 1 as it starts (the kernel in front of it is this rank's boundary-particle kernel of r - 1), waits for V
                // (this rank's interior particles) and for the neighbours' words here.
                const uint32_t par = static_cast<uint32_t>(h->p2p_round & 1u);
                PJPeerSync w;
                if (h->p2p_raise_pending) for (const PeerLink& l : h->links) if (l.arrived[par]) w.raise[w.n_raise++] = l.arrived[par];
                h->p2p_raise_pending = false;
                if (h->p2p_round > 0)
                    for (size_t i = 0; i < h->neigh.size(); i++) if (h->neigh[i].recv_count) w.wait[w.n_wait++] = h->d_arrived + par * kMaxPeers + i;
                if (h->loopback) { static const uint32_t delay_us = [] { const char* e = getenv("TETSIM_DEBUG_LOOPBACK_DELAY_US"); return e ? static_cast<uint32_t>(strtoul(e, nullptr, 10)) : 0u; }(); w.delay_us = delay_us; }
                PJSync yw = yv;
                if (!(v_open && h->blk.nb_interior)) yw.flag = nullptr;
                // One rank per process with boundary particles (fold_halo): no wait kernel -- the halo-side tiles raise, look at V and
                // at the neighbours' words themselves, and the boundary-particle kernel behind them puts those words back as it starts.
                const bool fold = h->fold_halo && h->group.empty() && nvb != 0u && !h->needs_halo_refresh && nbnd <= h->fold_tile_limit;   // (at most a quarter of the workgroups the device keeps resident may wait: pjb_wait_capacity)
                if (!fold && (yw.flag || w.n_raise || w.n_wait)) { HP("wait V + peers"); pjb_launch_wait_peers(h->comm_stream, yw, w); }
                if ((rc = halo_wait(h, h->comm_stream))) return rc;   // (a refresh exchange after a dt change, in-process groups)
                kb.ghost_alt = h->ghost_alt;
                PJPeer pr;
                pr.slots = h->d_peer_slots; pr.cols = h->p2p_cols; pr.stride = h->p2p_stride; pr.n = static_cast<uint32_t>(h->links.size());
                for (size_t i = 0; i < h->links.size(); i++) pr.ghost[i] = h->links[i].ghost[par ^ 1u];
                PJClear clr;
                if (fold) {
                    HP("launch tet halo-side (hand-overs inside)");
                    pjb_launch_tet_hwait(h->comm_stream, kb, h->blk.nb_interior, nbnd, yw, w, par ? h->ghost_alt : h->pj.pos_pred + h->pj.nv_owned);
                    if (yw.flag) clr.word[clr.n++] = yw.flag;
                    for (uint32_t i = 0; i < w.n_wait; i++) clr.word[clr.n++] = w.wait[i];
                } else { HP("launch tet halo-side"); if (par) pjb_launch_tet_alt(h->comm_stream, kb, h->blk.nb_interior, nbnd); else pjb_launch_tet(h->comm_stream, kb, h->blk.nb_interior, nbnd); }
                if (!nvb) { HP("signal G"); pjb_launch_signal(h->comm_stream, yg); }
                else { HP("launch vertex boundary"); pjb_launch_vertex_peer(h->comm_stream, kb, 0, nvb, pr, yg.flag, clr); h->p2p_raise_pending = true; }
                if (h->p2p_raise_pending && !h->group.empty()) {
                    // partitions of ONE process may share hardware queues: a wait of one body must never be submitted in front of the
                    // kernel of another body that raises its word.  Here the raise gets a kernel of its own right behind the
                    // boundary particles (all bodies finish a substep's submission before any starts the next: tetsim_group_step_n);
                    // one rank per process folds it into the next substep's wait kernel, whose peers live on other devices' queues.
                    PJPeerSync sg;
                    for (const PeerLink& l : h->links) if (l.arrived[par ^ 1u]) sg.raise[sg.n_raise++] = l.arrived[par ^ 1u];
                    PJSync none;
                    none.error = yg.error;
                    { HP("raise peers"); pjb_launch_wait_peers(h->comm_stream, none, sg); }
                    h->p2p_raise_pending = false;
                }
                h->p2p_round++;
                interior_particles(h, yg, ev);
                h->v_pending = true;
                return 0;
            }
            // (RCCL, one rank per process, TETSIM_HALO_FOLD_WAIT=1: the halo-side tiles look at V themselves -- as the peer-to-peer branch
            // does by default -- and the boundary-particle kernel puts it back)
            const bool fold_v = h->fold_halo && h->comm && h->group.empty() && nvb != 0u && v_open && h->blk.nb_interior && nbnd <= h->fold_tile_limit;
            if (!fold_v && v_open && h->blk.nb_interior) { HP("wait V"); pjb_launch_wait(h->comm_stream, yv); }
            if ((rc = halo_wait(h, h->comm_stream))) return rc;   // in-process groups: the neighbours' transfers of the previous substep (events)
            if (fold_v) { HP("launch tet halo-side (awaits V)"); pjb_launch_tet_hwait(h->comm_stream, kb, h->blk.nb_interior, nbnd, yv, PJPeerSync(), h->pj.pos_pred + h->pj.nv_owned); }
            else { HP("launch tet halo-side"); pjb_launch_tet(h->comm_stream, kb, h->blk.nb_interior, nbnd); }
            if (!nvb) { HP("signal G"); pjb_launch_signal(h->comm_stream, yg); }
            if (!h->comm) { HP("record boundary"); HIPCHK(h, hipEventRecord(h->ev_boundary2[h->halo_parity], h->comm_stream)); }  // group transport: ghosts are free again
            if (nvb) { HP("launch vertex boundary"); pjb_launch_vertex(h->comm_stream, kb, 0, nvb, nullptr, nullptr, yg.flag, fold_v ? yv.flag : nullptr); }
            interior_particles(h, yg, ev);
            h->v_pending = true;
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
int flush_v(tetsim_body* h) {   // the last substep's V hand-over as kernels of its own: signal on the main queue, wait on the halo queue
    if (!h->v_pending && !h->p2p_raise_pending) return 0;
    if (!h->group.empty()) HIPCHK(h, hipSetDevice(h->opt.device));
    PJSync yv;
    yv.flag = h->d_sync + 2; yv.error = h->d_sync + 4; yv.timeout_ms = halo_timeout_ms(h);
    const bool g_folded = h->fold_wait && h->group.empty() && (h->pj.nv_owned - h->pj.nv_boundary + 63u) / 64u <= h->fold_wave_limit;   // (interior_particles' rule)
    if (h->v_pending) { HP("signal V"); pjb_launch_signal(h->stream, yv, g_folded ? h->d_sync + 0 : nullptr); }
    if (h->p2p) {   // ... and the "arrived" words of the last boundary-particle kernel, which no following substep's wait will raise
        PJPeerSync w;
        const uint32_t par = static_cast<uint32_t>(h->p2p_round & 1u);
        if (h->p2p_raise_pending && h->deep) {   // (an even substep's early message: set of the substep just enqueued)
            const uint32_t st = static_cast<uint32_t>(((h->p2p_round - 1u) >> 1) & 1u);
            for (const PeerLink& l : h->links) if (l.arrived2[st][1]) w.raise[w.n_raise++] = l.arrived2[st][1];
        } else if (h->p2p_raise_pending) for (const PeerLink& l : h->links) if (l.arrived[par]) w.raise[w.n_raise++] = l.arrived[par];
        if (!h->v_pending) yv.flag = nullptr;
        { HP("wait V + raise peers"); pjb_launch_wait_peers(h->comm_stream, yv, w); }
        h->p2p_raise_pending = false;
    } else { HP("wait V"); pjb_launch_wait(h->comm_stream, yv); }
    h->v_pending = false;
    return 0;
}
int enqueue_phase_b(tetsim_body* h, bool refresh) {  // halo start
    if (h->p2p && !refresh) return 0;   // peer-to-peer bodies: the boundary-particle kernel has stored the halo already
    if (!h->group.empty()) HIPCHK(h, hipSetDevice(h->opt.device));
    int rc = halo_start(h);
    if (rc) return rc;
    h->halo_parity ^= 1u;
    return 0;
}

// The flag path's two chains wait for each other through device words, so replaying them from graphs is only live if the two
// streams are served by independent hardware queues (a wait kernel at the head of one queue must not block the signal kernel
// that sits in the other).  Probe it once: a bounded wait on the halo stream, submitted BEFORE its signal on the main stream.
// The answer is only valid for the set of streams that existed when it was taken (HIP maps streams onto a few hardware queues as
// it sees fit): the probe is repeated whenever this library has created another stream in the process since (g_stream_generation),
// and a device-side wait that times out after a replay (tetsim_sync) drops the graphs for good.  Streams created by OTHER code in
// the process are invisible to this check: the supported deployment is one process per GPU that steps one partitioned body.
void drop_flag_graphs(tetsim_body* h) {
    for (auto& kv : h->flag_graphs) { (void)hipGraphExecDestroy(kv.second.first); (void)hipGraphExecDestroy(kv.second.second); }
    h->flag_graphs.clear();
}
int probe_queue_independence(tetsim_body* h) {
    const uint64_t gen = g_stream_generation.load();
    if (h->queues_probed && h->probe_generation == gen) return 0;
    h->queues_probed = true;
    h->probe_generation = gen;
    if (!h->d_sync || !h->comm_stream) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    PJSync y;
    y.flag = h->d_sync + 6; y.error = h->d_sync + 7; y.timeout_ms = 200;
    HIPCHK(h, hipMemset(h->d_sync + 6, 0, 2 * sizeof(uint32_t)));
    pjb_launch_wait(h->comm_stream, y);
    pjb_launch_signal(h->stream, y);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    uint32_t w[2] = {0, 0};
    HIPCHK(h, hipMemcpy(w, h->d_sync + 6, sizeof w, hipMemcpyDeviceToHost));
    h->queues_independent = w[1] == 0u;
    if (!h->queues_independent) {
        fprintf(stderr, "[tetsim] the halo stream and the main stream share a hardware queue: the halo path stays eager (no graph replay)\n");
        HIPCHK(h, hipMemset(h->d_sync + 6, 0, 2 * sizeof(uint32_t)));
        drop_flag_graphs(h);   // chains captured under an earlier, better answer must not be replayed any more
    }
    return 0;
}

// n substeps of an RCCL body on the flag path as TWO captured linear chains, one per stream, replayed side by side:
//   main:  [interior tiles (raise V) -> wait G -> interior particles] x n -> signal V
//   halo:  [H tiles -> boundary particles (raise G) -> transfer -> wait V] x n          (enqueue_phase_a)
// No fork/join edge exists between them (those cost ~6 us each inside a graph on this stack): the chains meet only through the
// binary-semaphore words, whose kernels take constant arguments.  Inside a chain a kernel boundary costs 1.6 us instead of the
// 2.7 us of an eager launch, and the host enqueues two graph launches per call instead of 9 operations per substep.
int step_n_flag_graphs(tetsim_body* h, uint32_t n) {
    // (peer-to-peer bodies: the chain of a call depends on the parity of its first substep -- which ghost buffer, which words)
    const uint32_t key = n | (h->p2p ? static_cast<uint32_t>(h->p2p_round & (h->deep ? 3u : 1u)) << 30 : 0u);
    auto it = h->flag_graphs.find(key);
    if (it == h->flag_graphs.end()) {
        hipGraph_t gm = nullptr, gh = nullptr;
        hipGraphExec_t em = nullptr, eh = nullptr;
        // The capture walks enqueue_phase_a / _b / flush_v, which advance the host's picture of the choreography (substep parity of the
        // peer-to-peer buffers and words, pending raises and hand-overs) as if the substeps had RUN.  If the capture fails nothing has
        // run: every error path below puts that picture back, so that the eager fallback (tetsim_step_n) starts from where the device
        // really is -- an odd number of phantom substeps would make it read the wrong-parity ghost buffers and wait for words nobody raises.
        struct Picture { uint64_t p2p_round; bool p2p_raise_pending, v_pending, halo_pending, fork_needed; uint32_t halo_parity; };
        const Picture before = {h->p2p_round, h->p2p_raise_pending, h->v_pending, h->halo_pending, h->fork_needed, h->halo_parity};
        auto undo = [&]() {
            h->p2p_round = before.p2p_round; h->p2p_raise_pending = before.p2p_raise_pending; h->v_pending = before.v_pending;
            h->halo_pending = before.halo_pending; h->fork_needed = before.fork_needed; h->halo_parity = before.halo_parity;
        };
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipStreamBeginCapture(h->comm_stream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { (void)hipStreamEndCapture(h->stream, &gm); if (gm) (void)hipGraphDestroy(gm); undo(); return fail(h, TETSIM_EHIP, std::string("begin capture (halo stream): ") + hipGetErrorString(e)); }
        int rc = 0;
        for (uint32_t i = 0; i < n && !rc; i++) {
            h->vel_dead = i + 1 != n;
            rc = enqueue_phase_a(h);
            h->vel_dead = false;
            if (!rc) rc = enqueue_phase_b(h);
        }
        if (!rc) rc = flush_v(h);
        h->v_pending = false;
        const hipError_t e1 = hipStreamEndCapture(h->stream, &gm), e2 = hipStreamEndCapture(h->comm_stream, &gh);
        if (!rc && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail(h, TETSIM_EHIP, std::string("end capture: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        if (!rc && hipGraphInstantiate(&em, gm, nullptr, nullptr, 0) != hipSuccess) rc = fail(h, TETSIM_EHIP, "graph instantiate (main chain) failed");
        if (!rc && hipGraphInstantiate(&eh, gh, nullptr, nullptr, 0) != hipSuccess) rc = fail(h, TETSIM_EHIP, "graph instantiate (halo chain) failed");
        if (gm) (void)hipGraphDestroy(gm);
        if (gh) (void)hipGraphDestroy(gh);
        if (rc) { if (em) (void)hipGraphExecDestroy(em); if (eh) (void)hipGraphExecDestroy(eh); undo(); return rc; }
        it = h->flag_graphs.emplace(key, std::make_pair(em, eh)).first;
        if (h->p2p) h->p2p_round -= n;   // (the capture advanced the counter; the replay below is what runs)
    }
    if (h->p2p) h->p2p_round += n;
    HIPCHK(h, hipGraphLaunch(it->second.second, h->comm_stream));
    HIPCHK(h, hipGraphLaunch(it->second.first, h->stream));
    h->halo_pending = true;
    return 0;
}

}  // namespace tetsim

extern "C" {

static int tetsim_group_step_n_impl(tetsim_handle* hs, uint32_t count, uint32_t n, double dt, const TetSimParams* params);
int tetsim_group_step_n(tetsim_handle* hs, uint32_t count, uint32_t n, double dt, const TetSimParams* params) {
    group_begin(hs, count);
    return group_result(hs, count, tetsim_group_step_n_impl(hs, count, n, dt, params));
}
static int tetsim_group_step_n_impl(tetsim_handle* hs, uint32_t count, uint32_t n, double dt, const TetSimParams* params) {
    if (!hs || count == 0) return TETSIM_EINVAL;
    for (uint32_t i = 0; i < count; i++) {
        tetsim_body* h = hs[i];
        if (!h || h->opt.part_count != static_cast<int32_t>(count) || h->opt.part_index != static_cast<int32_t>(i) || h->comm)
            return fail(h, TETSIM_EINVAL, "handles[i] must be partition i of a count-way decomposition without an RCCL communicator");
        if (h->opt.solver != TETSIM_SOLVER_POLAR_JACOBI) return fail(h, TETSIM_ESTATE, "POLAR_JACOBI only");
        if (!h->group.empty() && (h->group.size() != count || !std::equal(h->group.begin(), h->group.end(), hs)))
            return fail(h, TETSIM_ESTATE, "this partition was wired to other handles by its first tetsim_group_step_n call (a member of the group was destroyed or replaced): create the partitions anew");
        if (h->group.empty()) {  // first use: wire the group and give every partition its halo stream (on ITS device)
            h->group.assign(hs, hs + count);
            HIPCHK(h, hipSetDevice(h->opt.device));
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
            HIPCHK(hs[i], hipSetDevice(hs[i]->opt.device));
            // "my ghosts may be overwritten": flag bodies record it on the halo stream, which ensure_prediction put behind the re-prediction
            HIPCHK(hs[i], hipEventRecord(hs[i]->ev_boundary2[hs[i]->halo_parity], hs[i]->flag_sync ? hs[i]->comm_stream : hs[i]->stream));
        }
        for (uint32_t i = 0; i < count; i++) { int rc = enqueue_phase_b(hs[i], true); if (rc) return rc; }
    }
    static const bool dbg_sync = getenv("TETSIM_DEBUG_GROUP_SYNC") != nullptr;  // development: serialise every phase
    for (uint32_t s = 0; s < n; s++) {
        for (uint32_t i = 0; i < count; i++) { hs[i]->vel_dead = s + 1 != n; int rc = enqueue_phase_a(hs[i]); hs[i]->vel_dead = false; if (rc) return rc; }
        if (dbg_sync) (void)hipDeviceSynchronize();
        for (uint32_t i = 0; i < count; i++) { int rc = enqueue_phase_b(hs[i]); if (rc) return rc; }
        if (dbg_sync) (void)hipDeviceSynchronize();
    }
    for (uint32_t i = 0; i < count; i++) { int rc = flush_v(hs[i]); if (rc) return rc; }
    return 0;
}

static int tetsim_halo_exchange_local_impl(tetsim_handle* hs, uint32_t count);
int tetsim_halo_exchange_local(tetsim_handle* hs, uint32_t count) {
    group_begin(hs, count);
    return group_result(hs, count, tetsim_halo_exchange_local_impl(hs, count));
}
static int tetsim_halo_exchange_local_impl(tetsim_handle* hs, uint32_t count) {
    if (!hs || count == 0) return TETSIM_EINVAL;
    for (uint32_t i = 0; i < count; i++) {
        if (!hs[i] || hs[i]->opt.part_count != static_cast<int32_t>(count) || hs[i]->opt.part_index != static_cast<int32_t>(i))
            return fail(hs[i], TETSIM_EINVAL, "handles[i] must be partition i of a count-way decomposition");
    }
    // every partition must have finished its vertex kernel before anyone's ghosts are overwritten
    for (uint32_t i = 0; i < count; i++) { HIPCHK(hs[i], hipSetDevice(hs[i]->opt.device)); HIPCHK(hs[i], hipStreamSynchronize(hs[i]->stream)); }
    for (uint32_t i = 0; i < count; i++) {
        tetsim_body* src = hs[i];
        HIPCHK(src, hipSetDevice(src->opt.device));
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
    for (uint32_t i = 0; i < count; i++) { HIPCHK(hs[i], hipSetDevice(hs[i]->opt.device)); HIPCHK(hs[i], hipStreamSynchronize(hs[i]->stream)); }
    return 0;
}
}  // extern "C"
