// mock_rccl.cpp -- TEST DOUBLE for librccl (tests only; loaded through TETSIM_RCCL_LIB).
//
// The multi-GPU halo path of libtetsim_hip talks to RCCL through ten entry points resolved with dlopen.  A real
// multi-rank run needs more than one GPU, which the test box does not have, so this file implements those ten entry
// points for ranks that live in ONE process on ONE device (one host thread per rank), strictly enough to catch misuse:
//   * every ncclSend must meet a ncclRecv of the same element count and type from the addressed peer, in order
//     (otherwise: error / 30 s rendezvous timeout instead of silent corruption);
//   * point-to-point calls outside a group, unknown peers, null buffers, a destroyed communicator -> ncclInvalidUsage;
//   * stream semantics: a transfer starts after everything already enqueued on the sender's stream AND on the receiver's
//     stream, and everything enqueued later on either stream waits for it -- what NCCL guarantees for stream-ordered p2p.
// Unlike NCCL, ncclGroupEnd blocks the host thread until the peer has posted its half (ranks are threads here).
//   hipcc -shared -fPIC -O1 tests/mock_rccl/mock_rccl.cpp -o tests/mock_rccl/libmock_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct SendOp {
    const void* src; size_t count; ncclDataType_t type;
    hipEvent_t ready = nullptr, done = nullptr;
    bool done_recorded = false;
};
struct World {
    int nranks = 0, joined = 0, destroyed = 0;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<SendOp>>> mail;  // (src, dst) -> posted sends, FIFO
};
struct Comm { World* world; int rank; bool alive = true; };
struct Pending { bool is_send; const void* sbuf; void* rbuf; size_t count; ncclDataType_t type; int peer; Comm* comm; hipStream_t stream; };

std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, World*> g_worlds;
unsigned g_uid_counter = 0;
thread_local int t_depth = 0;
thread_local std::vector<Pending> t_ops;
constexpr auto kTimeout = std::chrono::seconds(30);

size_t type_size(ncclDataType_t t) { return t == ncclFloat ? 4 : t == ncclInt32 ? 4 : t == ncclUint8 ? 1 : 0; }

ncclResult_t flush() {
    std::vector<Pending> ops;
    ops.swap(t_ops);
    std::vector<std::shared_ptr<SendOp>> mine;
    std::unique_lock<std::mutex> lk(g_mu);
    // 1. post every send: "the data is ready once the sender's stream gets here"
    for (auto& p : ops) {
        if (!p.is_send) continue;
        auto op = std::make_shared<SendOp>();
        op->src = p.sbuf; op->count = p.count; op->type = p.type;
        if (hipEventCreateWithFlags(&op->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&op->done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(op->ready, p.stream) != hipSuccess) return ncclUnhandledCudaError;
        p.comm->world->mail[{p.comm->rank, p.peer}].push_back(op);
        mine.push_back(op);
    }
    g_cv.notify_all();
    // 2. serve every receive: wait for the matching send, enqueue the copy on OUR stream behind the sender's "ready"
    for (auto& p : ops) {
        if (p.is_send) continue;
        auto& q = p.comm->world->mail[{p.peer, p.comm->rank}];
        if (!g_cv.wait_for(lk, kTimeout, [&] { return !q.empty(); })) return ncclInternalError;  // no matching ncclSend
        auto op = q.front();
        q.pop_front();
        if (op->count != p.count || op->type != p.type) return ncclInvalidArgument;              // mismatched message
        if (hipStreamWaitEvent(p.stream, op->ready, 0) != hipSuccess ||
            hipMemcpyAsync(p.rbuf, op->src, p.count * type_size(p.type), hipMemcpyDeviceToDevice, p.stream) != hipSuccess ||
            hipEventRecord(op->done, p.stream) != hipSuccess) return ncclUnhandledCudaError;
        op->done_recorded = true;
        g_cv.notify_all();
    }
    // 3. our sends: later work on the sending stream must wait until the receiver has taken the data
    size_t i = 0;
    for (auto& p : ops) {
        if (!p.is_send) continue;
        auto op = mine[i++];
        if (!g_cv.wait_for(lk, kTimeout, [&] { return op->done_recorded; })) return ncclInternalError;  // nobody received it
        if (hipStreamWaitEvent(p.stream, op->done, 0) != hipSuccess) return ncclUnhandledCudaError;
        // events are released lazily: destroying an event with pending waits is legal in HIP (resources freed on completion)
        (void)hipEventDestroy(op->ready);
        (void)hipEventDestroy(op->done);
    }
    return ncclSuccess;
}

ncclResult_t enqueue(const Pending& p) {
    if (!p.comm || !p.comm->alive) return ncclInvalidUsage;
    if (t_depth == 0) return ncclInvalidUsage;                       // libtetsim always groups its point-to-point calls
    if (p.peer < 0 || p.peer >= p.comm->world->nranks) return ncclInvalidArgument;
    if ((p.is_send ? p.sbuf == nullptr : p.rbuf == nullptr) || type_size(p.type) == 0) return ncclInvalidArgument;
    t_ops.push_back(p);
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof(*id));
    const unsigned n = ++g_uid_counter;
    std::memcpy(id->internal, "mock-rccl", 9);
    std::memcpy(id->internal + 16, &n, sizeof n);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lk(g_mu);
    const std::string key(id.internal, sizeof(id.internal));
    World*& w = g_worlds[key];
    if (!w) { w = new World(); w->nranks = nranks; }
    if (w->nranks != nranks) return ncclInvalidArgument;
    w->joined++;
    g_cv.notify_all();
    World* world = w;
    if (!g_cv.wait_for(lk, kTimeout, [&] { return world->joined >= nranks; })) return ncclInternalError;  // a rank never showed up
    *comm = reinterpret_cast<ncclComm_t>(new Comm{world, rank});
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !c->alive) return ncclInvalidArgument;
    c->alive = false;   // kept allocated on purpose: a use-after-destroy is reported, not a crash
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (!c || !c->alive || !count) return ncclInvalidArgument;
    *count = c->world->nranks;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (!c || !c->alive || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_depth == 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return flush();
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(Pending{true, buf, nullptr, count, type, peer, reinterpret_cast<Comm*>(comm), stream});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(Pending{false, nullptr, buf, count, type, peer, reinterpret_cast<Comm*>(comm), stream});
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "success";
        case ncclInvalidUsage: return "mock-rccl: invalid usage (ungrouped p2p call or dead communicator)";
        case ncclInvalidArgument: return "mock-rccl: invalid argument (peer, buffer, type or message size mismatch)";
        case ncclInternalError: return "mock-rccl: rendezvous timed out (unmatched send/recv or missing rank)";
        case ncclUnhandledCudaError: return "mock-rccl: HIP call failed";
        default: return "mock-rccl: error";
    }
}

}  // extern "C"
