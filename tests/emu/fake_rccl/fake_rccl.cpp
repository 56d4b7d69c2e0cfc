// TEST INFRASTRUCTURE -- a stand-in for librccl.so.1 for the emulated build (tests/emu): the ranks of a communicator are
// THREADS of one process, device memory is host memory, and a collective is a rendezvous -- it returns at once, like the real
// one, leaving a ticket on the caller's (emulated) stream that completes when the last rank has arrived and the bytes have
// been copied.  What it lets run on a CPU for the first time: the RCCL branches of galah_amd/csrc/comm.cpp (ncclAllGather,
// grouped ncclSend/ncclRecv with their per-peer counts, the async-error poll and the deadline of rccl_wait, ncclCommAbort).
// It CHECKS what real RCCL would hang or corrupt on: ranks entering different collectives, all-gather counts that differ,
// a send whose count differs from its receive.
//
// Failure injection (environment, read at every call):
//   FAKE_RCCL_ABORT_SEEN_BY_PEERS=1   after one rank's ncclCommAbort the peers' ncclCommGetAsyncError reports ncclRemoteError
//                                     (default: they see nothing -- the worst case, only a deadline gets them out)
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <time.h>
#include <vector>

namespace {

typedef void *(*pending_new_fn)(hipStream_t);
typedef void (*pending_complete_fn)(void *, int);
pending_new_fn pending_new = nullptr;
pending_complete_fn pending_complete = nullptr;

bool bind_emulator() {
    if (pending_new) return true;
    const char *lib = getenv("HIPEMU_LIB");
    void *h = lib ? dlopen(lib, RTLD_NOW | RTLD_NOLOAD) : nullptr;
    if (!h) { fprintf(stderr, "[fake_rccl] HIPEMU_LIB does not name the loaded emulator library\n"); return false; }
    pending_new = (pending_new_fn)dlsym(h, "hipemu_pending_new");
    pending_complete = (pending_complete_fn)dlsym(h, "hipemu_pending_complete");
    return pending_new && pending_complete;
}

struct Ticket {   // one rank's pending stream work: completes when `left` reaches 0 (all calls under the group's mutex)
    void *handle;
    int rank;
    std::atomic<int> left;
    std::atomic<int> failed{0};
    Ticket(void *h, int n, int r) : handle(h), rank(r), left(n) {}
    void part_done(bool fail) {
        if (fail) failed = 1;
        if (--left == 0) { pending_complete(handle, failed.load()); }
    }
};

struct Gather { int rank; const void *send; void *recv; size_t bytes; std::shared_ptr<Ticket> t; };
struct P2P { void *buf; size_t bytes; std::shared_ptr<Ticket> t; };

struct Group {
    std::mutex mu;
    int world = 0, joined = 0;
    std::atomic<int> aborted{0};
    std::string error;
    // all-gathers by sequence number (every rank issues its collectives in the same order)
    std::map<uint64_t, std::vector<Gather>> gathers;
    // point-to-point: (src, dst) -> queues of posted sends / receives, matched in order
    std::map<std::pair<int, int>, std::vector<P2P>> sends, recvs;
    std::vector<std::shared_ptr<Ticket>> open;   // tickets not yet complete (a rank's own are failed by its abort)
    void keep(const std::shared_ptr<Ticket> &t) {   // mu held
        open.erase(std::remove_if(open.begin(), open.end(), [](const std::shared_ptr<Ticket> &x) { return x->left.load() <= 0; }), open.end());
        open.push_back(t);
    }
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<Group>> g_groups;
std::atomic<uint64_t> g_id{1};

}  // namespace

struct ncclComm {
    std::shared_ptr<Group> g;
    int rank = 0;
    uint64_t next_gather = 0;
    bool aborted = false;
};

namespace {
thread_local int t_group_depth = 0;
struct Queued { bool is_send; void *buf; size_t bytes; int peer; ncclComm *comm; hipStream_t stream; };
thread_local std::vector<Queued> t_queue;

size_t type_size(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclInt32: case ncclUint32: return 4; default: return 8; }
}

void match_p2p(Group &g, int src, int dst) {   // g.mu held
    auto &s = g.sends[{src, dst}];
    auto &r = g.recvs[{src, dst}];
    while (!s.empty() && !r.empty()) {
        P2P a = s.front(), b = r.front();
        s.erase(s.begin());
        r.erase(r.begin());
        const bool bad = a.bytes != b.bytes;
        if (bad) {
            g.error = "ncclSend of " + std::to_string(a.bytes) + " bytes from rank " + std::to_string(src) + " meets an ncclRecv of " + std::to_string(b.bytes) + " on rank " + std::to_string(dst);
            fprintf(stderr, "[fake_rccl] %s (real RCCL would hang or corrupt)\n", g.error.c_str());
        } else if (a.bytes) memcpy(b.buf, a.buf, a.bytes);
        a.t->part_done(bad);
        b.t->part_done(bad);
    }
}

ncclResult_t flush_queue() {
    if (t_queue.empty()) return ncclSuccess;
    if (!bind_emulator()) return ncclSystemError;
    // one ticket per (communicator, stream) of the group call: this rank's sends and receives complete together
    std::vector<Queued> q;
    q.swap(t_queue);
    ncclComm *comm = q[0].comm;
    hipStream_t stream = q[0].stream;
    for (auto &e : q) if (e.comm != comm || e.stream != stream) { fprintf(stderr, "[fake_rccl] one group, several communicators or streams: not modelled\n"); return ncclInvalidUsage; }
    Group &g = *comm->g;
    auto ticket = std::make_shared<Ticket>(pending_new(stream), (int)q.size(), comm->rank);
    std::lock_guard<std::mutex> lk(g.mu);
    g.keep(ticket);
    for (auto &e : q) {
        if (e.peer < 0 || e.peer >= g.world) { ticket->part_done(true); continue; }
        if (e.is_send) { g.sends[{comm->rank, e.peer}].push_back(P2P{e.buf, e.bytes, ticket}); match_p2p(g, comm->rank, e.peer); }
        else { g.recvs[{e.peer, comm->rank}].push_back(P2P{e.buf, e.bytes, ticket}); match_p2p(g, e.peer, comm->rank); }
    }
    return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu", (unsigned long long)g_id.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!bind_emulator()) return ncclSystemError;
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &slot = g_groups[std::string(id.internal)];
        if (!slot) { slot = std::make_shared<Group>(); slot->world = nranks; }
        g = slot;
    }
    if (g->world != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->joined++;
    }
    // like the real call: returns when every rank has joined (bounded: a test that forgets a rank gets an error, not a hang)
    for (int spins = 0; spins < 600000; spins++) {
        { std::lock_guard<std::mutex> lk(g->mu); if (g->joined >= nranks) break; }
        struct timespec ts = {0, 100000};
        nanosleep(&ts, nullptr);
    }
    { std::lock_guard<std::mutex> lk(g->mu); if (g->joined < nranks) return ncclSystemError; }
    ncclComm *c = new ncclComm();
    c->g = g;
    c->rank = rank;
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }

ncclResult_t ncclCommAbort(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    Group &g = *c->g;
    {
        // THIS rank's work in flight ends, failed -- the aborted kernel leaves its stream.  Its peers notice nothing by
        // themselves: what they have in flight with this rank stays in flight (a deadline, or ncclCommGetAsyncError when
        // FAKE_RCCL_ABORT_SEEN_BY_PEERS is set, is their way out), exactly the situation rccl_wait exists for.
        std::lock_guard<std::mutex> lk(g.mu);
        g.aborted = 1;
        for (auto &t : g.open)
            if (t->rank == c->rank) while (t->left.load() > 0) t->part_done(true);
        for (auto &kv : g.gathers) kv.second.erase(std::remove_if(kv.second.begin(), kv.second.end(), [&](const Gather &x) { return x.rank == c->rank; }), kv.second.end());
        for (auto *m : {&g.sends, &g.recvs})
            for (auto &kv : *m) kv.second.erase(std::remove_if(kv.second.begin(), kv.second.end(), [&](const P2P &x) { return x.t->rank == c->rank; }), kv.second.end());
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t *e) {
    *e = ncclSuccess;
    if (c && c->g->aborted.load()) { const char *v = getenv("FAKE_RCCL_ABORT_SEEN_BY_PEERS"); if (v && *v == '1') *e = ncclRemoteError; }
    if (c && !c->g->error.empty()) *e = ncclInternalError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "unhandled system error";
    case ncclInternalError: return "internal error";
    case ncclInvalidArgument: return "invalid argument";
    case ncclInvalidUsage: return "invalid usage";
    case ncclRemoteError: return "remote process exited or there was a network error";
    default: return "unknown result code";
    }
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
    if (!c || !bind_emulator()) return ncclInvalidArgument;
    Group &g = *c->g;
    const size_t bytes = count * type_size(type);
    auto ticket = std::make_shared<Ticket>(pending_new(stream), 1, c->rank);
    std::lock_guard<std::mutex> lk(g.mu);
    g.keep(ticket);
    auto &v = g.gathers[c->next_gather++];
    v.push_back(Gather{c->rank, send, recv, bytes, ticket});
    if ((int)v.size() == g.world) {   // the last rank to arrive moves the bytes for everyone (it is all host memory)
        bool bad = false;
        for (auto &x : v) bad = bad || x.bytes != bytes;
        if (bad) { g.error = "ncclAllGather: the ranks pass different counts"; fprintf(stderr, "[fake_rccl] %s (real RCCL would hang or corrupt)\n", g.error.c_str()); }
        else
            for (auto &dst : v)
                for (auto &src : v)
                    if (bytes) memmove((char *)dst.recv + (size_t)src.rank * bytes, src.send, bytes);
        std::vector<Gather> done;
        done.swap(v);
        g.gathers.erase(c->next_gather - 1);
        for (auto &x : done) x.t->part_done(bad);
    }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    t_queue.push_back(Queued{true, const_cast<void *>(send), count * type_size(type), peer, c, stream});
    return t_group_depth ? ncclSuccess : flush_queue();
}
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    t_queue.push_back(Queued{false, recv, count * type_size(type), peer, c, stream});
    return t_group_depth ? ncclSuccess : flush_queue();
}
ncclResult_t ncclGroupStart() { t_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_group_depth == 0) return ncclInvalidUsage;
    return --t_group_depth == 0 ? flush_queue() : ncclSuccess;
}
}
