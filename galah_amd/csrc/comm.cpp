// Multi-GPU exchange behind the C ABI (include/galah_hip.h "multi-GPU"): the two real exchange steps of the path --
// all-gather of the packed sketch matrix, gather of candidate lists / ANI index slices / ANI values -- over one of
// three transports that share every line of the logic above the three primitives (all-gather of device bytes,
// all-gather of host bytes, all-to-all-v of device bytes):
//   RCCL      one process per GPU: ncclAllGather (and grouped ncclSend / ncclRecv for the ANI index slices, which go
//             only where they are wanted) over xGMI on the context's stream.  librccl is bound at run time
//             (dlopen), so the library loads on hosts without it and shares the copy a host such as PyTorch mapped.
//   LOCAL     one process, one context (and one thread) per GPU -- how a single-process host like galah's CLI drives
//             8 devices: every rank pulls its peers' blocks with hipMemcpyPeerAsync (xGMI peer copies).
//   CALLBACK  the host supplies an all-gather of host bytes (MPI, gloo, a test harness); device payloads are staged
//             through host memory.  The functional fallback, and what the CPU tests drive.
// Reference: the finch path has no multi-process form (src/finch.rs:74-96 is one serial loop); this is the sharding
// SURVEY.md section 8(e) specifies: genomes in contiguous blocks, full sketch matrix everywhere, pair work dealt over
// the ranks, a pair's ANI computed where its first genome lives.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <thread>

#include "ghip_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------- RCCL, bound lazily
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;   // optional
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;   // optional
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        // GHIP_RCCL_LIBRARY: the RCCL build to use, by path (a site's own build; the test suite's stand-in) -- the names below
        // resolve to whatever librccl the process has mapped already
        const char *chosen = getenv("GHIP_RCCL_LIBRARY");
        if (chosen && *chosen) {
            x.handle = dlopen(chosen, RTLD_NOW | RTLD_GLOBAL);
            if (!x.handle) { x.err = std::string("GHIP_RCCL_LIBRARY: ") + (dlerror() ? dlerror() : chosen); return x; }
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (x.handle) break;
            x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!x.handle) { x.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return x; }
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
        x.AllGather = (decltype(x.AllGather))dlsym(x.handle, "ncclAllGather");
        x.Send = (decltype(x.Send))dlsym(x.handle, "ncclSend");
        x.Recv = (decltype(x.Recv))dlsym(x.handle, "ncclRecv");
        x.GroupStart = (decltype(x.GroupStart))dlsym(x.handle, "ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))dlsym(x.handle, "ncclGroupEnd");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
        x.CommAbort = (decltype(x.CommAbort))dlsym(x.handle, "ncclCommAbort");
        x.CommGetAsyncError = (decltype(x.CommGetAsyncError))dlsym(x.handle, "ncclCommGetAsyncError");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.handle, "ncclGetErrorString");
        if (!x.GetUniqueId || !x.CommInitRank || !x.AllGather || !x.Send || !x.Recv || !x.GroupStart || !x.GroupEnd || !x.CommDestroy ||
            !x.GetErrorString)
            x.err = "librccl lacks an expected symbol";
        return x;
    }();
    return r;
}

// ---------------------------------------------------------------------------------------------- LOCAL transport state
struct LocalGroup {
    uint32_t world = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> slot;   // what each rank published for the collective in flight
    std::vector<std::vector<uint64_t>> offer;   // exchange_device: each rank's send offsets (copies: they outlive a failing rank's frame)
    std::vector<int> device;
    std::atomic<int> refs{0};
    std::atomic<int> failed{0};       // a rank hit an error: every later barrier returns it instead of waiting for ever

    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen || failed.load(); });
        return failed.load() == 0;
    }
    void fail() { failed = 1; std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
};

enum Transport { T_SELF = 0, T_RCCL = 1, T_LOCAL = 2, T_CALLBACK = 3 };

}  // namespace

struct ghip_comm {
    ghip_ctx *ctx = nullptr;   // may be NULL for a CALLBACK communicator used for host payloads only
    uint32_t rank = 0, world = 1;
    Transport transport = T_SELF;
    ncclComm_t nccl = nullptr;
    bool arriving = false;         // inside agree(): the collective under way is the one the ranks ARRIVE at (see rccl_wait)
    bool agreed_failure = false;   // the error a call is returning was agreed by every rank (agree(), exchange 1 of the pair stage): the caller must not agree on it again
    bool dead = false;   // an RCCL call failed: the communicator was aborted, every later collective returns an error at once
    LocalGroup *group = nullptr;
    ghip_allgather_fn fn = nullptr;
    void *user = nullptr;
    std::string err;
    // CALLBACK transport: pinned staging of device payloads, kept and grown on demand.  (Pageable staging made the runtime pin
    // and unpin the vectors around every copy: the NEXT small device-to-host copy of the caller then stalled 12-24 ms.)
    void *pin_send = nullptr, *pin_recv = nullptr;
    size_t pin_send_bytes = 0, pin_recv_bytes = 0;
    // RCCL transport, made once at ghip_comm_init_rank so that nothing can fail on one rank between "decided to enter a
    // collective" and "entered it": the event a rank polls behind every collective, and device + pinned staging of the
    // SMALL host payloads (status words, sizes, flags: up to HC_BYTES per rank)
    hipEvent_t wait_ev = nullptr;
    void *hc_send = nullptr, *hc_recv = nullptr, *hc_pin = nullptr;
    static constexpr size_t HC_BYTES = 4096;
};

namespace {

int cerr_(ghip_comm *c, int code, const std::string &msg) {
    c->err = msg;
    ghip_set_error(c->ctx, code, msg);   // (no context: the thread's context-less error text, what ghip_comm_note_error reads back)
    if (c->group) c->group->fail();
    return code;
}

// an RCCL call itself failed: the communicator is unusable.  Abort it (peers blocked on this rank inside RCCL are released
// by their own error path rather than waiting for a collective this rank will never join) and refuse every later call.
int rccl_failed(ghip_comm *c, const char *what, ncclResult_t r) {
    const std::string msg = std::string(what) + ": " + rccl().GetErrorString(r);
    if (c->nccl && rccl().CommAbort) { rccl().CommAbort(c->nccl); c->nccl = nullptr; }
    c->dead = true;
    return cerr_(c, GHIP_EHIP, msg);
}

#define COMM_HIP(c, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return cerr_((c), GHIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Waits for the RCCL work just put on the context's stream WITHOUT handing the thread to hipStreamSynchronize: a peer that died,
// or never enters the collective, would hold this rank inside it for ever (the status words cover failures BETWEEN collectives,
// not inside one).  The stream's event is polled next to ncclCommGetAsyncError -- RCCL's own report of a broken link or a
// remote abort -- and, when the caller asked for one, against a deadline (ghip_options.comm_timeout_ms; 0, the default: none).
// Either way out: ncclCommAbort (which also releases the kernel RCCL has on the stream), the communicator is dead, the caller
// gets GHIP_EPEER.
//
// What the deadline means.  It counts from the moment THIS rank entered the collective, so it also bounds how late a healthy
// peer may be.  Ranks are late for the status-word exchange of a phase boundary (agree()): that is where an uneven shard, a
// cold page cache under one rank's files or a retry on one rank shows.  They are not late for the data collectives behind
// it -- every rank has just been seen at the boundary and nothing but a few launches stands between it and the collective.
// Hence: the data collectives get comm_timeout_ms, the boundary GHIP_COMM_ARRIVAL_FACTOR times that.  With the default of 0
// nothing is timed: RCCL's own report and the stream's are the failure signals, and a peer that vanished without either is
// the job launcher's to notice (torchrun and mpirun tear the job down when a rank exits).
constexpr uint64_t GHIP_COMM_ARRIVAL_FACTOR = 10;
int rccl_wait(ghip_comm *c, const char *what) {
    ghip_ctx *ctx = c->ctx;
    if (!c->wait_ev) {   // (a communicator made before the event could be: plain wait)
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));
        return GHIP_OK;
    }
    COMM_HIP(c, hipEventRecord(c->wait_ev, ctx->stream));
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t limit_ms = (uint64_t)ctx->opt.comm_timeout_ms * (c->arriving ? GHIP_COMM_ARRIVAL_FACTOR : 1);
    auto give_up = [&](const std::string &why) {
        if (c->nccl && rccl().CommAbort) { rccl().CommAbort(c->nccl); c->nccl = nullptr; }
        c->dead = true;
        (void)hipStreamSynchronize(ctx->stream);   // the aborted kernel leaves the stream
        (void)hipGetLastError();
        return cerr_(c, GHIP_EPEER, std::string(what) + ": " + why + " -- the communicator was aborted");
    };
    for (uint64_t spins = 0;; spins++) {
        const hipError_t q = hipEventQuery(c->wait_ev);
        if (q == hipSuccess) return GHIP_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return give_up(std::string("the stream reports ") + hipGetErrorString(q)); }
        (void)hipGetLastError();   // (hipErrorNotReady is sticky in hipGetLastError)
        if (rccl().CommGetAsyncError && c->nccl && (spins & 15) == 0) {
            ncclResult_t async = ncclSuccess;
            const ncclResult_t r = rccl().CommGetAsyncError(c->nccl, &async);
            if (r != ncclSuccess || (async != ncclSuccess && async != ncclInProgress))
                return give_up(std::string("RCCL reports ") + rccl().GetErrorString(r != ncclSuccess ? r : async));
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (limit_ms && ms > (double)limit_ms)
            return give_up("no answer from the peers within " + std::to_string(limit_ms) + " ms (ghip_options.comm_timeout_ms" +
                           (c->arriving ? " x " + std::to_string(GHIP_COMM_ARRIVAL_FACTOR) + " at a phase boundary)" : ")"));
        if (ms < 2.0) std::this_thread::yield();   // a collective of this path takes 0.05-10 ms: stay close at first
        else std::this_thread::sleep_for(std::chrono::microseconds(ms < 50.0 ? 50 : 500));
    }
}

// tests of the error paths: ghip_options.fault_stage / fault_rank make one rank fail at a named point
bool fault_here(const ghip_comm *c, uint32_t stage) {
    return c->ctx && c->ctx->opt.fault_stage == stage && c->ctx->opt.fault_rank == c->rank;
}

struct PoolBuf {  // device scratch from the context's pool
    ghip_ctx *ctx;
    void *p = nullptr;
    PoolBuf(ghip_ctx *c, size_t bytes) : ctx(c) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipSetDevice(ctx->device);
        p = ghip_pool_alloc(ctx, std::max<size_t>(bytes, 16));
    }
    ~PoolBuf() {
        if (!p) return;
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStreamSynchronize(ctx->stream);
        ghip_pool_free(ctx, p);
    }
    void *release() { void *q = p; p = nullptr; return q; }
};

// every rank contributes `bytes` from d_send; d_recv (world * bytes) receives the blocks in rank order
int allgather_device(ghip_comm *c, const void *d_send, void *d_recv, size_t bytes) {
    if (bytes == 0) return GHIP_OK;
    ghip_ctx *ctx = c->ctx;
    if (!ctx) return cerr_(c, GHIP_EINVAL, "this communicator has no device context");
    COMM_HIP(c, hipSetDevice(ctx->device));
    switch (c->transport) {
    case T_SELF:
        COMM_HIP(c, hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return GHIP_OK;
    case T_RCCL: {
        if (c->dead) return cerr_(c, GHIP_EHIP, "the RCCL communicator was aborted by an earlier failure");
        ncclResult_t r = rccl().AllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, ctx->stream);
        if (r != ncclSuccess) return rccl_failed(c, "ncclAllGather", r);
        return rccl_wait(c, "ncclAllGather");
    }
    case T_LOCAL: {
        LocalGroup *g = c->group;
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));   // my block is written before anyone reads it
        g->slot[c->rank] = d_send;
        if (!g->barrier()) return cerr_(c, GHIP_EHIP, "a peer rank failed");
        for (uint32_t r = 0; r < c->world; r++)           // pull: own block first is as good as any order
            COMM_HIP(c, hipMemcpyPeerAsync((char *)d_recv + (size_t)r * bytes, ctx->device, g->slot[r], g->device[r], bytes, ctx->stream));
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));
        if (!g->barrier()) return cerr_(c, GHIP_EHIP, "a peer rank failed");   // peers are done reading my block
        return GHIP_OK;
    }
    case T_CALLBACK: {
        auto grow = [&](void *&p, size_t &have, size_t want) {
            if (have >= want) return true;
            if (p) hipHostFree(p);
            p = nullptr; have = 0;
            const size_t cap = std::max<size_t>(want + want / 4, 1u << 16);
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return false; }
            have = cap;
            return true;
        };
        if (!grow(c->pin_send, c->pin_send_bytes, bytes) || !grow(c->pin_recv, c->pin_recv_bytes, bytes * c->world))
            return cerr_(c, GHIP_ENOMEM, "out of pinned host memory for the host-callback transport");
        COMM_HIP(c, hipMemcpyAsync(c->pin_send, d_send, bytes, hipMemcpyDeviceToHost, ctx->stream));
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));
        if (c->fn(c->user, c->pin_send, bytes, c->pin_recv) != 0) return cerr_(c, GHIP_ECALLBACK, "the host all-gather callback failed");
        COMM_HIP(c, hipMemcpyAsync(d_recv, c->pin_recv, bytes * c->world, hipMemcpyHostToDevice, ctx->stream));
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));
        return GHIP_OK;
    }
    }
    return GHIP_EINVAL;
}

int allgather_host(ghip_comm *c, const void *send, size_t bytes, void *recv);
int agree(ghip_comm *c, int status, const char *what);

// All-to-all-v of device bytes: my bytes [send_off[d], send_off[d + 1]) of d_send arrive at rank d's d_recv + its
// recv_off[me]; recv_off[r + 1] - recv_off[r] is what I expect from rank r (both arrays hold world + 1 offsets, and the
// two sides of every (sender, receiver) must agree on the size -- the callers derive both from data every rank holds).
// Every rank calls it, also with nothing to send or receive.
//   RCCL      one ncclSend and one ncclRecv per peer with bytes, in one group: a slice crosses exactly one xGMI link
//   LOCAL     every rank pulls its parts out of the peers' send buffers (hipMemcpyPeerAsync)
//   CALLBACK  the host all-gather is the only collective there is: whole send buffers are gathered, a rank keeps its parts
int exchange_device(ghip_comm *c, const void *d_send, const uint64_t *send_off, void *d_recv, const uint64_t *recv_off) {
    ghip_ctx *ctx = c->ctx;
    if (!ctx) return cerr_(c, GHIP_EINVAL, "this communicator has no device context");
    const uint32_t me = c->rank, world = c->world;
    COMM_HIP(c, hipSetDevice(ctx->device));
    auto sends = [&](uint32_t d) { return send_off[d + 1] - send_off[d]; };
    auto recvs = [&](uint32_t r) { return recv_off[r + 1] - recv_off[r]; };
    switch (c->transport) {
    case T_SELF:
        if (sends(0) != recvs(0)) return cerr_(c, GHIP_EINVAL, "exchange: send and receive sizes differ");
        if (sends(0)) COMM_HIP(c, hipMemcpyAsync((char *)d_recv + recv_off[0], (const char *)d_send + send_off[0], sends(0), hipMemcpyDeviceToDevice, ctx->stream));
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));
        return GHIP_OK;
    case T_RCCL: {
        if (c->dead) return cerr_(c, GHIP_EHIP, "the RCCL communicator was aborted by an earlier failure");
        ncclResult_t r = rccl().GroupStart();
        for (uint32_t p = 0; p < world && r == ncclSuccess; p++) {
            if (sends(p)) r = rccl().Send((const char *)d_send + send_off[p], sends(p), ncclUint8, (int)p, c->nccl, ctx->stream);
            if (r == ncclSuccess && recvs(p)) r = rccl().Recv((char *)d_recv + recv_off[p], recvs(p), ncclUint8, (int)p, c->nccl, ctx->stream);
        }
        const ncclResult_t e = rccl().GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) return rccl_failed(c, "ncclSend/ncclRecv", r);
        return rccl_wait(c, "ncclSend/ncclRecv");
    }
    case T_LOCAL: {
        LocalGroup *g = c->group;
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));   // my send buffer is written before anyone reads it
        g->slot[me] = d_send;
        g->offer[me].assign(send_off, send_off + world + 1);
        if (!g->barrier()) return cerr_(c, GHIP_EHIP, "a peer rank failed");
        bool mismatch = false;
        hipError_t e = hipSuccess;
        for (uint32_t r = 0; r < world && e == hipSuccess; r++) {
            const std::vector<uint64_t> &o = g->offer[r];
            const uint64_t b = o[me + 1] - o[me];
            if (b != recvs(r)) { mismatch = true; continue; }
            if (b) e = hipMemcpyPeerAsync((char *)d_recv + recv_off[r], ctx->device, (const char *)g->slot[r] + o[me], g->device[r], b, ctx->stream);
        }
        const hipError_t es = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = es;
        const bool peers_ok = g->barrier();   // peers are done reading my buffer
        if (e != hipSuccess) return cerr_(c, GHIP_EHIP, std::string("exchange (peer copy): ") + hipGetErrorString(e));
        if (mismatch) return cerr_(c, GHIP_EINVAL, "exchange: a peer offers a different size than this rank expects");
        if (!peers_ok) return cerr_(c, GHIP_EHIP, "a peer rank failed");
        return GHIP_OK;
    }
    case T_CALLBACK: {
        std::vector<uint64_t> all_off((size_t)(world + 1) * world);
        int rc = allgather_host(c, send_off, (size_t)(world + 1) * sizeof(uint64_t), all_off.data());
        if (rc) return rc;
        uint64_t m = 0;
        bool mismatch = false;
        for (uint32_t r = 0; r < world; r++) {
            const uint64_t *o = &all_off[(size_t)r * (world + 1)];
            m = std::max(m, o[world] - o[0]);
            if (o[me + 1] - o[me] != recvs(r)) mismatch = true;
        }
        // a rank that cannot take part -- no staging memory, or a peer offers it another size than it expects -- says so in a
        // status word first: nobody enters the device collective alone, nobody leaves the exchange alone
        int mine = mismatch ? cerr_(c, GHIP_EINVAL, "exchange: a peer offers a different size than this rank expects") : GHIP_OK;
        if (m == 0) return agree(c, mine, "an empty exchange");
        m = (m + 15) / 16 * 16;
        PoolBuf sb(ctx, m), rb(ctx, m * world);
        if (!mine && (!sb.p || !rb.p)) mine = cerr_(c, GHIP_EHIP, "out of device memory for staging the exchange");
        if (!mine && send_off[world] > send_off[0] &&
            hipMemcpyAsync(sb.p, (const char *)d_send + send_off[0], send_off[world] - send_off[0], hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            mine = cerr_(c, GHIP_EHIP, "staging the exchange: copy failed");
        if ((rc = agree(c, mine, "staging the exchange"))) return rc;
        if ((rc = allgather_device(c, sb.p, rb.p, m))) return rc;
        for (uint32_t r = 0; r < world; r++) {
            const uint64_t *o = &all_off[(size_t)r * (world + 1)];
            if (recvs(r)) COMM_HIP(c, hipMemcpyAsync((char *)d_recv + recv_off[r], (const char *)rb.p + (size_t)r * m + (o[me] - o[0]), recvs(r), hipMemcpyDeviceToDevice, ctx->stream));
        }
        COMM_HIP(c, hipStreamSynchronize(ctx->stream));
        return GHIP_OK;
    }
    }
    return GHIP_EINVAL;
}

int allgather_host(ghip_comm *c, const void *send, size_t bytes, void *recv) {
    if (bytes == 0) return GHIP_OK;
    switch (c->transport) {
    case T_SELF: memcpy(recv, send, bytes); return GHIP_OK;
    case T_CALLBACK:
        if (c->fn(c->user, send, bytes, recv) != 0) return cerr_(c, GHIP_ECALLBACK, "the host all-gather callback failed");
        return GHIP_OK;
    case T_LOCAL: {
        LocalGroup *g = c->group;
        g->slot[c->rank] = send;
        if (!g->barrier()) return cerr_(c, GHIP_EHIP, "a peer rank failed");
        for (uint32_t r = 0; r < c->world; r++) memcpy((char *)recv + (size_t)r * bytes, g->slot[r], bytes);
        if (!g->barrier()) return cerr_(c, GHIP_EHIP, "a peer rank failed");
        return GHIP_OK;
    }
    case T_RCCL: {  // host payloads ride the same collective
        ghip_ctx *ctx = c->ctx;
        const size_t padded = (bytes + 15) / 16 * 16;
        COMM_HIP(c, hipSetDevice(ctx->device));
        if (padded <= ghip_comm::HC_BYTES && c->hc_send) {
            // small payloads (every status word is one): buffers made with the communicator -- no allocation, hence nothing
            // that could fail on this rank alone, stands between the call and the collective
            memcpy(c->hc_pin, send, bytes);
            COMM_HIP(c, hipMemcpyAsync(c->hc_send, c->hc_pin, padded, hipMemcpyHostToDevice, ctx->stream));
            int rc = allgather_device(c, c->hc_send, c->hc_recv, padded);
            if (rc) return rc;
            COMM_HIP(c, hipMemcpyAsync(c->hc_pin, c->hc_recv, padded * c->world, hipMemcpyDeviceToHost, ctx->stream));
            COMM_HIP(c, hipStreamSynchronize(ctx->stream));
            for (uint32_t r = 0; r < c->world; r++) memcpy((char *)recv + (size_t)r * bytes, (const char *)c->hc_pin + (size_t)r * padded, bytes);
            return GHIP_OK;
        }
        // larger ones (a lazy round's answers): the buffers can only be sized now, and a rank that cannot get them says so in a
        // status word (a small payload: the branch above) before anyone enters the collective
        PoolBuf ds(ctx, padded), dr(ctx, padded * c->world);
        int rc = (!ds.p || !dr.p) ? cerr_(c, GHIP_ENOMEM, "out of device memory for a host collective") : GHIP_OK;
        std::vector<uint8_t> hr;
        if (!rc) { try { hr.resize(padded * c->world); } catch (const std::bad_alloc &) { rc = cerr_(c, GHIP_ENOMEM, "out of host memory for a host collective"); } }
        if (!rc && ghip_copy_to_device(ctx, ds.p, send, bytes) != GHIP_OK) rc = cerr_(c, GHIP_EHIP, "host collective: upload failed");
        if ((rc = agree(c, rc, "the buffers of a host collective"))) return rc;
        if ((rc = allgather_device(c, ds.p, dr.p, padded))) return rc;
        if (ghip_copy_to_host(ctx, hr.data(), dr.p, hr.size()) != GHIP_OK) return cerr_(c, GHIP_EHIP, "host collective: download failed");
        for (uint32_t r = 0; r < c->world; r++) memcpy((char *)recv + (size_t)r * bytes, hr.data() + (size_t)r * padded, bytes);
        return GHIP_OK;
    }
    }
    return GHIP_EINVAL;
}

// variable-length host all-gather: sizes first (unless every rank already knows them: known_sizes), then blocks padded
// to the largest
int allgatherv_host(ghip_comm *c, const void *send, size_t bytes, std::vector<uint8_t> &out, std::vector<uint64_t> &sizes,
                    bool known_sizes = false) {
    int rc = GHIP_OK;
    if (!known_sizes) {
        sizes.assign(c->world, 0);
        const uint64_t mine = bytes;
        if ((rc = allgather_host(c, &mine, sizeof(mine), sizes.data()))) return rc;
    } else if (sizes.size() != c->world || sizes[c->rank] != bytes) return cerr_(c, GHIP_EINVAL, "variable-length gather: inconsistent sizes");
    const uint64_t m = *std::max_element(sizes.begin(), sizes.end());
    uint64_t total = 0;
    for (uint64_t s : sizes) total += s;
    out.resize(total);
    if (m == 0) return GHIP_OK;
    std::vector<uint8_t> sp(m, 0), rp(m * c->world);
    if (bytes) memcpy(sp.data(), send, bytes);
    if ((rc = allgather_host(c, sp.data(), m, rp.data()))) return rc;
    uint64_t at = 0;
    for (uint32_t r = 0; r < c->world; r++) { if (sizes[r]) memcpy(out.data() + at, rp.data() + (size_t)r * m, sizes[r]); at += sizes[r]; }
    return GHIP_OK;
}

// The settings that decide WHICH collectives a call issues and with what sizes: pair_form / join_ranks choose between the
// hash-sharded join (two exchanges) and ghip_precluster_ranks (none); lazy_flush_below shapes a lazy round's request list and
// with it the sizes every rank derives for the round's gather.  They are per context (ghip_options), so nothing but a
// collective can show that the ranks hold the same: the word travels with every status word.
uint32_t settings_word(const ghip_comm *c) {
    const ghip_options o = c->ctx ? c->ctx->opt : ghip_process_options();
    uint32_t h = 2166136261u;
    for (uint32_t v : {o.pair_form, o.join_ranks, o.lazy_flush_below}) { h ^= v; h *= 16777619u; }
    return h;
}

// Phase boundary: every rank contributes its status (GHIP_OK or the code of what failed on it) and its settings word; all
// return together -- GHIP_OK when every rank is fine, its own code on a rank that failed, GHIP_EPEER on the others,
// GHIP_EINVAL on every rank when the settings words differ.  A rank that fails between two collectives therefore never
// leaves its peers waiting in the next one: they meet it here first.  (One small host all-gather: ~0.05 ms over RCCL.)
int agree(ghip_comm *c, int status, const char *what) {
    if (c->world == 1) return status;
    if (c->transport == T_LOCAL && status) {
        // one process, a thread per rank: the group's flag IS the status word -- the peers' barrier returns with it; this rank
        // keeps its own code and message (a failed group stays failed: it is made per call, ghip_cluster_files_multi)
        c->group->fail();
        c->agreed_failure = true;
        return status;
    }
    struct Word { int32_t status; uint32_t settings; };
    std::vector<Word> all(c->world, Word{0, 0});
    const Word mine{status, settings_word(c)};
    c->arriving = true;
    const int rc = allgather_host(c, &mine, sizeof(mine), all.data());
    c->arriving = false;
    if (rc) {
        if (c->transport == T_LOCAL && c->group->failed.load()) {
            c->agreed_failure = true;
            return cerr_(c, GHIP_EPEER, std::string("a peer rank failed at: ") + what);
        }
        return rc;
    }
    if (status) { c->agreed_failure = true; return status; }
    for (uint32_t r = 0; r < c->world; r++)
        if (all[r].status) return c->agreed_failure = true, cerr_(c, GHIP_EPEER, std::string("rank ") + std::to_string(r) + " failed at: " + what + " (code " + std::to_string(all[r].status) + ")");
    for (uint32_t r = 1; r < c->world; r++)
        if (all[r].settings != all[0].settings) {
            c->agreed_failure = true;
            // (cerr_ fails a LOCAL group: the group is made per call, and every rank of it returns this error)
            return cerr_(c, GHIP_EINVAL, std::string("the ranks hold different ghip_options (pair_form / join_ranks / lazy_flush_below decide the collectives "
                                                     "of a call and must be the same on every rank): rank ") + std::to_string(r) + " differs from rank 0, seen at: " + what);
        }
    return GHIP_OK;
}

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

int ghip_comm_gatherv_known(ghip_comm *c, const void *send, size_t bytes, std::vector<uint64_t> &sizes, std::vector<uint8_t> &out) {
    return allgatherv_host(c, send, bytes, out, sizes, true);
}
bool ghip_comm_fault(const ghip_comm *c, uint32_t stage) { return fault_here(c, stage); }
int ghip_comm_note_error(ghip_comm *c, int rc) {
    if (rc && c) c->err = ghip_last_error(c->ctx);   // (ctx == NULL: a host-payload communicator -- the thread's context-less error text)
    return rc;
}

// ================================================================================================ C ABI
extern "C" int ghip_comm_agree(ghip_comm *c, int status) {
    if (!c) return GHIP_EINVAL;
    return agree(c, status, "a phase of the host's own");
}

extern "C" int ghip_comm_unique_id(uint8_t id[GHIP_UNIQUE_ID_BYTES]) {
    if (!id) return GHIP_EINVAL;
    static_assert(GHIP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!rccl().err.empty()) return ghip_set_error(nullptr, GHIP_EUNSUPPORTED, rccl().err);
    ncclUniqueId u;
    ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) return ghip_set_error(nullptr, GHIP_EHIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
    memcpy(id, u.internal, GHIP_UNIQUE_ID_BYTES);
    return GHIP_OK;
}

extern "C" int ghip_comm_init_rank(ghip_ctx *ctx, uint32_t rank, uint32_t world, const uint8_t id[GHIP_UNIQUE_ID_BYTES], ghip_comm **out) {
    if (!ctx || !out || world == 0 || rank >= world || !id) return GHIP_EINVAL;
    *out = nullptr;
    if (!rccl().err.empty()) return ghip_set_error(ctx, GHIP_EUNSUPPORTED, rccl().err);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, GHIP_UNIQUE_ID_BYTES);
    ncclComm_t nc = nullptr;
    ncclResult_t r = rccl().CommInitRank(&nc, (int)world, u, (int)rank);
    if (r != ncclSuccess) return ghip_set_error(ctx, GHIP_EHIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    ghip_comm *c = new ghip_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->transport = T_RCCL; c->nccl = nc;
    // (failures here are local and precede every collective of this communicator: the caller's set-up protocol -- a first probe
    // all-gather under a watchdog, galah_amd/distributed.py -- is where the ranks find out)
    const size_t hc = ghip_comm::HC_BYTES;
    if (hipEventCreateWithFlags(&c->wait_ev, hipEventDisableTiming) != hipSuccess || hipMalloc(&c->hc_send, hc) != hipSuccess ||
        hipMalloc(&c->hc_recv, hc * world) != hipSuccess || hipHostMalloc(&c->hc_pin, hc * world, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ghip_comm_destroy(c);
        return ghip_set_error(ctx, GHIP_ENOMEM, "RCCL communicator: no memory for its status-word buffers");
    }
    *out = c;
    return GHIP_OK;
}

extern "C" int ghip_comm_init_local(ghip_ctx *const *ctxs, uint32_t world, ghip_comm **out_comms) {
    if (!ctxs || !out_comms || world == 0) return GHIP_EINVAL;
    for (uint32_t r = 0; r < world; r++) if (!ctxs[r]) return GHIP_EINVAL;
    LocalGroup *g = new LocalGroup();
    g->world = world;
    g->slot.assign(world, nullptr);
    g->offer.assign(world, std::vector<uint64_t>(world + 1, 0));
    g->device.resize(world);
    for (uint32_t r = 0; r < world; r++) g->device[r] = ctxs[r]->device;
    for (uint32_t r = 0; r < world; r++) {   // direct xGMI reads of the peers' HBM (best effort: staged copies work without it)
        if (hipSetDevice(ctxs[r]->device) != hipSuccess) continue;
        for (uint32_t q = 0; q < world; q++)
            if (g->device[q] != g->device[r]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, g->device[r], g->device[q]) == hipSuccess && can) {
                    hipError_t e = hipDeviceEnablePeerAccess(g->device[q], 0);
                    if (e != hipSuccess) (void)hipGetLastError();   // already enabled
                }
            }
    }
    for (uint32_t r = 0; r < world; r++) {
        ghip_comm *c = new ghip_comm();
        c->ctx = ctxs[r]; c->rank = r; c->world = world; c->transport = world == 1 ? T_SELF : T_LOCAL; c->group = g;
        g->refs++;
        out_comms[r] = c;
    }
    return GHIP_OK;
}

extern "C" int ghip_comm_init_callback(ghip_ctx *ctx, uint32_t rank, uint32_t world, ghip_allgather_fn fn, void *user, ghip_comm **out) {
    if (!out || world == 0 || rank >= world || (world > 1 && !fn)) return GHIP_EINVAL;
    ghip_comm *c = new ghip_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->transport = world == 1 ? T_SELF : T_CALLBACK; c->fn = fn; c->user = user;
    *out = c;
    return GHIP_OK;
}

extern "C" void ghip_comm_destroy(ghip_comm *c) {
    if (!c) return;
    if (c->nccl) rccl().CommDestroy(c->nccl);
    if (c->pin_send) hipHostFree(c->pin_send);
    if (c->pin_recv) hipHostFree(c->pin_recv);
    if (c->wait_ev) hipEventDestroy(c->wait_ev);
    if (c->hc_send) hipFree(c->hc_send);
    if (c->hc_recv) hipFree(c->hc_recv);
    if (c->hc_pin) hipHostFree(c->hc_pin);
    if (c->group && --c->group->refs == 0) delete c->group;
    delete c;
}

extern "C" ghip_ctx *ghip_comm_context(const ghip_comm *c) { return c ? c->ctx : nullptr; }
extern "C" uint32_t ghip_comm_rank(const ghip_comm *c) { return c ? c->rank : 0; }
extern "C" uint32_t ghip_comm_world(const ghip_comm *c) { return c ? c->world : 0; }
extern "C" const char *ghip_comm_transport(const ghip_comm *c) {
    if (!c) return "";
    switch (c->transport) { case T_SELF: return "self"; case T_RCCL: return "rccl"; case T_LOCAL: return "local-peer-copy"; case T_CALLBACK: return "host-callback"; }
    return "";
}
extern "C" const char *ghip_comm_last_error(const ghip_comm *c) { return c ? c->err.c_str() : ""; }

extern "C" int ghip_comm_allgather_device(ghip_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank) {
    if (!c || (bytes_per_rank && (!d_send || !d_recv))) return GHIP_EINVAL;
    return allgather_device(c, d_send, d_recv, bytes_per_rank);
}

extern "C" int ghip_comm_exchange_device(ghip_comm *c, const void *d_send, const uint64_t *send_off, void *d_recv, const uint64_t *recv_off) {
    if (!c || !send_off || !recv_off) return GHIP_EINVAL;
    if ((send_off[c->world] > send_off[0] && !d_send) || (recv_off[c->world] > recv_off[0] && !d_recv)) return GHIP_EINVAL;
    for (uint32_t r = 0; r < c->world; r++) if (send_off[r + 1] < send_off[r] || recv_off[r + 1] < recv_off[r]) return GHIP_EINVAL;
    return exchange_device(c, d_send, send_off, d_recv, recv_off);
}

extern "C" int ghip_comm_allgather_host(ghip_comm *c, const void *send, size_t bytes_per_rank, void *recv) {
    if (!c || (bytes_per_rank && (!send || !recv))) return GHIP_EINVAL;
    return allgather_host(c, send, bytes_per_rank, recv);
}

// ------------------------------------------------------------------------------------------------ sharding rules
extern "C" void ghip_shard_range(size_t n_total, uint32_t rank, uint32_t world, size_t *first, size_t *count, size_t *block) {
    const size_t b = world ? (n_total + world - 1) / world : n_total;
    const size_t f = std::min(n_total, (size_t)rank * b);
    if (first) *first = f;
    if (count) *count = std::min(b, n_total - f);
    if (block) *block = b;
}

// ------------------------------------------------------------------------------------------------ sketch matrix
// The one bulk exchange of the path: N*s*8 bytes, once.  Every rank contributes `block` rows (its count
// rows, the rest padding), every rank ends with the matrix of all n_total genomes.
extern "C" int ghip_allgather_sketches(ghip_comm *c, const ghip_sketches *local, size_t n_total, ghip_sketches **out_full) {
    if (!c || !local || !out_full) return GHIP_EINVAL;
    ghip_ctx *ctx = c->ctx;
    if (!ctx) return GHIP_EINVAL;
    size_t first, count, block;
    ghip_shard_range(n_total, c->rank, c->world, &first, &count, &block);
    if (local->n != count) return cerr_(c, GHIP_EINVAL, "local sketch count does not match this rank's block");
    const uint32_t s = local->s;
    const size_t rows = block * c->world;
    PoolBuf sendh(ctx, block * (size_t)s * 8), sendl(ctx, block * 4), recvh(ctx, rows * (size_t)s * 8), recvl(ctx, rows * 4);
    // everything that can fail on this rank alone happens before the status word; the two collectives follow it
    int rc = GHIP_OK;
    if (!sendh.p || !sendl.p || !recvh.p || !recvl.p) rc = cerr_(c, GHIP_EHIP, "out of device memory for the sketch all-gather");
    if (!rc && hipSetDevice(ctx->device) != hipSuccess) rc = cerr_(c, GHIP_EHIP, "hipSetDevice failed");
    if (!rc && count < block) {  // padding rows: empty sketches (never read: the matrix is cut at n_total)
        if (hipMemsetAsync((char *)sendh.p + count * (size_t)s * 8, 0xff, (block - count) * (size_t)s * 8, ctx->stream) != hipSuccess ||
            hipMemsetAsync((char *)sendl.p + count * 4, 0, (block - count) * 4, ctx->stream) != hipSuccess)
            rc = cerr_(c, GHIP_EHIP, "sketch all-gather: hipMemsetAsync failed");
    }
    if (!rc && count && (rc = ghip_sketches_copy_into(ctx, local, sendh.p, sendl.p))) rc = cerr_(c, rc, std::string("sketch all-gather: ") + ghip_last_error(ctx));
    if ((rc = agree(c, rc, "packing the sketch block"))) return rc;
    if ((rc = allgather_device(c, sendh.p, recvh.p, block * (size_t)s * 8))) return rc;
    if ((rc = allgather_device(c, sendl.p, recvl.p, block * 4))) return rc;
    ghip_sketches *full = nullptr;
    if ((rc = ghip_sketches_wrap_device(ctx, recvh.p, recvl.p, n_total, s, local->k, &full))) return rc;
    full->owned = true;   // the handle now owns the two gather buffers (pool blocks)
    recvh.release(); recvl.release();
    *out_full = full;
    return GHIP_OK;
}

// ------------------------------------------------------------------------------------------------ candidate lists
// Every rank holds a share sorted by (i, j); all end with the whole list in (i, j) order (k-way merge of the runs).
extern "C" int ghip_allgather_pairs(ghip_comm *c, const ghip_pair *local, size_t n_local, ghip_pair **out_all, size_t *out_n) {
    if (!c || !out_all || !out_n || (n_local && !local)) return GHIP_EINVAL;
    std::vector<uint8_t> all;
    std::vector<uint64_t> sizes;
    int rc = allgatherv_host(c, local, n_local * sizeof(ghip_pair), all, sizes);
    if (rc) return rc;
    const size_t n = all.size() / sizeof(ghip_pair);
    ghip_pair *res = (ghip_pair *)malloc(std::max<size_t>(n, 1) * sizeof(ghip_pair));
    if (!res) return GHIP_ENOMEM;
    const ghip_pair *src = reinterpret_cast<const ghip_pair *>(all.data());
    // merge the world sorted runs pairwise (log2(world) passes over the list)
    std::vector<size_t> bounds{0};
    for (uint64_t sz : sizes) bounds.push_back(bounds.back() + sz / sizeof(ghip_pair));
    std::vector<ghip_pair> a(src, src + n), b(n);
    auto key = [](const ghip_pair &p) { return ((uint64_t)p.i << 32) | p.j; };
    while (bounds.size() > 2) {
        std::vector<size_t> nb{0};
        for (size_t r = 0; r + 1 < bounds.size(); r += 2) {
            const size_t lo = bounds[r], mid = bounds[r + 1], hi = r + 2 < bounds.size() ? bounds[r + 2] : mid;
            std::merge(a.begin() + lo, a.begin() + mid, a.begin() + mid, a.begin() + hi, b.begin() + lo,
                       [&](const ghip_pair &x, const ghip_pair &y) { return key(x) < key(y); });
            nb.push_back(hi);
        }
        a.swap(b);
        bounds.swap(nb);
    }
    if (n) memcpy(res, a.data(), n * sizeof(ghip_pair));
    *out_all = res; *out_n = n;
    return GHIP_OK;
}

// ------------------------------------------------------------------------------------------------ pair stage over the ranks
// This rank's share of the precluster pair list of the gathered matrix `sk` (identical on every rank), sorted by (i, j);
// the shares of all ranks partition the list.  From 1 200 genomes the inverted-index form runs HASH-SHARDED (pairs_join.hip:
// ghip_pairs_join_partials / _finish): every rank partitions a 1/world share of the hashes, the per-pair partial counts
// are exchanged (one small host all-gather of sizes and decline flags, one device all-gather of 16 bytes per sharing
// pair and rank), every rank finishes the pairs with (i + j) % world == rank.  Where that form does not apply -- or
// declines on any rank: all ranks learn it from the first collective and decide alike -- ghip_precluster_ranks delivers the
// share (dense tiles, or the record-sharded join whose element stage is replicated; GHIP_JOIN_RANKS=records forces it).
extern "C" int ghip_precluster_comm(ghip_comm *c, const ghip_sketches *sk, float min_ani, ghip_pair **out_pairs, size_t *out_n,
                                    int *out_replicated) {
    if (!c || !sk || !out_pairs || !out_n || !out_replicated) return GHIP_EINVAL;
    ghip_ctx *ctx = c->ctx;
    if (!ctx) return GHIP_EINVAL;
    *out_pairs = nullptr; *out_n = 0; *out_replicated = 0;
    c->agreed_failure = false;
    const size_t n = sk->n;
    const uint32_t s = sk->s, world = c->world, rank = c->rank;
    // pair_form / join_ranks are per context and decide below whether this rank enters exchange 1 at all: the ranks meet
    // first and compare them (GHIP_EINVAL on every rank when they differ)
    if (world > 1) { const int rc0 = agree(c, GHIP_OK, "the pair stage's settings"); if (rc0) return rc0; }
    const uint32_t force = ctx->opt.pair_form, mode = ctx->opt.join_ranks;
    const bool want = world > 1 && mode == GHIP_JOIN_HASH &&
                      (force != GHIP_PAIR_AUTO ? force == GHIP_PAIR_JOIN : (n >= GHIP_JOIN_MIN_N || s > 4096)) && (uint64_t)n * s < (1ull << 32) && n >= 2;
    if (!want) return ghip_precluster_ranks(ctx, sk, min_ani, rank, world, out_pairs, out_n, out_replicated);

    const bool dbg = ghip_dbg(ctx->opt, GHIP_DEBUG_COMM);
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) { fprintf(stderr, "[precluster_comm rank %u] %s %.3f ms\n", rank, what, ms_since(t0)); t0 = std::chrono::steady_clock::now(); } };
    // ---- stage 1: my share of the hashes -> partial common per sharing pair
    void *d_ent = nullptr;
    uint32_t n_ent = 0, status = 1, floor = 0;
    unsigned long long rec_total = 0;
    int rc = GHIP_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (hipSetDevice(ctx->device) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "hipSetDevice failed");   // (a flag of exchange 1, like every error of stage 1)
        if (!rc) rc = ghip_pair_filter_prepare(ctx, s, sk->k, min_ani);
        floor = ctx->cmin.floor;
        // (min_ani <= 0: pairs without a common hash qualify -- the dense forms; the same verdict on every rank)
        if (!rc && floor != 0 && floor != 0xffffu) rc = ghip_pairs_join_partials(ctx, sk->d_hashes, sk->d_lens, n, s, rank, world, &d_ent, &n_ent, &status, &rec_total);
        if (!rc && fault_here(c, GHIP_FAULT_PAIRS_STAGE1)) rc = ghip_set_error(ctx, GHIP_EHIP, "injected fault: pair stage 1");
    }
    struct Freer { ghip_ctx *ctx; void *&p; ~Freer() { if (p) { std::lock_guard<std::mutex> lk(ctx->mu); hipStreamSynchronize(ctx->stream); ghip_pool_free(ctx, p); } } } freer{ctx, d_ent};
    lap("stage 1 (partials)");
    // ---- exchange 1: who declined, how many records, how many entries (an error is a decline flag too: no rank waits alone)
    uint64_t mine3[3] = {(uint64_t)(rc ? 3 : status), rec_total, n_ent};
    std::vector<uint64_t> all3(3 * (size_t)world);
    const int rc_x = allgather_host(c, mine3, sizeof(mine3), all3.data());
    if (rc_x) return rc_x;
    if (rc) { c->agreed_failure = true; return ghip_comm_note_error(c, rc); }   // (the peers read flag 3 in exchange 1: they return too)
    uint64_t any = 0, records = 0, m = 0;
    for (uint32_t r = 0; r < world; r++) { any |= all3[3 * r]; records += all3[3 * r + 1]; m = std::max<uint64_t>(m, all3[3 * r + 2]); }
    const uint64_t P = (uint64_t)n * (n - 1) / 2;
    // a dense pass costs ~1 ns per pair, a record ~0.3 ns: beyond 4 records per pair the dense kernel is the better tool
    if (any || records > 4 * P + (1u << 20) || records >= (1ull << 31) || m * world >= (1ull << 31))
        return (any & 2) ? (c->agreed_failure = true, cerr_(c, GHIP_EPEER, "a peer rank failed in the pair stage (stage 1)")) : ghip_precluster_ranks(ctx, sk, min_ani, rank, world, out_pairs, out_n, out_replicated);
    lap("exchange 1");
    // ---- exchange 2: the entries of every rank (blocks padded to the longest with key = 2^64 - 1).  Its buffers can only be
    // sized now; a rank that cannot get them says so in a status word before anyone enters the device collective.
    const size_t block = std::max<uint64_t>(m, 1) * GHIP_JOIN_ENTRY_BYTES;
    PoolBuf sb(ctx, block), rb(ctx, block * world);
    if (!sb.p || !rb.p) rc = cerr_(c, GHIP_EHIP, "out of device memory for the partial-count exchange");
    if (!rc && (hipMemsetAsync(sb.p, 0xff, block, ctx->stream) != hipSuccess ||
                (n_ent && hipMemcpyAsync(sb.p, d_ent, (size_t)n_ent * GHIP_JOIN_ENTRY_BYTES, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)))
        rc = cerr_(c, GHIP_EHIP, "partial-count exchange: staging failed");
    if ((rc = agree(c, rc, "staging the partial-count exchange"))) return rc;
    if ((rc = allgather_device(c, sb.p, rb.p, block))) return rc;
    lap("exchange 2");
    // ---- stage 2: the pairs I own
    std::vector<ghip_pair> host;
    bool ok = false;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        COMM_HIP(c, hipSetDevice(ctx->device));
        const uint32_t n_all = (uint32_t)(std::max<uint64_t>(m, 1) * world);
        const uint64_t cap = std::max<uint64_t>(1u << 16, n_all);   // at most one candidate per owned entry
        unsigned long long *d_count = (unsigned long long *)ghip_pool_alloc(ctx, sizeof(unsigned long long));
        ghip_pair *d_out = (ghip_pair *)ghip_pool_alloc(ctx, cap * sizeof(ghip_pair));
        if (!d_count || !d_out) { ghip_pool_free(ctx, d_count); ghip_pool_free(ctx, d_out); return cerr_(c, GHIP_EHIP, "out of device memory for the candidate list"); }
        unsigned long long cnt = 0;
        hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream);
        lap("stage 2 set-up");
        if (e == hipSuccess) rc = ghip_pairs_join_finish(ctx, rb.p, n_all, n_all, sk->d_hashes, sk->d_lens, s, ctx->cmin.d_cmin, floor, rank, world, d_out, d_count, cap, &ok);
        if (e == hipSuccess && !rc && fault_here(c, GHIP_FAULT_PAIRS_STAGE2)) rc = ghip_set_error(ctx, GHIP_EHIP, "injected fault: pair stage 2");
        lap("stage 2 kernels");
        if (e == hipSuccess && !rc && ok) {
            e = hipMemcpyAsync(&cnt, d_count, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess && cnt > cap) ok = false;   // cannot happen (one candidate per owned pair at most): the dense share then
            if (e == hipSuccess && ok) {
                host.resize(cnt);
                if (cnt && ghip_copy_to_host(ctx, host.data(), d_out, cnt * sizeof(ghip_pair)) != GHIP_OK) e = hipErrorUnknown;
            }
        }
        hipStreamSynchronize(ctx->stream);
        ghip_pool_free(ctx, d_count); ghip_pool_free(ctx, d_out);
        lap("stage 2 copy-out");
        if (e != hipSuccess) return cerr_(c, GHIP_EHIP, std::string("pair stage (sharded join): ") + hipGetErrorString(e));
        if (rc) { if (c->group) c->group->fail(); return rc; }
        if (ok) {   // pairs this rank answered for: (i + j) % world == rank
            uint64_t cmp = 0;
            for (uint64_t i = 0; i + 1 < n; i++) {
                const uint64_t first = i + 1 + ((rank + 2 * (uint64_t)world - (2 * i + 1) % world) % world);
                if (first < n) cmp += (n - 1 - first) / world + 1;
            }
            ctx->last_pairs = cmp;
        }
    }
    lap(ok ? "stage 2 (finish)" : "stage 2 gave up: dense share");
    // my tables overflowed (huge families): the other ranks deliver their shares, I owe exactly mine -- by a dense pass
    if (!ok) return ghip_precluster_dense_share(ctx, sk, min_ani, rank, world, out_pairs, out_n);
    std::lock_guard<std::mutex> lk(ctx->mu);   // (the recheck's table of f64 results lives in the context)
    return ghip_pairs_finalize(ctx, host, sk->k, min_ani, n, false, rank, world, out_pairs, out_n);
}

// ------------------------------------------------------------------------------------------------ ANI index slices
// A pair's ANI is computed on the rank that owns its FIRST genome; that rank needs the index slices of second genomes
// it does not own.  `pairs` is the whole list (identical on every rank).  On return *out_index serves this rank's pairs:
// the local index itself (*out_index == local, nothing was exchanged) or a combined index = local genomes followed by
// the foreign genomes THIS rank's pairs reference, in ascending order; out_local_ids[g] maps global genome ids to
// positions in that index (UINT32_MAX: not present on this rank).
// Transport: every rank derives from `pairs` which genomes each rank wants of each owner.  An owner packs, per
// destination, the slices that destination wants (one copy_runs launch per array of the index) and the all-to-all-v
// above delivers them straight into the combined arrays -- a slice moves only to the ranks whose pairs reference it,
// and neither the combined index nor the traffic into a rank grows with the world size.  (The host-callback transport
// has no point-to-point primitive: there the packed buffers are all-gathered and a rank keeps its parts.)  Per-genome
// metadata (length, seed capacity, seed count: 24 bytes) of every exchanged genome is all-gathered once on the host.
extern "C" int ghip_exchange_ani_index(ghip_comm *c, const ghip_ani_index *local, size_t n_total, const ghip_pair *pairs,
                                       size_t n_pairs, ghip_ani_index **out_index, uint32_t *out_local_ids /* [n_total] */) {
    if (!c || !local || !out_index || !out_local_ids || (n_pairs && !pairs)) return GHIP_EINVAL;
    ghip_ctx *ctx = c->ctx;
    if (!ctx) return GHIP_EINVAL;
    const uint32_t me = c->rank, world = c->world;
    size_t first, count, block;
    ghip_shard_range(n_total, me, world, &first, &count, &block);
    if (local->n != count) return cerr_(c, GHIP_EINVAL, "local ANI index does not match this rank's block");
    int rc = GHIP_OK;
    { std::lock_guard<std::mutex> lk(ctx->mu); if (ghip_index_wait(ctx, local) != GHIP_OK) rc = cerr_(c, GHIP_EHIP, "ANI index kernels failed"); }
    if (!rc && fault_here(c, GHIP_FAULT_INDEX_PACK)) rc = cerr_(c, GHIP_EHIP, "injected fault: ANI index pack");
    for (size_t g = 0; g < n_total; g++) out_local_ids[g] = UINT32_MAX;
    for (size_t g = 0; g < count; g++) out_local_ids[first + g] = (uint32_t)g;
    // second genomes of pairs that span two ranks: wants[g] = the ranks (bit d of word d / 64) whose pairs reference g
    const size_t W = (world + 63) / 64;
    std::vector<uint64_t> wants(n_total * W, 0);
    bool any = false;
    for (size_t x = 0; x < n_pairs; x++) {
        if (pairs[x].i >= n_total || pairs[x].j >= n_total) return cerr_(c, GHIP_EINVAL, "pair index out of range");   // (the list is the same on every rank: all return)
        const size_t d = pairs[x].i / block;
        if (d != pairs[x].j / block) { wants[pairs[x].j * W + d / 64] |= 1ull << (d % 64); any = true; }
    }
    if (world == 1) { if (rc) return rc; *out_index = const_cast<ghip_ani_index *>(local); return GHIP_OK; }
    if (!any) { if ((rc = agree(c, rc, "the ANI index"))) return rc; *out_index = const_cast<ghip_ani_index *>(local); return GHIP_OK; }
    auto wanted_by = [&](size_t g, uint32_t d) { return (wants[g * W + d / 64] >> (d % 64)) & 1; };
    std::vector<uint32_t> needed;   // exchanged genomes, ascending: the same list on every rank
    for (size_t g = 0; g < n_total; g++) {
        bool some = false;
        for (size_t w = 0; w < W; w++) some |= wants[g * W + w] != 0;
        if (some) needed.push_back((uint32_t)g);
    }
    // per-genome metadata of my genomes among them: length, seed capacity, seed count
    std::vector<uint64_t> meta;
    for (uint32_t g : needed)
        if (g >= first && g < first + count) {
            const uint32_t l = (uint32_t)(g - first);
            meta.push_back(local->glen[l]);
            meta.push_back(local->seed_start[l + 1] - local->seed_start[l]);
            meta.push_back(local->seed_count[l]);
        }
    std::vector<uint8_t> meta_all;
    std::vector<uint64_t> meta_sizes;
    if ((rc = agree(c, rc, "the ANI index"))) return rc;
    rc = allgatherv_host(c, meta.data(), meta.size() * 8, meta_all, meta_sizes);
    if (rc) return rc;
    const size_t n_recv = meta_all.size() / 24;
    if (n_recv != needed.size()) return cerr_(c, GHIP_EINVAL, "ranks disagree on the genomes to exchange");
    const uint64_t *rm = reinterpret_cast<const uint64_t *>(meta_all.data());   // rank order = ascending genome order

    // combined metadata: local genomes, then the foreign genomes my pairs reference (ascending, so grouped by owner)
    std::vector<size_t> wanted;   // positions in `needed`
    for (size_t x = 0; x < n_recv; x++) if (wanted_by(needed[x], me)) wanted.push_back(x);
    const size_t nc = count + wanted.size();
    std::vector<uint64_t> glen(nc), cap(nc);
    std::vector<uint32_t> cnt(nc);
    for (size_t g = 0; g < count; g++) { glen[g] = local->glen[g]; cap[g] = local->seed_start[g + 1] - local->seed_start[g]; cnt[g] = local->seed_count[g]; }
    for (size_t w = 0; w < wanted.size(); w++) {
        const size_t x = wanted[w];
        glen[count + w] = rm[3 * x]; cap[count + w] = rm[3 * x + 1]; cnt[count + w] = (uint32_t)rm[3 * x + 2];
        out_local_ids[needed[x]] = (uint32_t)(count + w);
    }
    std::vector<uint64_t> cseed(nc + 1, 0), cchunk(nc + 1, 0);
    for (size_t g = 0; g < nc; g++) {
        cseed[g + 1] = cseed[g] + cap[g];
        cchunk[g + 1] = cchunk[g] + (glen[g] + local->chunk - 1) / local->chunk;
    }
    const uint64_t bins = GHIP_ANI_BIN_COUNT + 1;
    std::vector<uint64_t> lbin(count + 1), cbin(nc + 1);
    for (size_t g = 0; g <= count; g++) lbin[g] = g * bins;
    for (size_t g = 0; g <= nc; g++) cbin[g] = g * bins;

    // the flat arrays of the index (all of 32-bit words): per-genome slot offsets in the local and the combined layout
    struct Field { const uint32_t *src; const std::vector<uint64_t> *lstart; const std::vector<uint64_t> *cstart; uint32_t *dst; };
    Field fields[4] = {
        {local->d_seed_code, &local->seed_start, &cseed, nullptr},
        {local->d_seed_loc, &local->seed_start, &cseed, nullptr},
        {local->d_bin_start, &lbin, &cbin, nullptr},
        {local->d_chunk_total, &local->chunk_start, &cchunk, nullptr},
    };
    std::vector<void *> owned;
    auto drop = [&]() { std::lock_guard<std::mutex> lk(ctx->mu); hipStreamSynchronize(ctx->stream); for (void *p : owned) ghip_pool_free(ctx, p); owned.clear(); };
    // Phase 1 (local work only): the combined arrays, the local parts copied into them, and per array the slices every
    // destination wants packed back to back.  Whatever fails here fails on this rank alone, so it is remembered, not
    // returned: the status word after the phase takes every rank out together.  Phase 2: the four all-to-all-v exchanges.
    struct Packed { void *send = nullptr; std::vector<uint64_t> send_off, recv_off; };
    Packed packed[4];
    std::vector<uint64_t> runs;
    auto fail = [&](int code, const std::string &msg) { if (!rc) rc = cerr_(c, code, msg); };
    if (hipSetDevice(ctx->device) != hipSuccess) fail(GHIP_EHIP, "hipSetDevice failed");
    for (int fi = 0; fi < 4 && !rc; fi++) {
        Field &f = fields[fi];
        Packed &pk = packed[fi];
        pk.send_off.assign(world + 1, 0); pk.recv_off.assign(world + 1, 0);
        PoolBuf dst(ctx, std::max<uint64_t>((*f.cstart)[nc], 1) * 4);
        if (!dst.p) { fail(GHIP_EHIP, "out of device memory for the combined ANI index"); break; }
        f.dst = static_cast<uint32_t *>(dst.release());
        owned.push_back(f.dst);
        // local part: one copy
        const uint64_t lbytes = (*f.lstart)[count] * 4;
        hipError_t e = lbytes ? hipMemcpyAsync(f.dst, f.src, lbytes, hipMemcpyDeviceToDevice, ctx->stream) : hipSuccess;
        if (e != hipSuccess) { fail(GHIP_EHIP, std::string("ANI index exchange: ") + hipGetErrorString(e)); break; }
        // what I send: per destination, the slices of my genomes it wants, packed back to back (consecutive genomes are
        // one run: they are neighbours in the index arrays)
        runs.clear();
        uint64_t at = 0;
        for (uint32_t d = 0; d < world; d++) {
            pk.send_off[d] = at * 4;
            if (d == me) continue;
            uint64_t prev_end = ~0ull;
            for (size_t l = 0; l < count; l++) {
                if (!wanted_by(first + l, d)) continue;
                const uint64_t a = (*f.lstart)[l], n = (*f.lstart)[l + 1] - a;
                if (n == 0) continue;
                if (a == prev_end) runs.back() += n;
                else { runs.push_back(a); runs.push_back(at); runs.push_back(n); }
                prev_end = a + n;
                at += n;
            }
        }
        pk.send_off[world] = at * 4;
        // what I receive: my wanted genomes, owner by owner, land behind the local part in the combined layout
        {
            size_t w = 0;
            uint64_t got = 0;
            for (uint32_t r = 0; r < world; r++) {
                pk.recv_off[r] = got * 4;
                for (; w < wanted.size() && needed[wanted[w]] / block == r; w++) got += (*f.cstart)[count + w + 1] - (*f.cstart)[count + w];
            }
            pk.recv_off[world] = got * 4;
            if (w != wanted.size() || got != (*f.cstart)[nc] - (*f.cstart)[count]) { fail(GHIP_EINVAL, "ANI index exchange: sizes do not add up"); break; }
        }
        PoolBuf sb(ctx, std::max<uint64_t>(at, 1) * 4), dr(ctx, std::max<size_t>(runs.size(), 1) * sizeof(uint64_t));
        if (!sb.p || !dr.p) { fail(GHIP_EHIP, "out of device memory for the ANI index exchange"); break; }
        if (!runs.empty()) {
            e = hipMemcpyAsync(dr.p, runs.data(), runs.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // (`runs` is pageable and reused by the next array)
            if (e != hipSuccess) { fail(GHIP_EHIP, std::string("ANI index exchange: ") + hipGetErrorString(e)); break; }
            ghip_launch_copy_runs(ctx->stream, f.src, static_cast<uint32_t *>(sb.p), static_cast<const uint64_t *>(dr.p), runs.size() / 3);
        }
        pk.send = sb.release();
        owned.push_back(pk.send);
        // (dr goes back to the pool here: its destructor waits for the stream, i.e. for copy_runs)
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) fail(GHIP_EHIP, "ANI index exchange: packing failed");
    if ((rc = agree(c, rc, "packing the ANI index slices"))) { drop(); return rc; }
    // the four exchanges are collectives: a rank whose own copy fails inside one still enters the remaining ones (its peers
    // do), remembers the first error, and the status word behind them takes every rank out together
    int first_rc = GHIP_OK;
    for (int fi = 0; fi < 4; fi++) {
        Field &f = fields[fi];
        rc = exchange_device(c, packed[fi].send, packed[fi].send_off.data(), (char *)f.dst + (*f.cstart)[count] * 4, packed[fi].recv_off.data());
        if (rc && (c->agreed_failure || c->dead)) { drop(); return rc; }   // agreed inside (every rank returns) / the communicator is gone
        if (rc && !first_rc) first_rc = rc;
    }
    if ((rc = agree(c, first_rc, "the ANI index exchanges"))) { drop(); return rc; }
    {   // the send buffers go back to the pool (exchange_device returns with the stream idle); the combined arrays stay
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (int fi = 0; fi < 4; fi++) { ghip_pool_free(ctx, packed[fi].send); owned.erase(std::find(owned.begin(), owned.end(), packed[fi].send)); }
    }
    ghip_ani_index *idx = nullptr;
    rc = ghip_ani_index_wrap_device(ctx, nc, local->k, local->c, local->chunk, glen.data(), cap.data(), cnt.data(), fields[0].dst,
                                    fields[1].dst, fields[2].dst, fields[3].dst, &idx);
    if ((rc = agree(c, ghip_comm_note_error(c, rc), "wrapping the combined ANI index"))) {
        if (idx) { idx->owned = false; ghip_ani_index_free(idx); }   // (the handle alone: the arrays are still in `owned`)
        drop();
        return rc;
    }
    idx->owned = true;   // the combined arrays belong to the handle
    *out_index = idx;
    return GHIP_OK;
}

// ------------------------------------------------------------------------------------------------ one whole pass
namespace {
// What both whole-pass entry points share: sketch + seed this rank's block, gather the sketch matrix, the pair stage and its
// gather, the ANI index slices.  A STATUS WORD is agreed at every phase boundary whose phase can fail on one rank alone
// (sketching, the pair stage's second half): no rank walks into the next collective without the others.
struct RankPass {
    ghip_sketches *sk_l = nullptr, *sk = nullptr;
    ghip_ani_index *idx_l = nullptr, *idx = nullptr;
    ghip_pair *all = nullptr;
    size_t n_all = 0, first = 0, count = 0, block = 0;
    std::vector<uint32_t> local_ids;
    bool keep_sketches = false;
    ~RankPass() {
        if (sk && sk != sk_l && !keep_sketches) ghip_sketches_free(sk);
        if (sk_l && !(keep_sketches && sk == sk_l)) ghip_sketches_free(sk_l);
        if (idx && idx != idx_l) ghip_ani_index_free(idx);
        if (idx_l) ghip_ani_index_free(idx_l);
    }
};

int rank_pass_front(ghip_comm *c, const ghip_genomes *local, size_t n_total, uint32_t k, uint32_t s, uint64_t seed, float min_ani,
                    uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, RankPass &p, ghip_rank_times &tm) {
    ghip_ctx *ctx = c->ctx;
    c->agreed_failure = false;
    ghip_shard_range(n_total, c->rank, c->world, &p.first, &p.count, &p.block);
    if (local->n != p.count) return cerr_(c, GHIP_EINVAL, "local genome count does not match this rank's block");
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](double &slot) { slot += ms_since(t0); t0 = std::chrono::steady_clock::now(); };
    int rc = ghip_sketch_and_index(ctx, local, k, s, seed, ani_k, ani_c, ani_chunk, &p.sk_l, &p.idx_l);
    if (!rc && fault_here(c, GHIP_FAULT_SKETCH)) rc = ghip_set_error(ctx, GHIP_EHIP, "injected fault: sketch");
    if (rc && c->group) c->group->fail();
    if ((rc = agree(c, ghip_comm_note_error(c, rc), "sketching"))) return rc;
    lap(tm.sketch_ms);
    if (c->world == 1) p.sk = p.sk_l;
    else if ((rc = ghip_allgather_sketches(c, p.sk_l, n_total, &p.sk))) return rc;
    lap(tm.allgather_sketches_ms);
    int replicated = 0;
    ghip_pair *mine = nullptr;
    size_t n_mine = 0;
    rc = ghip_precluster_comm(c, p.sk, min_ani, &mine, &n_mine, &replicated);
    // (an error the pair stage agreed on inside -- exchange 1, the staging of exchange 2 -- is held by every rank already;
    // anything else is this rank's own and is agreed here)
    if (rc && c->agreed_failure) return rc;
    if (rc && c->group) c->group->fail();
    if ((rc = agree(c, ghip_comm_note_error(c, rc), "the pair stage"))) { ghip_free(mine); return rc; }
    tm.pairs_compared = ghip_last_pairs_compared(ctx);
    lap(tm.pairs_ms);
    if (c->world == 1 || replicated) { p.all = mine; p.n_all = n_mine; }
    else {
        rc = ghip_allgather_pairs(c, mine, n_mine, &p.all, &p.n_all);
        ghip_free(mine);
        if (rc) return rc;
    }
    lap(tm.allgather_pairs_ms);
    p.local_ids.resize(std::max<size_t>(n_total, 1));
    if ((rc = ghip_exchange_ani_index(c, p.idx_l, n_total, p.all, p.n_all, &p.idx, p.local_ids.data()))) { ghip_free(p.all); p.all = nullptr; return rc; }
    lap(tm.exchange_ani_index_ms);
    return GHIP_OK;
}
}  // namespace

// FinchPreclusterer::distances + the batched ClusterDistanceFinder::calculate_ani of one dereplication job, on this
// rank's block of genomes, EVERY precluster pair's ANI (eager); every rank returns the same (pairs, ani).
extern "C" int ghip_distances_and_ani_ranks(ghip_comm *c, const ghip_genomes *local, size_t n_total, uint32_t k, uint32_t s,
                                            uint64_t seed, float min_ani, uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk,
                                            float min_aligned_fraction, ghip_pair **out_pairs, float **out_ani, size_t *out_n,
                                            ghip_sketches **out_sketches /* nullable */, ghip_rank_times *times /* nullable */) {
    if (!c || !local || !out_pairs || !out_ani || !out_n) return GHIP_EINVAL;
    ghip_ctx *ctx = c->ctx;
    if (!ctx) return GHIP_EINVAL;
    *out_pairs = nullptr; *out_ani = nullptr; *out_n = 0;
    if (out_sketches) *out_sketches = nullptr;
    ghip_rank_times tm{};
    RankPass p;
    int rc = rank_pass_front(c, local, n_total, k, s, seed, min_ani, ani_k, ani_c, ani_chunk, p, tm);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](double &slot) { slot += ms_since(t0); t0 = std::chrono::steady_clock::now(); };
    ghip_pair *all = p.all;
    const size_t n_all = p.n_all, block = p.block;
    // my share: the pairs whose first genome I own -- a consecutive run of the (i, j)-sorted list
    size_t lo = 0, hi = 0;
    {
        auto owner = [&](const ghip_pair &q) { return q.i / block; };
        lo = std::partition_point(all, all + n_all, [&](const ghip_pair &q) { return owner(q) < c->rank; }) - all;
        hi = std::partition_point(all, all + n_all, [&](const ghip_pair &q) { return owner(q) <= c->rank; }) - all;
    }
    std::vector<uint32_t> pi(2 * (hi - lo));
    for (size_t x = lo; x < hi; x++) { pi[2 * (x - lo)] = p.local_ids[all[x].i]; pi[2 * (x - lo) + 1] = p.local_ids[all[x].j]; }
    std::vector<float> ani_mine(hi - lo);
    rc = hi > lo ? ghip_ani_pairs(ctx, p.idx, pi.data(), hi - lo, min_aligned_fraction, ani_mine.data(), nullptr) : GHIP_OK;
    if (!rc && fault_here(c, GHIP_FAULT_ANI_ROUND)) rc = ghip_set_error(ctx, GHIP_EHIP, "injected fault: ANI share");
    if (rc && c->group) c->group->fail();
    if ((rc = agree(c, ghip_comm_note_error(c, rc), "the ANI shares"))) { ghip_free(all); return rc; }
    lap(tm.ani_pairs_ms);
    float *ani = (float *)malloc(std::max<size_t>(n_all, 1) * sizeof(float));
    if ((rc = agree(c, ani ? GHIP_OK : GHIP_ENOMEM, "the ANI values' host buffer"))) { ghip_free(all); free(ani); return rc; }
    if (c->world == 1) { if (n_all) memcpy(ani, ani_mine.data(), n_all * sizeof(float)); }
    else {  // the ranks' runs are consecutive in rank order: a variable-length gather IS the whole array
        std::vector<uint8_t> got;
        std::vector<uint64_t> sizes(c->world, 0);   // every rank can count every rank's run from the list: one collective less
        for (size_t x = 0; x < n_all; x++) sizes[all[x].i / block] += sizeof(float);
        if ((rc = allgatherv_host(c, ani_mine.data(), ani_mine.size() * sizeof(float), got, sizes, true))) { ghip_free(all); free(ani); return rc; }
        if (got.size() != n_all * sizeof(float)) { ghip_free(all); free(ani); return cerr_(c, GHIP_EINVAL, "ANI gather: sizes do not add up"); }
        if (n_all) memcpy(ani, got.data(), got.size());
    }
    lap(tm.gather_ani_ms);
    if (out_sketches) { *out_sketches = p.sk; p.keep_sketches = true; }
    *out_pairs = all; *out_ani = ani; *out_n = n_all;
    if (times) *times = tm;
    return GHIP_OK;
}

// clusterer::cluster of one dereplication job over the ranks (src/clusterer.rs:14-152 with the finch preclusterer and the
// batched ANI clusterer): the same front, then the LAZY rounds of the native clusterer with each round's requests dealt
// to the ranks (ghip_cluster_index_comm) -- the algorithm one rank runs (ghip_cluster_index), for every world size.  Every
// rank returns the same clusters (and, when asked, the same pair list).  order (nullable): galah's quality order.
extern "C" int ghip_cluster_ranks(ghip_comm *c, const ghip_genomes *local, size_t n_total, uint32_t k, uint32_t s, uint64_t seed,
                                  float min_ani, uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, float min_aligned_fraction,
                                  const uint32_t *order, float ani_threshold, uint32_t **out_members, uint64_t **out_offsets,
                                  size_t *out_n_clusters, ghip_pair **out_pairs /* nullable */, size_t *out_n_pairs /* nullable */,
                                  ghip_sketches **out_sketches /* nullable */, ghip_cluster_times *times /* nullable */) {
    if (!c || !local || !out_members || !out_offsets || !out_n_clusters) return GHIP_EINVAL;
    ghip_ctx *ctx = c->ctx;
    if (!ctx) return GHIP_EINVAL;
    *out_members = nullptr; *out_offsets = nullptr; *out_n_clusters = 0;
    if (out_pairs) *out_pairs = nullptr;
    if (out_n_pairs) *out_n_pairs = 0;
    if (out_sketches) *out_sketches = nullptr;
    ghip_rank_times tm{};
    RankPass p;
    int rc = rank_pass_front(c, local, n_total, k, s, seed, min_ani, ani_k, ani_c, ani_chunk, p, tm);
    if (rc) return rc;
    uint64_t st[5] = {0, 0, 0, 0, 0};
    const bool identity = p.idx == p.idx_l;   // nothing was exchanged: the local index serves global ids only on one rank
    rc = ghip_cluster_index_comm(c, p.idx, (identity && c->world == 1) ? nullptr : p.local_ids.data(), n_total, p.all, p.n_all, order, ani_threshold,
                                 min_aligned_fraction, out_members, out_offsets, out_n_clusters, st);
    if (rc) { ghip_free(p.all); return ghip_comm_note_error(c, rc); }
    if (times) {
        times->sketch_ms = tm.sketch_ms; times->allgather_sketches_ms = tm.allgather_sketches_ms; times->pairs_ms = tm.pairs_ms;
        times->allgather_pairs_ms = tm.allgather_pairs_ms; times->exchange_ani_index_ms = tm.exchange_ani_index_ms;
        times->ani_rounds_ms = (double)st[2] * 1e-6; times->cluster_host_ms = (double)(st[3] - st[2]) * 1e-6;
        times->pairs_compared = tm.pairs_compared; times->ani_pairs_asked = st[0]; times->ani_pairs_here = st[4]; times->lazy_rounds = st[1];
    }
    if (out_sketches) { *out_sketches = p.sk; p.keep_sketches = true; }
    if (out_pairs) { *out_pairs = p.all; if (out_n_pairs) *out_n_pairs = p.n_all; }
    else { if (out_n_pairs) *out_n_pairs = p.n_all; ghip_free(p.all); }
    return GHIP_OK;
}

// ------------------------------------------------------------------------------------------------ one process, many GPUs
extern "C" int ghip_cluster_files_multi(ghip_ctx *const *ctxs, uint32_t world, const char *const *paths, size_t n, uint32_t k,
                                        uint32_t s, float min_ani, float ani_threshold, float min_af, uint32_t ani_c,
                                        int io_threads, uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters) {
    if (!ctxs || world == 0 || !out_members || !out_offsets || !out_n_clusters || (n && !paths)) return GHIP_EINVAL;
    std::vector<ghip_comm *> comms(world, nullptr);
    int rc = ghip_comm_init_local(ctxs, world, comms.data());
    if (rc) return rc;
    std::vector<int> rcs(world, GHIP_OK);
    // every rank runs the whole pass with the lazy ANI rounds of the native clusterer dealt over the devices
    // (ghip_cluster_ranks: the algorithm one device runs) and ends with the same clusters; rank 0's are handed back
    uint32_t *members0 = nullptr;
    uint64_t *offsets0 = nullptr;
    size_t nc0 = 0;
    auto work = [&](uint32_t r) {
        size_t first, count, block;
        ghip_shard_range(n, r, world, &first, &count, &block);
        ghip_genomes *g = nullptr;
        int e = ghip_genomes_from_files(ctxs[r], paths + first, count, std::max(1, io_threads / (int)world), &g);
        if (e) { comms[r]->group->fail(); rcs[r] = e; return; }
        uint32_t *m = nullptr;
        uint64_t *o = nullptr;
        size_t nc = 0;
        e = ghip_cluster_ranks(comms[r], g, n, k, s, 0, min_ani, 15, ani_c ? ani_c : 125, 20000, min_af, nullptr, ani_threshold, &m, &o, &nc,
                               nullptr, nullptr, nullptr, nullptr);
        ghip_genomes_free(g);
        rcs[r] = e;
        if (e == GHIP_OK && r == 0) { members0 = m; offsets0 = o; nc0 = nc; }
        else { ghip_free(m); ghip_free(o); }
    };
    std::vector<std::thread> pool;
    for (uint32_t r = 1; r < world; r++) pool.emplace_back(work, r);
    work(0);
    for (auto &t : pool) t.join();
    // report the rank that actually failed, not the peers its failure released from a collective
    int culprit = -1;
    for (uint32_t r = 0; r < world && culprit < 0; r++)
        if (rcs[r] && rcs[r] != GHIP_EPEER && !strstr(ghip_last_error(ctxs[r]), "a peer rank failed")) culprit = (int)r;
    for (uint32_t r = 0; r < world && culprit < 0; r++) if (rcs[r]) culprit = (int)r;
    if (culprit >= 0) {
        rc = rcs[culprit];
        const std::string msg = std::string("rank ") + std::to_string(culprit) + ": " + ghip_last_error(ctxs[culprit]);
        ghip_set_error(ctxs[0], rc, msg);
    }
    for (uint32_t r = 0; r < world; r++) ghip_comm_destroy(comms[r]);
    if (rc) { ghip_free(members0); ghip_free(offsets0); return rc; }
    *out_members = members0; *out_offsets = offsets0; *out_n_clusters = nc0;
    return GHIP_OK;
}
