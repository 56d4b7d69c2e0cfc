// TEST INFRASTRUCTURE -- "wavesan": a race detector with the GPU's own ordering rules, for the emulated build (make SAN=wavesan).
//
// Why.  The emulator runs the waves of a workgroup one after the other between two __syncthreads, and x86 orders stores: a
// __syncthreads that is MISSING behind a producer wave (wave 0 writes LDS, wave 1 reads it in the same barrier interval) gives
// the right bytes here in every schedule in which wave 0 happens to run first, and the wrong ones on the GPU whenever wave 1
// gets there earlier.  HIPEMU_ORDER=reverse/<seed> catches such a defect only if the wrong order changes a tested result.
// This detector does not depend on the result: it looks at the ACCESSES.
//
// How.  The kernel files (*.hip) are compiled with clang's ThreadSanitizer instrumentation pass -- which turns every load,
// store, atomic and memcpy into a call -- but NOT linked against ThreadSanitizer's runtime: the callbacks are the functions
// below, and they apply the rules of the machine the code is written for instead of those of C++ threads:
//
//   * inside a workgroup, two accesses to overlapping bytes by DIFFERENT WAVES in the SAME barrier interval (no __syncthreads
//     in between), at least one of them a write, not both atomic, are a race -- in LDS and in global memory alike.  Lanes
//     of one wave run in lockstep and cannot race each other; work-items' own stacks are private and skipped;
//   * between workgroups of one launch (second rule, reported separately as "inter-block"): overlapping accesses by two
//     workgroups, at least one a write, not both atomic, are a race UNLESS the later one is ordered behind the earlier one
//     the way the HSA memory model requires at agent scope: the earlier workgroup executed a release (a __threadfence, or an
//     atomic of release order or stronger) after its access and then an atomic; the later workgroup executed an atomic after
//     that and an acquire (a __threadfence, or an atomic of acquire order or stronger) before its access -- fences and orders
//     of AGENT scope or wider only: a workgroup-scope fence is nothing another CU can observe.  This is what the
//     fused scan's grid barrier and the "last block reduces" idiom do; a relaxed flag without the fences passes on x86 and
//     reads stale lines out of another XCD's L2 on the GPU.  (An approximation built on a global sequence number of atomic
//     operations, not vector clocks: it cannot prove a protocol right, it shows the ones that have no fence at all.)
//
// Shadow state: a direct-mapped table of 8-byte granules (last write, last read, byte masks); a collision only forgets, it
// never invents a report.  Reports are deduplicated by (kernel, kind, the two code addresses) and written to WAVESAN_LOG
// (default stderr) as `library+offset` pairs that scripts/wavesan_symbolize.py turns into file:line.
#include <hipemu_wavesan.h>

#include <dlfcn.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <tuple>

namespace {

using hipemu::wavesan_state;

struct Access {
    uint32_t launch, block, epoch;
    uint8_t wave, mask, atomic, lane;
    uint32_t wave_epoch;
    uint64_t seq;                        // global atomic-operation count when the access was made
    const void *pc;
};
struct alignas(64) Cell {
    std::atomic<uint32_t> lock;
    uint32_t pad;
    uint64_t granule;
    Access w, r[2];   // the last write; the reads of up to two waves in their barrier interval (a third forgets one: never invents)
};
constexpr size_t CELL_BITS = 20, CELLS = size_t(1) << CELL_BITS;
Cell *g_cells = nullptr;
std::once_flag g_once;
std::atomic<bool> g_ready{false};

// per workgroup (by serial number, direct-mapped): what the inter-block rule needs to know about the OTHER workgroup
struct BlockSync {
    std::atomic<uint32_t> block;          // whose record this is
    std::atomic<uint64_t> last_release;   // seq of its last atomic that was preceded by a release since its previous plain write
    std::atomic<uint64_t> last_atomic;
};
constexpr size_t SYNC_SLOTS = 1u << 16;
BlockSync *g_sync = nullptr;
std::atomic<uint64_t> g_seq{1};
// the running workgroup's own side (one OS thread runs one workgroup at a time)
struct Mine {
    uint32_t block = 0;
    uint64_t released_at = 0;   // g_seq when a release (fence / release atomic) was last executed
    uint64_t acquired_at = 0;   // g_seq when an acquire (fence / acquire atomic) was last executed
    uint64_t last_atomic = 0;   // g_seq of this workgroup's last atomic operation
    uint64_t last_plain_write = 0;
};
thread_local Mine t_mine;

std::mutex g_rep_mu;
std::set<std::tuple<std::string, int, const void *, const void *>> g_seen;
std::atomic<uint64_t> g_reports[4];   // distinct reports by kind: 0 intra write-write, 1 intra read/write, 2 inter-block, 3 lanes of one wave
std::atomic<uint64_t> g_checked{0};
FILE *g_log = nullptr;
int g_inter = 1, g_lanes = 0;   // (the lane rule is opt-in: WAVESAN_LANES=1)

void init() {
    g_cells = (Cell *)mmap(nullptr, CELLS * sizeof(Cell), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    g_sync = (BlockSync *)mmap(nullptr, SYNC_SLOTS * sizeof(BlockSync), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_cells == (Cell *)MAP_FAILED || g_sync == (BlockSync *)MAP_FAILED) { fprintf(stderr, "[wavesan] cannot map the shadow tables\n"); abort(); }
    const char *path = getenv("WAVESAN_LOG");
    if (path && *path) {
        char name[4096];
        snprintf(name, sizeof name, "%s.%d", path, (int)getpid());
        g_log = fopen(name, "a");
    }
    if (!g_log) g_log = stderr;
    if (const char *v = getenv("WAVESAN_INTER_BLOCK")) g_inter = atoi(v);
    if (const char *v = getenv("WAVESAN_LANES")) g_lanes = atoi(v);
}

std::string where(const void *pc) {
    Dl_info di;
    char buf[512];
    if (dladdr(pc, &di) && di.dli_fname) snprintf(buf, sizeof buf, "%s+0x%zx", di.dli_fname, (size_t)((const char *)pc - (const char *)di.dli_fbase));
    else snprintf(buf, sizeof buf, "%p", pc);
    return buf;
}

void report(int kind, const char *what, const Access &prev, bool prev_write, const void *pc, bool write, bool atomic, uintptr_t addr) {
    const auto &s = wavesan_state;
    const void *a = prev.pc < pc ? prev.pc : pc, *b = prev.pc < pc ? pc : prev.pc;
    std::lock_guard<std::mutex> lk(g_rep_mu);
    if (!g_seen.emplace(s.kernel, kind, a, b).second) return;
    g_reports[kind]++;
    fprintf(g_log, "WAVESAN %s in %s: %s%s by wave %u lane %u at %s  vs  %s%s by wave %u at %s  (address %p, workgroup %u/%u, barrier interval %u)\n", what,
            s.kernel, atomic ? "atomic " : "", write ? "write" : "read", s.wave, s.lane, where(pc).c_str(), prev.atomic ? "atomic " : "",
            prev_write ? "write" : "read", (unsigned)prev.wave, where(prev.pc).c_str(), (void *)addr, s.block, prev.block, s.epoch);
    fflush(g_log);
}

inline BlockSync &sync_of(uint32_t block) { return g_sync[block & (SYNC_SLOTS - 1)]; }

// is this workgroup's access ordered behind `prev`, an access of another workgroup of the same launch?
inline bool ordered_behind(const Access &prev) {
    BlockSync &o = sync_of(prev.block);
    if (o.block.load(std::memory_order_relaxed) != prev.block) return true;   // (slot taken over: unknown, say nothing)
    const uint64_t rel = o.last_release.load(std::memory_order_relaxed);
    // the other side published after its access; this side synchronised after that, and acquired after synchronising
    return rel > prev.seq && t_mine.last_atomic > rel && t_mine.acquired_at >= t_mine.last_atomic;
}

inline void one(uintptr_t granule, uint8_t mask, bool write, bool atomic, const void *pc) {
    const auto &s = wavesan_state;
    Cell &c = g_cells[(granule * 0x9E3779B97F4A7C15ull) >> (64 - CELL_BITS)];
    while (c.lock.exchange(1, std::memory_order_acquire)) {}
    if (c.granule != granule) { c.granule = granule; c.w = Access{}; c.r[0] = Access{}; c.r[1] = Access{}; }
    const uint64_t seq = g_seq.load(std::memory_order_relaxed);
    const uintptr_t addr = granule << 3;
    const bool shared_memory = addr - s.dyn_lds_lo >= s.dyn_lds_len && addr - s.static_lds_lo >= s.static_lds_len;   // (LDS addresses are reused here, not there)
    // rule 0 (LDS only): the same wave, different lanes, no wave_barrier between -- the lanes run in lockstep, but the COMPILER orders a
    // work-item's own LDS store and load only if they may alias; lds[lane] = x; y = lds[lane ^ 1] may be emitted load-first unless a
    // wavefront fence + wave_barrier (wave_sync()) stands between them
    if (g_lanes && !shared_memory) {
        if (c.w.block == s.block && c.w.epoch == s.epoch && c.w.wave == s.wave && c.w.lane != s.lane && c.w.wave_epoch == s.wave_epoch && (c.w.mask & mask) &&
            !(c.w.atomic && atomic))
            report(3, write ? "missing wave barrier (write after another lane's write)" : "missing wave barrier (read after another lane's write)", c.w, true, pc,
                   write, atomic, addr);
        if (write)
            for (const Access &r : c.r)
                if ((r.mask & mask) && !(r.atomic && atomic) && r.block == s.block && r.epoch == s.epoch && r.wave == s.wave && r.lane != s.lane &&
                    r.wave_epoch == s.wave_epoch)
                    report(3, "missing wave barrier (write after another lane's read)", r, false, pc, write, atomic, addr);
    }
    // rule 1: same workgroup, same barrier interval, different waves
    if (c.w.block == s.block && c.w.epoch == s.epoch && c.w.wave != s.wave && (c.w.mask & mask) && !(c.w.atomic && atomic))
        report(write ? 0 : 1, write ? "missing barrier (write after write)" : "missing barrier (read after write)", c.w, true, pc, write, atomic, addr);
    // rule 2: another workgroup of the same launch, in memory workgroups share
    if (g_inter && shared_memory && c.w.launch == s.launch && c.w.block != s.block && c.w.block && (c.w.mask & mask) && !(c.w.atomic && atomic) &&
        !ordered_behind(c.w))
        report(2, write ? "inter-block (write after write, no release/acquire between)" : "inter-block (read after write, no release/acquire between)", c.w, true,
               pc, write, atomic, addr);
    if (write)
        for (const Access &r : c.r) {
            if (!(r.mask & mask) || (r.atomic && atomic)) continue;
            if (r.block == s.block && r.epoch == s.epoch && r.wave != s.wave) report(1, "missing barrier (write after read)", r, false, pc, write, atomic, addr);
            if (g_inter && shared_memory && r.launch == s.launch && r.block != s.block && r.block && !ordered_behind(r))
                report(2, "inter-block (write after read, no release/acquire between)", r, false, pc, write, atomic, addr);
        }
    if (write) {
        if (c.w.block == s.block && c.w.epoch == s.epoch && c.w.wave == s.wave && (!g_lanes || (c.w.lane == s.lane && c.w.wave_epoch == s.wave_epoch))) { c.w.mask |= mask; c.w.atomic &= (uint8_t)atomic; }
        else c.w = Access{s.launch, s.block, s.epoch, (uint8_t)s.wave, mask, (uint8_t)atomic, (uint8_t)s.lane, s.wave_epoch, seq, pc};
        c.w.seq = seq;
        c.w.pc = pc;
    } else {
        Access *mine = nullptr, *stale = nullptr;
        for (Access &r : c.r) {
            if (r.block == s.block && r.epoch == s.epoch && r.wave == s.wave && (!g_lanes || (r.lane == s.lane && r.wave_epoch == s.wave_epoch))) mine = &r;
            else if (r.block != s.block || r.epoch != s.epoch) stale = &r;
        }
        if (mine) { mine->mask |= mask; mine->atomic &= (uint8_t)atomic; mine->seq = seq; }
        else *(stale ? stale : &c.r[1]) = Access{s.launch, s.block, s.epoch, (uint8_t)s.wave, mask, (uint8_t)atomic, (uint8_t)s.lane, s.wave_epoch, seq, pc};
    }
    c.lock.store(0, std::memory_order_release);
}

inline void enter_block() {
    const auto &s = wavesan_state;
    if (t_mine.block == s.block) return;
    t_mine = Mine{};
    t_mine.block = s.block;
    BlockSync &me = sync_of(s.block);
    me.block.store(s.block, std::memory_order_relaxed);
    me.last_release.store(0, std::memory_order_relaxed);
    me.last_atomic.store(0, std::memory_order_relaxed);
}

inline void access(const volatile void *p, size_t size, bool write, bool atomic, const void *pc) {
    const auto &s = wavesan_state;
    if (!s.in_kernel || s.suppress || size == 0) return;
    const uintptr_t addr = (uintptr_t)p;
    if (addr - s.stack_lo < s.stack_len || addr - s.ctx_lo < s.ctx_len) return;   // a work-item's own stack; its threadIdx etc.
    if (__builtin_expect(!g_ready.load(std::memory_order_acquire), 0)) { std::call_once(g_once, init); g_ready.store(true, std::memory_order_release); }
    enter_block();
    static thread_local uint32_t n_mine = 0;
    if ((++n_mine & 4095u) == 0) g_checked.fetch_add(4096, std::memory_order_relaxed);
    if (write && !atomic) t_mine.last_plain_write = g_seq.load(std::memory_order_relaxed);
    for (uintptr_t a = addr, end = addr + size; a < end;) {
        const uintptr_t g = a >> 3, lo = a & 7, hi = std::min<uintptr_t>(8, end - (g << 3));
        one(g, (uint8_t)(((1u << (hi - lo)) - 1u) << lo), write, atomic, pc);
        a = (g + 1) << 3;
    }
}

// memory orders as the instrumentation passes them: 0 relaxed, 1 consume, 2 acquire, 3 release, 4 acq_rel, 5 seq_cst
// the scope note of the operation that is executing (hip_runtime.h sets it in front of every fence and scoped atomic): consumed here
inline bool took_sub_agent_scope() {
    auto &s = const_cast<hipemu::WaveSanState &>(wavesan_state);
    const bool local = s.sub_agent;
    s.sub_agent = false;
    return local;
}

inline void did_fence(int mo) {
    if (took_sub_agent_scope()) return;   // a workgroup / wavefront fence: nothing another CU can observe
    if (!wavesan_state.in_kernel) return;
    if (!g_ready.load(std::memory_order_acquire)) { std::call_once(g_once, init); g_ready.store(true, std::memory_order_release); }
    enter_block();
    const uint64_t now = g_seq.fetch_add(1, std::memory_order_relaxed) + 1;
    if (mo >= 3) t_mine.released_at = now;
    if (mo == 2 || mo >= 4 || mo == 1) t_mine.acquired_at = now;
}
inline void did_atomic(int mo) {
    if (took_sub_agent_scope()) mo = 0;   // its order holds inside the workgroup only: between workgroups it is a relaxed atomic
    if (!wavesan_state.in_kernel) return;
    if (!g_ready.load(std::memory_order_acquire)) return;   // (the access() before it has made the tables)
    enter_block();
    const uint64_t now = g_seq.fetch_add(1, std::memory_order_relaxed) + 1;
    if (mo >= 3) t_mine.released_at = now;
    t_mine.last_atomic = now;
    if (mo == 2 || mo >= 4 || mo == 1) t_mine.acquired_at = now;
    BlockSync &me = sync_of(t_mine.block);
    me.last_atomic.store(now, std::memory_order_relaxed);
    // a release that covers this workgroup's plain writes so far: the fence (or release order) came after the last of them
    if (t_mine.released_at && t_mine.released_at >= t_mine.last_plain_write) me.last_release.store(now, std::memory_order_relaxed);
}

}  // namespace

#define PC __builtin_return_address(0)
extern "C" {
void __tsan_init() {}
void __tsan_func_entry(void *) {}
void __tsan_func_exit() {}
void __tsan_vptr_update(void **, void *) {}
void __tsan_vptr_read(void **) {}
#define RW(N)                                                                                   \
    void __tsan_read##N(void *p) { access(p, N, false, false, PC); }                            \
    void __tsan_write##N(void *p) { access(p, N, true, false, PC); }                            \
    void __tsan_unaligned_read##N(void *p) { access(p, N, false, false, PC); }                  \
    void __tsan_unaligned_write##N(void *p) { access(p, N, true, false, PC); }                  \
    void __tsan_volatile_read##N(void *p) { access(p, N, false, false, PC); }                   \
    void __tsan_volatile_write##N(void *p) { access(p, N, true, false, PC); }                   \
    void __tsan_read_write##N(void *p) { access(p, N, false, false, PC); access(p, N, true, false, PC); }
RW(1) RW(2) RW(4) RW(8) RW(16)
#undef RW
void __tsan_read_range(void *p, unsigned long n) { access(p, n, false, false, PC); }
void __tsan_write_range(void *p, unsigned long n) { access(p, n, true, false, PC); }
void *__tsan_memcpy(void *d, const void *s, unsigned long n) { access(s, n, false, false, PC); access(d, n, true, false, PC); return memcpy(d, s, n); }
void *__tsan_memmove(void *d, const void *s, unsigned long n) { access(s, n, false, false, PC); access(d, n, true, false, PC); return memmove(d, s, n); }
void *__tsan_memset(void *d, int v, unsigned long n) { access(d, n, true, false, PC); return memset(d, v, n); }

void __tsan_atomic_thread_fence(int mo) { did_fence(mo); __atomic_thread_fence(__ATOMIC_SEQ_CST); }
void __tsan_atomic_signal_fence(int) { __atomic_signal_fence(__ATOMIC_SEQ_CST); }

// the atomics themselves are the host's (always seq_cst here: the emulator's workgroups are OS threads); the ORDER the kernel
// asked for is what the inter-block rule looks at
#define ATOMICS(BITS, T)                                                                                                                  \
    T __tsan_atomic##BITS##_load(const volatile T *p, int mo) { access(p, sizeof(T), false, true, PC); T v = __atomic_load_n(p, __ATOMIC_SEQ_CST); did_atomic(mo); return v; } \
    void __tsan_atomic##BITS##_store(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); __atomic_store_n(p, v, __ATOMIC_SEQ_CST); } \
    T __tsan_atomic##BITS##_exchange(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); } \
    T __tsan_atomic##BITS##_fetch_add(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); } \
    T __tsan_atomic##BITS##_fetch_sub(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); } \
    T __tsan_atomic##BITS##_fetch_and(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); } \
    T __tsan_atomic##BITS##_fetch_or(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); } \
    T __tsan_atomic##BITS##_fetch_xor(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); return __atomic_fetch_xor(p, v, __ATOMIC_SEQ_CST); } \
    T __tsan_atomic##BITS##_fetch_nand(volatile T *p, T v, int mo) { access(p, sizeof(T), true, true, PC); did_atomic(mo); return __atomic_fetch_nand(p, v, __ATOMIC_SEQ_CST); } \
    int __tsan_atomic##BITS##_compare_exchange_strong(volatile T *p, T *c, T v, int mo, int) {                                          \
        access(p, sizeof(T), true, true, PC); did_atomic(mo);                                                                           \
        return __atomic_compare_exchange_n(p, c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);                                         \
    }                                                                                                                                     \
    int __tsan_atomic##BITS##_compare_exchange_weak(volatile T *p, T *c, T v, int mo, int) {                                            \
        access(p, sizeof(T), true, true, PC); did_atomic(mo);                                                                           \
        return __atomic_compare_exchange_n(p, c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);                                         \
    }                                                                                                                                     \
    T __tsan_atomic##BITS##_compare_exchange_val(volatile T *p, T c, T v, int mo, int) {                                                \
        access(p, sizeof(T), true, true, PC); did_atomic(mo);                                                                           \
        __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);                                               \
        return c;                                                                                                                         \
    }
ATOMICS(8, uint8_t) ATOMICS(16, uint16_t) ATOMICS(32, uint32_t) ATOMICS(64, uint64_t)
#undef ATOMICS

// what the detector saw: distinct reports of rule 1 (write/write), rule 1 (read/write), rule 2 (inter-block); accesses checked; rule 0 (lanes)
void hipemu_wavesan_counts(uint64_t out[5]) {
    out[0] = g_reports[0]; out[1] = g_reports[1]; out[2] = g_reports[2]; out[3] = g_checked; out[4] = g_reports[3];
}
}
