// TEST INFRASTRUCTURE -- the wave64 emulator behind tests/emu/include/hip/hip_runtime.h (see the header there).
//
// Execution model
//   work-item   a fibre (own stack, hand-written x86-64 context switch) on the OS thread that runs its workgroup
//   wavefront   64 consecutive work-items; a lane that enters a cross-lane operation waits until every lane of the wave
//               that can still reach it has blocked somewhere; the lanes waiting at ONE call site then exchange values.
//               Two sites waited in at once (divergence): the lower code address goes first (if/loop bodies sit before
//               the code that follows them) and the event is counted (hipemu_stats) -- exited lanes and lanes outside the
//               group read as 0, also counted.
//   workgroup   waves run one after the other up to their next __syncthreads; `__shared__` is `static thread_local`
//               (one OS thread = one workgroup at a time), dynamic LDS is a per-thread buffer poisoned between groups
//   grid        workgroups are dealt to a pool of OS threads (HIPEMU_THREADS, default = cores), so global atomics and
//               fences are the host's; a workgroup that sleeps in a spin loop (grid barrier) makes the launch start one OS
//               thread per remaining workgroup, i.e. all of them co-resident
//   runtime     streams are synchronous (a valid schedule of any stream program), events are host clock readings,
//               device memory is host memory (poisoned with 0xA5 at hipMalloc)
#include <hip/hip_runtime.h>
#include <hipemu_wavesan.h>

#include <link.h>
#include <sys/mman.h>

// The sanitizer builds (make SAN=asan: AddressSanitizer + UBSan): device buffers are heap blocks here, so a store past the
// end of one is a report instead of silent corruption of its neighbour; the fibre switches are announced to the runtime.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#endif
#endif

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl hipemu_switch
.hidden hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

thread_local WaveSanState wavesan_state;
void wavesan_suppress(int delta) { wavesan_state.suppress += (uint32_t)delta; }
void wavesan_scope(bool sub_agent) { wavesan_state.sub_agent = sub_agent; }

namespace {
std::atomic<uint32_t> g_launch_serial{0}, g_block_serial{0};

enum State : uint8_t { READY, WAVE_WAIT, BARRIER, YIELDED, DONE };

struct Launch {
    const char *name;
    const void *fn = nullptr;
    dim3 grid, block;
    size_t lds;
    std::function<void()> body;
    uint64_t total;
    uint32_t serial = 0;
    std::atomic<uint64_t> next{0}, done{0};
    int refs = 0;   // pool workers inside (under the pool's mutex)
    std::mutex extra_mu;
    std::vector<std::thread> extra;
    std::atomic<bool> coresident{false};
};

struct Worker {   // per OS thread
    char *stack_map = nullptr, *stacks = nullptr;
    size_t n_stacks = 0, stack_map_bytes = 0;
    std::vector<ThreadCtx> ctx;
    std::vector<State> st;
    std::vector<void *> sp;
    void *sched_sp = nullptr;
    unsigned char *lds = nullptr;
    size_t lds_cap = 0;
    Launch *cur = nullptr;
    uint32_t cur_thread = 0;
    std::vector<uint32_t> wave_barriers;   // per wave of the running workgroup: wave_barriers passed (the race detector's lane rule)
#ifdef HIPEMU_ASAN
    std::vector<void *> fake;           // each fibre's fake-stack handle while it is switched out
    void *sched_fake = nullptr;
    const void *sched_bottom = nullptr; // the OS thread's own stack, as the runtime reports it at the first switch
    size_t sched_size = 0;
#endif
    ~Worker() {
        if (stack_map) munmap(stack_map, stack_map_bytes);
#ifdef HIPEMU_ASAN
        if (lds) ASAN_UNPOISON_MEMORY_REGION(lds, lds_cap);
#endif
        free(lds);
    }
};
thread_local Worker *tw = nullptr;
thread_local Worker tw_storage;

struct Stats {
    std::atomic<uint64_t> launches{0}, blocks{0}, wave_ops{0}, divergent_ops{0}, inactive_reads{0}, partial_ops{0}, coresident_launches{0};
} g_stats;
std::mutex g_diag_mu;
std::unordered_map<std::string, int> g_diag_seen;
int g_strict = -1;

void diag(const Launch *L, const char *what, const void *site) {
    if (g_strict < 0) g_strict = getenv("HIPEMU_STRICT") ? atoi(getenv("HIPEMU_STRICT")) : 0;
    if (!g_strict) return;
    std::lock_guard<std::mutex> lk(g_diag_mu);
    char key[512];
    snprintf(key, sizeof key, "%s|%s|%p", L->name, what, site);
    if (g_diag_seen[key]++ == 0) fprintf(stderr, "[hipemu] %s: %s at site %p\n", L->name, what, site);
}

size_t env_size(const char *name, size_t dflt) {
    const char *v = getenv(name);
    return v && *v ? (size_t)strtoull(v, nullptr, 10) : dflt;
}

void fiber_entry() {
    Worker *w = static_cast<Worker *>(cur()->worker);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &w->sched_bottom, &w->sched_size);   // first time on this stack
#endif
    w->cur->body();
    w = static_cast<Worker *>(cur()->worker);
    w->st[w->cur_thread] = DONE;
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(nullptr, w->sched_bottom, w->sched_size);   // nullptr: this fibre does not come back
#endif
    hipemu_switch(&w->sp[w->cur_thread], w->sched_sp);
    __builtin_trap();
}

inline void to_sched(ThreadCtx *c, State s) {
    Worker *w = static_cast<Worker *>(c->worker);
    const uint32_t t = c->flat;
    w->st[t] = s;
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(&w->fake[t], w->sched_bottom, w->sched_size);
#endif
    hipemu_switch(&w->sp[t], w->sched_sp);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(w->fake[t], nullptr, nullptr);
#endif
}

inline void run_fiber(Worker *w, uint32_t t) {
    w->cur_thread = t;
    WaveSanState &ws = wavesan_state;
    ws.wave = t >> 6;
    ws.lane = t & 63;
    ws.wave_epoch = w->wave_barriers[t >> 6];
    ws.in_kernel = true;
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(&w->sched_fake, w->stacks + (size_t)t * STACK_BYTES, STACK_BYTES);
#endif
    hipemu_switch(&w->sched_sp, w->sp[t]);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(w->sched_fake, nullptr, nullptr);
#endif
    ws.in_kernel = false;
}

// the lanes of wave [lo, hi) that wait in a cross-lane operation: the group at the lowest site exchanges and becomes READY
void resolve_wave(Worker *w, Launch *L, uint32_t lo, uint32_t hi) {
    const void *site = nullptr;
    bool several = false;
    for (uint32_t t = lo; t < hi; t++)
        if (w->st[t] == WAVE_WAIT) {
            const void *s = w->ctx[t].op_site;
            if (!site) site = s;
            else if (s != site) { several = true; if (s < site) site = s; }
        }
    if (several) { g_stats.divergent_ops++; diag(L, "two cross-lane sites waited in at once (divergent wave)", site); }
    uint64_t mask = 0;
    int kind = 0;
    for (uint32_t t = lo; t < hi; t++)
        if (w->st[t] == WAVE_WAIT && w->ctx[t].op_site == site) { mask |= 1ull << (t - lo); kind = w->ctx[t].op_kind; }
    g_stats.wave_ops++;
    if (kind == OP_WAVE_BARRIER) w->wave_barriers[lo >> 6]++;
    if (__builtin_popcountll(mask) != (int)(hi - lo) || hi - lo != 64) g_stats.partial_ops++;
    uint64_t ballot = 0, first = 0;
    if (kind == OP_BALLOT)
        for (uint32_t t = lo; t < hi; t++)
            if ((mask >> (t - lo)) & 1 && w->ctx[t].op_val) ballot |= 1ull << (t - lo);
    if (kind == OP_FIRST) first = w->ctx[lo + (uint32_t)__builtin_ctzll(mask)].op_val;
    for (uint32_t t = lo; t < hi; t++) {
        if (!((mask >> (t - lo)) & 1)) continue;
        ThreadCtx &c = w->ctx[t];
        const int self = (int)(t - lo), width = c.op_width;
        int src = self;
        switch (kind) {
        case OP_SHFL: src = (c.op_arg & (width - 1)) + (self & ~(width - 1)); break;
        case OP_SHFL_UP: src = self - c.op_arg; if (src < (self & ~(width - 1))) src = self; break;
        case OP_SHFL_DOWN: src = self + c.op_arg; if ((self & (width - 1)) + c.op_arg >= width) src = self; break;
        case OP_SHFL_XOR: src = self ^ c.op_arg; if (src >= ((self + width) & ~(width - 1))) src = self; break;
        default: break;
        }
        switch (kind) {
        case OP_SHFL: case OP_SHFL_UP: case OP_SHFL_DOWN: case OP_SHFL_XOR:
            if (src < 0 || src >= 64 || !((mask >> src) & 1)) {
                c.op_res = 0;
                g_stats.inactive_reads++;
                diag(L, "shuffle reads a lane that is not in the operation (exited, diverged or past the block)", site);
            } else c.op_res = w->ctx[lo + (uint32_t)src].op_val;
            break;
        case OP_BALLOT: c.op_res = ballot; break;
        case OP_FIRST: c.op_res = first; break;
        default: c.op_res = 0; break;
        }
    }
    // results are read from op_res; op_val of a source lane must not change before every reader has its copy: they are
    // all written above, only now do the lanes run again
    for (uint32_t t = lo; t < hi; t++)
        if ((mask >> (t - lo)) & 1) w->st[t] = READY;
}

void start_coresident(Launch *L);

void run_block(Worker *w, Launch *L, uint64_t b) {
    const uint32_t n = L->block.x * L->block.y * L->block.z, nw = (n + 63) / 64;
    const size_t stack_bytes = STACK_BYTES;
    if (w->n_stacks < n) {
        if (w->stack_map) munmap(w->stack_map, w->stack_map_bytes);
        w->stack_map_bytes = (size_t)(n + 1) * stack_bytes;   // one more: the first stack starts at a multiple of its size
        w->stack_map = (char *)mmap(nullptr, w->stack_map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (w->stack_map == (char *)MAP_FAILED) { fprintf(stderr, "[hipemu] cannot map %u fibre stacks\n", n); abort(); }
        w->stacks = (char *)(((uintptr_t)w->stack_map + stack_bytes - 1) & ~(uintptr_t)(stack_bytes - 1));
        w->n_stacks = n;
    }
    if (w->ctx.size() < n) { w->ctx.resize(n); w->st.resize(n); w->sp.resize(n); }
    w->wave_barriers.assign(nw, 0);
#ifdef HIPEMU_ASAN
    if (w->fake.size() < n) w->fake.resize(n);
    if (w->lds) ASAN_UNPOISON_MEMORY_REGION(w->lds, w->lds_cap);
#endif
    if (w->lds_cap < L->lds + 64) {
        free(w->lds);
        w->lds_cap = L->lds + 64;
        if (posix_memalign((void **)&w->lds, 256, w->lds_cap)) abort();
    }
    memset(w->lds, 0xCD, w->lds_cap);   // LDS holds whatever the previous workgroup left: nothing may rely on zeros
#ifdef HIPEMU_ASAN
    ASAN_POISON_MEMORY_REGION(w->lds + L->lds, w->lds_cap - L->lds);   // dynamic LDS ends where the launch said it does
#endif
    w->cur = L;
    {
        WaveSanState &ws = wavesan_state;
        ws.launch = L->serial;
        ws.block = g_block_serial.fetch_add(1) + 1;
        ws.epoch = 0;
        ws.kernel = L->name;
        ws.stack_lo = (uintptr_t)w->stack_map;
        ws.stack_len = w->stack_map_bytes;
        ws.ctx_lo = (uintptr_t)w->ctx.data();
        ws.ctx_len = w->ctx.size() * sizeof(ThreadCtx);
        ws.dyn_lds_lo = (uintptr_t)w->lds;
        ws.dyn_lds_len = w->lds_cap;
        if (!ws.static_lds_len) {   // this thread's block of the thread_local statics of the module the kernels live in
            struct Q { const void *fn; uintptr_t lo, len; } q{(const void *)L->fn, 0, 0};
            dl_iterate_phdr([](dl_phdr_info *info, size_t, void *qp) {
                Q *q = (Q *)qp;
                bool here = false;
                uintptr_t tls_len = 0;
                for (int i = 0; i < info->dlpi_phnum; i++) {
                    const auto &ph = info->dlpi_phdr[i];
                    if (ph.p_type == PT_LOAD && (uintptr_t)q->fn - (info->dlpi_addr + ph.p_vaddr) < ph.p_memsz) here = true;
                    if (ph.p_type == PT_TLS) tls_len = ph.p_memsz;
                }
                if (!here) return 0;
                q->lo = (uintptr_t)info->dlpi_tls_data;
                q->len = q->lo ? tls_len : 0;
                return 1;
            }, &q);
            ws.static_lds_lo = q.lo;
            ws.static_lds_len = q.len ? q.len : 1;   // (1: looked, found none)
        }
    }
    dim3 bid;
    bid.x = (uint32_t)(b % L->grid.x);
    bid.y = (uint32_t)((b / L->grid.x) % L->grid.y);
    bid.z = (uint32_t)(b / ((uint64_t)L->grid.x * L->grid.y));
    for (uint32_t t = 0; t < n; t++) {
        ThreadCtx &c = w->ctx[t];
        c.tid = dim3(t % L->block.x, (t / L->block.x) % L->block.y, t / (L->block.x * L->block.y));
        c.bid = bid;
        c.bdim = L->block;
        c.gdim = L->grid;
        c.lane = t & 63;
        c.wave = t >> 6;
        c.flat = t;
        c.worker = w;
        *reinterpret_cast<ThreadCtx **>(w->stacks + (size_t)t * stack_bytes) = &c;   // what cur() finds from the stack pointer
        w->st[t] = READY;
        uint64_t *top = (uint64_t *)(w->stacks + (size_t)(t + 1) * stack_bytes);   // 16-byte aligned (page aligned)
        top[-1] = 0;                          // the return address fiber_entry would see (never used)
        top[-2] = (uint64_t)&fiber_entry;     // `ret` of the first switch lands here with rsp = top - 8
        for (int i = 3; i <= 8; i++) top[-i] = 0;
        w->sp[t] = top - 8;
#ifdef HIPEMU_ASAN
        // the previous fibre on this stack left through fiber_entry's frame without returning: its redzones are still marked
        ASAN_UNPOISON_MEMORY_REGION((char *)top - 16384, 16384);
        w->fake[t] = nullptr;
#endif
    }
    // HIPEMU_ORDER: the order waves of a workgroup (and lanes of a wave) take their turns in between two meeting points --
    // "forward" (default), "reverse", or a seed for a fresh random order at every turn.  Any order is a schedule the hardware
    // may produce; a result that changes with it is a missing barrier (or a reliance on wave lockstep without wave_barrier).
    static const int order_mode = [] { const char *v = getenv("HIPEMU_ORDER"); return !v || !*v || !strcmp(v, "forward") ? 0 : !strcmp(v, "reverse") ? 1 : 2; }();
    static const uint64_t order_seed = [] { const char *v = getenv("HIPEMU_ORDER"); return v ? strtoull(v, nullptr, 10) : 0ull; }();
    uint64_t rng = order_seed * 0x9E3779B97F4A7C15ull + b * 0xD1B54A32D192ED03ull + 1;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    std::vector<uint32_t> wave_turn(nw), lane_turn(64);
    uint64_t yield_turns = 0;
    for (;;) {
        for (uint32_t x = 0; x < nw; x++) wave_turn[x] = order_mode == 1 ? nw - 1 - x : x;
        if (order_mode == 2) for (uint32_t x = nw; x > 1; x--) std::swap(wave_turn[x - 1], wave_turn[next() % x]);
        for (uint32_t wx = 0; wx < nw; wx++) {
            const uint32_t wv = wave_turn[wx];
            const uint32_t lo = wv * 64, hi = std::min(n, lo + 64), nl = hi - lo;
            for (;;) {
                for (uint32_t x = 0; x < nl; x++) lane_turn[x] = order_mode == 1 ? nl - 1 - x : x;
                if (order_mode == 2) for (uint32_t x = nl; x > 1; x--) std::swap(lane_turn[x - 1], lane_turn[next() % x]);
                for (uint32_t x = 0; x < nl; x++) {
                    const uint32_t t = lo + lane_turn[x];
                    if (w->st[t] == READY) run_fiber(w, t);
                }
                // every lane has run until it blocked: those that wait in a cross-lane operation exchange and run on; when none
                // does, the wave stands at a block barrier, sleeps in a spin loop, or is done
                bool waiting = false;
                for (uint32_t t = lo; t < hi && !waiting; t++) waiting = w->st[t] == WAVE_WAIT;
                if (!waiting) break;
                resolve_wave(w, L, lo, hi);
            }
        }
        uint32_t n_done = 0, n_bar = 0, n_yield = 0;
        for (uint32_t t = 0; t < n; t++) {
            n_done += w->st[t] == DONE;
            n_bar += w->st[t] == BARRIER;
            n_yield += w->st[t] == YIELDED;
        }
        if (n_done == n) break;
        if (n_yield) {   // somebody spins on another workgroup: make sure that one exists, let it run, try again
            start_coresident(L);
            // A poll of a spin loop stands for ~1 us of GPU time: a BOUNDED wait (the fused scan's grid barrier gives up after
            // 2^18 polls) must not run out here just because the host is busy with the other workgroups' fibres -- after the
            // first thousand turns a turn takes at least 20 us of wall time (2^18 polls: five seconds and more).
            if (++yield_turns > 1000) std::this_thread::sleep_for(std::chrono::microseconds(20));
            else std::this_thread::yield();
            for (uint32_t t = 0; t < n; t++)
                if (w->st[t] == YIELDED) w->st[t] = READY;
            continue;
        }
        if (n_bar + n_done != n) { fprintf(stderr, "[hipemu] %s: scheduler found nothing to run\n", L->name); abort(); }
        for (uint32_t t = 0; t < n; t++)   // s_barrier counts the waves that are still alive
            if (w->st[t] == BARRIER) w->st[t] = READY;
        wavesan_state.epoch++;
    }
    g_stats.blocks++;
}

void block_loop(Launch *L) {
    if (!tw) tw = &tw_storage;
    for (;;) {
        const uint64_t b = L->next.fetch_add(1);
        if (b >= L->total) break;
        run_block(tw, L, b);
        L->done.fetch_add(1);
    }
}

void start_coresident(Launch *L) {
    // (called at every turn of a spinning workgroup: the common case must not touch the mutex -- the launcher holds it while it
    // joins the extra threads, and a spinner that blocked on it here would never finish: a deadlock seen once in four runs)
    if (L->coresident.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(L->extra_mu);
    if (L->coresident.load(std::memory_order_relaxed)) return;
    L->coresident.store(true, std::memory_order_release);
    g_stats.coresident_launches++;
    for (;;) {
        const uint64_t b = L->next.fetch_add(1);
        if (b >= L->total) break;
        L->extra.emplace_back([L, b]() {
            tw = &tw_storage;
            run_block(tw, L, b);
            L->done.fetch_add(1);
        });
    }
}

struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Launch *> active;
    std::vector<std::thread> threads;
    bool stop = false, started = false;
    void start() {
        size_t n = env_size("HIPEMU_THREADS", std::max(1u, std::thread::hardware_concurrency()));
        for (size_t i = 1; i < n; i++) threads.emplace_back([this]() { main(); });
        started = true;
    }
    void main() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            Launch *L = nullptr;
            for (Launch *a : active)
                if (a->next.load() < a->total) { L = a; break; }
            if (!L) {
                if (stop) return;
                cv_work.wait(lk);
                continue;
            }
            L->refs++;
            lk.unlock();
            block_loop(L);
            lk.lock();
            L->refs--;
            cv_done.notify_all();
        }
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
};
Pool *g_pool = nullptr;
std::once_flag g_pool_once;

thread_local hipError_t t_last_error = hipSuccess;
std::mutex g_attr_mu;
std::unordered_map<const void *, size_t> g_dyn_lds_allow;
std::mutex g_mem_mu;
std::unordered_map<void *, size_t> g_allocs;
std::atomic<size_t> g_mem_now{0}, g_mem_peak{0};

}  // namespace

void launch_bound_exceeded(const char *kernel, unsigned threads, unsigned bound) {
    fprintf(stderr, "[hipemu] %s launched with %u work-items per workgroup but declares __launch_bounds__(%u): the GPU refuses this launch\n", kernel, threads, bound);
    abort();
}

uint64_t wave_op(int kind, uint64_t val, int arg, int width, const void *site) {
    ThreadCtx *c = cur();
    c->op_kind = kind;
    c->op_val = val;
    c->op_arg = arg;
    c->op_width = width <= 0 || width > 64 ? 64 : width;
    c->op_site = site;
    to_sched(c, WAVE_WAIT);
    return c->op_res;
}
void block_barrier() { to_sched(cur(), BARRIER); }
void yield() { to_sched(cur(), YIELDED); }
unsigned char *dyn_lds() { return static_cast<Worker *>(cur()->worker)->lds; }

void drain_stream(hipStream_t s);
void launch(const char *name, const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t stream, std::function<void()> body) {
    drain_stream(stream);
    const uint64_t threads = (uint64_t)block.x * block.y * block.z, total = (uint64_t)grid.x * grid.y * grid.z;
    size_t allow = 64 * 1024;
    {
        std::lock_guard<std::mutex> lk(g_attr_mu);
        auto it = g_dyn_lds_allow.find(fn);
        if (it != g_dyn_lds_allow.end()) allow = std::max(allow, it->second);
    }
    // what a real launch refuses: empty or oversized grids and blocks, dynamic LDS above the kernel's allowance (64 KiB
    // until hipFuncSetAttribute raises it; 160 KiB is all a gfx950 CU has)
    if (threads == 0 || threads > 1024 || total == 0 || grid.x > 0x7fffffffu || grid.y > 65535u || grid.z > 65535u || lds > allow || lds > 160 * 1024) {
        fprintf(stderr, "[hipemu] launch of %s refused: grid (%u,%u,%u) block (%u,%u,%u) lds %zu (allowance %zu)\n", name, grid.x, grid.y, grid.z,
                block.x, block.y, block.z, lds, allow);
        t_last_error = hipErrorInvalidValue;
        return;
    }
    std::call_once(g_pool_once, []() { g_pool = new Pool(); g_pool->start(); });
    g_stats.launches++;
    Launch L;
    L.name = name;
    L.fn = fn;
    L.grid = grid;
    L.block = block;
    L.lds = lds;
    L.body = std::move(body);
    L.total = total;
    L.serial = g_launch_serial.fetch_add(1) + 1;
    const bool shared = total > 1 && !g_pool->threads.empty();
    if (shared) {
        std::lock_guard<std::mutex> lk(g_pool->mu);
        g_pool->active.push_back(&L);
        g_pool->cv_work.notify_all();
    }
    block_loop(&L);
    if (shared) {
        std::unique_lock<std::mutex> lk(g_pool->mu);
        g_pool->cv_done.wait(lk, [&]() { return L.refs == 0; });
        g_pool->active.erase(std::find(g_pool->active.begin(), g_pool->active.end(), &L));
    }
    {
        std::vector<std::thread> extra;
        {
            std::lock_guard<std::mutex> lk(L.extra_mu);
            L.coresident.store(true, std::memory_order_release);   // (nothing left to start: a late spinner need not look)
            extra.swap(L.extra);
        }
        for (auto &t : extra) t.join();   // outside the lock
    }
    while (L.done.load() < L.total) std::this_thread::yield();
}

}  // namespace hipemu

// ------------------------------------------------------------------ the runtime API
using namespace hipemu;

// Streams run what they are given at once, in the caller's thread -- except for PENDING work: something outside the
// emulator (the stand-in for RCCL, tests/emu/fake_rccl) puts a ticket on a stream that completes later, from another thread
// (the peers of a collective arriving).  Everything put on the stream after a ticket waits for it first; an event recorded
// behind one is not ready until it is; hipEventQuery is the one call that does not wait.
struct Pending { std::atomic<int> state{0}; };   // 0 in flight, 1 done, 2 failed
struct hipemuStream {
    std::mutex mu;
    std::vector<std::shared_ptr<Pending>> pending;
};
struct hipemuEvent {
    double t_ms = 0.0;
    std::vector<std::shared_ptr<Pending>> behind;   // tickets in flight on the stream when the event was recorded
};
static hipemuStream g_null_stream;
static hipemuStream *stream_of(hipStream_t s) { return s ? s : &g_null_stream; }
static bool all_done(const std::vector<std::shared_ptr<Pending>> &v) {
    for (auto &p : v) if (p->state.load(std::memory_order_acquire) == 0) return false;
    return true;
}
// waits for the stream's tickets (what the hardware queue does before it runs the next item); returns false if one failed
static bool drain(hipStream_t s_) {
    hipemuStream *s = stream_of(s_);
    std::vector<std::shared_ptr<Pending>> v;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->pending.empty()) return true;
        v = s->pending;
    }
    for (uint64_t spins = 0; !all_done(v); spins++) {
        if (spins < 200) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    bool ok = true;
    for (auto &p : v) ok = ok && p->state.load() == 1;
    std::lock_guard<std::mutex> lk(s->mu);
    for (auto &p : v) s->pending.erase(std::remove(s->pending.begin(), s->pending.end(), p), s->pending.end());
    return ok;
}

namespace hipemu { void drain_stream(hipStream_t s) { (void)drain(s); } }

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static thread_local int t_device = 0;

extern "C" {

hipError_t hipGetDeviceCount(int *n) { *n = (int)env_size("HIPEMU_DEVICES", 1); return hipSuccess; }
hipError_t hipSetDevice(int d) {
    int n; hipGetDeviceCount(&n);
    if (d < 0 || d >= n) return t_last_error = hipErrorInvalidValue;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "hipemu (wave64 emulator on the host CPU)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950");
    p->totalGlobalMem = (size_t)288 << 30;
    p->sharedMemPerBlock = 64 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    p->multiProcessorCount = (int)env_size("HIPEMU_CUS", 256);
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->clockRate = 2400000;
    p->l2CacheSize = 4 << 20;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { drain(nullptr); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t bytes) {
    *p = nullptr;
    const size_t limit = env_size("HIPEMU_MEM_MB", 48 * 1024) << 20;
    if (g_mem_now.load() + bytes > limit) return t_last_error = hipErrorOutOfMemory;
    void *q = nullptr;
    if (posix_memalign(&q, 256, std::max<size_t>(bytes, 1))) return t_last_error = hipErrorOutOfMemory;
    if (env_size("HIPEMU_POISON", 1)) memset(q, 0xA5, bytes);
    {
        std::lock_guard<std::mutex> lk(g_mem_mu);
        g_allocs[q] = bytes;
    }
    const size_t now = g_mem_now.fetch_add(bytes) + bytes;
    size_t peak = g_mem_peak.load();
    while (now > peak && !g_mem_peak.compare_exchange_weak(peak, now)) {}
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mem_mu);
        auto it = g_allocs.find(p);
        if (it == g_allocs.end()) return t_last_error = hipErrorInvalidValue;
        g_mem_now.fetch_sub(it->second);
        g_allocs.erase(it);
    }
    free(p);
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes) {
    const size_t limit = env_size("HIPEMU_MEM_MB", 48 * 1024) << 20, now = g_mem_now.load();
    *total_bytes = limit;
    *free_bytes = now < limit ? limit - now : 0;
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
    *p = nullptr;
    if (posix_memalign(p, 4096, std::max<size_t>(bytes, 1))) return t_last_error = hipErrorOutOfMemory;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { drain(nullptr); if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s) { drain(s); if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t bytes, hipStream_t s) { drain(s); if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s) { drain(s); if (bytes) memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemset(void *dst, int value, size_t bytes) { drain(nullptr); if (bytes) memset(dst, value, bytes); return hipSuccess; }

hipError_t hipStreamCreate(hipStream_t *s) { *s = new hipemuStream(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new hipemuStream(); return hipSuccess; }
// (the coverage build, covrt.cpp: rank processes of the multi-process tests leave through os._exit, which runs no atexit handler --
// a context that is closed destroys its streams, and that is where their edge list is written out)
extern "C" void hipemu_cov_dump() __attribute__((weak));
hipError_t hipStreamDestroy(hipStream_t s) { drain(s); delete s; if (hipemu_cov_dump) hipemu_cov_dump(); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { return drain(s) ? hipSuccess : (t_last_error = hipErrorUnknown); }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) {
    while (!all_done(e->behind)) std::this_thread::sleep_for(std::chrono::microseconds(100));
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemuEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipemuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s_) {
    hipemuStream *s = stream_of(s_);
    std::lock_guard<std::mutex> lk(s->mu);
    e->behind.clear();
    for (auto &p : s->pending) if (p->state.load() == 0) e->behind.push_back(p);
    e->t_ms = now_ms();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    while (!all_done(e->behind)) std::this_thread::sleep_for(std::chrono::microseconds(100));
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
    if (!all_done(e->behind)) return t_last_error = hipErrorNotReady;
    for (auto &p : e->behind) if (p->state.load() == 2) return t_last_error = hipErrorUnknown;
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipGetLastError() { const hipError_t e = t_last_error; t_last_error = hipSuccess; return e; }
hipError_t hipPeekAtLastError() { return t_last_error; }
const char *hipGetErrorString(hipError_t e) {
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorNotReady: return "not ready";
    default: return "unknown error";
    }
}
hipError_t hipFuncSetAttribute(const void *fn, hipFuncAttribute attr, int value) {
    if (attr != hipFuncAttributeMaxDynamicSharedMemorySize || value < 0 || value > 160 * 1024) return t_last_error = hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(g_attr_mu);
    g_dyn_lds_allow[fn] = std::max(g_dyn_lds_allow[fn], (size_t)value);
    return hipSuccess;
}

// what the emulation saw (tests read it): launches, workgroups, cross-lane operations, of those with a partial wave,
// divergent waits, shuffle reads of lanes outside the operation, launches that needed co-resident workgroups, peak bytes
void hipemu_stats(uint64_t out[8]) {
    out[0] = g_stats.launches; out[1] = g_stats.blocks; out[2] = g_stats.wave_ops; out[3] = g_stats.partial_ops;
    out[4] = g_stats.divergent_ops; out[5] = g_stats.inactive_reads; out[6] = g_stats.coresident_launches; out[7] = g_mem_peak;
}
int hipemu_is_emulator() { return 1; }
// a ticket on a stream (see Pending): returns a handle for hipemu_pending_complete; used by tests/emu/fake_rccl
void *hipemu_pending_new(hipStream_t s_) {
    hipemuStream *s = stream_of(s_);
    auto *h = new std::shared_ptr<Pending>(std::make_shared<Pending>());
    std::lock_guard<std::mutex> lk(s->mu);
    s->pending.push_back(*h);
    return h;
}
void hipemu_pending_complete(void *h_, int failed) {
    auto *h = static_cast<std::shared_ptr<Pending> *>(h_);
    (*h)->state.store(failed ? 2 : 1, std::memory_order_release);
    delete h;
}
}
