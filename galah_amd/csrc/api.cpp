// C-ABI of libgalah_hip.so (include/galah_hip.h): context, ingest, sketch, precluster, ANI.
// Host orchestration only -- all data-parallel work is in sketch.hip / pairs.hip / ani.hip.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <sys/stat.h>

#include "ghip_internal.h"

int ghip_read_fasta_streams(const char *const *paths, size_t n, int threads,
                            std::vector<std::vector<uint8_t>> &streams, std::vector<ghip_genome_stats> &stats,
                            std::string &err);
int ghip_parse_fasta(const uint8_t *buf, size_t n, const char *path, uint8_t *out, size_t cap, size_t *out_len,
                     ghip_genome_stats &st, std::string &err);
bool ghip_slurp(const char *path, std::vector<uint8_t> &buf);
uint64_t ghip_stream_capacity_hint(const char *path);

static thread_local std::string g_init_error;

int ghip_set_error(ghip_ctx *ctx, int code, const std::string &msg) {
    if (ctx) { std::lock_guard<std::mutex> l(ctx->err_mu); ctx->err = msg; } else g_init_error = msg;
    return code;
}

void ghip_ensure_dyn_lds(ghip_ctx *ctx, const void *kernel, size_t bytes) {
    std::lock_guard<std::mutex> l(ctx->dyn_lds_mu);
    size_t &have = ctx->dyn_lds[kernel];
    if (bytes <= have) return;
    hipSetDevice(ctx->device);
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) have = bytes;
    else (void)hipGetLastError();
}

// ------------------------------------------------------------------------------------ memory pool
void *ghip_pool_alloc(ghip_ctx *ctx, size_t bytes) {
    bytes = std::max<size_t>((bytes + 255) / 256 * 256, 256);
    std::lock_guard<std::mutex> pl(ctx->pool_mu);
    ghip_pool_block *best = nullptr;
    for (auto &b : ctx->pool)
        if (!b.used && b.bytes >= bytes && b.bytes <= 2 * bytes + (1u << 20) && (!best || b.bytes < best->bytes)) best = &b;
    if (best) { best->used = true; return best->p; }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {  // release cached blocks and retry once
        for (auto it = ctx->pool.begin(); it != ctx->pool.end();)
            if (!it->used) { hipFree(it->p); it = ctx->pool.erase(it); } else ++it;
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { ghip_set_error(ctx, GHIP_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e)); return nullptr; }
    ctx->pool.push_back({p, bytes, true});
    return p;
}

void ghip_pool_free(ghip_ctx *ctx, void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> pl(ctx->pool_mu);
    for (auto &b : ctx->pool)
        if (b.p == p) { b.used = false; return; }
}

void ghip_io_pool::run(int n, std::function<void(int)> fn) {
    std::unique_lock<std::mutex> l(mu);
    while ((int)threads.size() < n) {
        const int id = (int)threads.size();
        raw.emplace_back();
        ascii.emplace_back();
        threads.emplace_back([this, id] {
            uint64_t seen = 0;
            for (;;) {
                std::function<void(int)> f;
                {
                    std::unique_lock<std::mutex> lw(mu);
                    cv_work.wait(lw, [&] { return stop || (generation != seen && id < active); });
                    if (stop) return;
                    seen = generation;
                    f = job;
                }
                f(id);
                std::lock_guard<std::mutex> ld(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        });
    }
    job = std::move(fn);
    active = pending = n;
    generation++;
    cv_work.notify_all();
    cv_done.wait(l, [&] { return pending == 0; });
    active = 0;
}

void ghip_io_pool::shutdown() {
    { std::lock_guard<std::mutex> l(mu); stop = true; }
    cv_work.notify_all();
    for (auto &t : threads) t.join();
    threads.clear();
    raw.clear();
    ascii.clear();
}

static void ctx_release(ghip_ctx *ctx) {  // called with ctx->mu NOT held
    bool del;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        del = ctx->destroyed && ctx->live_handles == 0;
    }
    if (!del) return;
    hipSetDevice(ctx->device);
    ctx->io.shutdown();
    for (auto &b : ctx->pool) hipFree(b.p);
    for (auto &sl : ctx->ingest_slots) { if (sl.ev) hipEventDestroy(sl.ev); if (sl.p) hipHostFree(sl.p); }
    if (ctx->pin_buf) hipHostFree(ctx->pin_buf);
    if (ctx->ingest_stage) hipFree(ctx->ingest_stage);
    for (hipStream_t cs : ctx->copy_stream) if (cs) hipStreamDestroy(cs);
    if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// ------------------------------------------------------------------------------------ profiling
static void drain_events(ghip_ctx *ctx) {
    for (auto &pe : ctx->pending) {
        hipEventSynchronize(pe.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
            auto &st = ctx->stats[pe.name];
            st.launches++;
            st.total_ms += ms;
        }
        hipEventDestroy(pe.start);
        hipEventDestroy(pe.stop);
    }
    ctx->pending.clear();
}

void ghip_prof_begin(ghip_ctx *ctx, const char *name) {
    if (!ctx->profile) return;
    ghip_pending_event pe;
    pe.name = name;
    hipEventCreate(&pe.start);
    hipEventCreate(&pe.stop);
    hipEventRecord(pe.start, ctx->stream);
    ctx->pending.push_back(pe);
}

void ghip_prof_end(ghip_ctx *ctx) {
    if (!ctx->profile || ctx->pending.empty()) return;
    hipEventRecord(ctx->pending.back().stop, ctx->stream);
}

namespace {

template <typename T>
int dmalloc(ghip_ctx *ctx, T **p, size_t count) {
    *p = (T *)ghip_pool_alloc(ctx, std::max<size_t>(count, 1) * sizeof(T));
    return *p ? GHIP_OK : GHIP_EHIP;
}

// Copies between pageable host memory and the device go through a pinned bounce buffer from 256 KiB on.  The runtime
// stages small pageable copies itself; a larger buffer it pins in place, and the NEXT copy or event wait of the process
// then stalls for 12-30 ms while it is unpinned (measured at 50 000 genomes: the 3.6 MB candidate list coming back made
// the first ANI round's 360 KB upload take 12-32 ms; scripts/ani_round_overhead.py).  One memcpy at host speed instead.
constexpr size_t GHIP_PIN_MIN = 256u << 10, GHIP_PIN_MAX = 32u << 20;
static void *pinned_bounce(ghip_ctx *ctx, size_t bytes) {   // ctx->pin_mu held; nullptr: no pinned memory to be had (the caller copies directly)
    if (ctx->pin_bytes >= bytes) return ctx->pin_buf;
    if (ctx->pin_buf) { hipHostFree(ctx->pin_buf); ctx->pin_buf = nullptr; ctx->pin_bytes = 0; }
    const size_t want = std::min(bytes + bytes / 2, GHIP_PIN_MAX);
    if (hipHostMalloc(&ctx->pin_buf, want, hipHostMallocDefault) != hipSuccess) { ctx->pin_buf = nullptr; (void)hipGetLastError(); return nullptr; }
    ctx->pin_bytes = want;
    return ctx->pin_buf;
}

template <typename T>
int h2d(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    const size_t bytes = count * sizeof(T);
    if (bytes >= GHIP_PIN_MIN) {
        std::lock_guard<std::mutex> pl(ctx->pin_mu);
        if (void *p = pinned_bounce(ctx, std::min(bytes, GHIP_PIN_MAX))) {
            for (size_t at = 0; at < bytes; at += GHIP_PIN_MAX) {   // (longer copies: piece by piece)
                const size_t m = std::min(GHIP_PIN_MAX, bytes - at);
                memcpy(p, reinterpret_cast<const char *>(src) + at, m);
                GHIP_HIP_CHECK(ctx, hipMemcpyAsync(reinterpret_cast<char *>(dst) + at, p, m, hipMemcpyHostToDevice, ctx->stream));
                GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            }
            return GHIP_OK;
        }
    }
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // src may be pageable / short-lived
    return GHIP_OK;
}

// Batched form: several small copies, ONE synchronisation (each costs ~15 us of host time).  The host buffers must
// stay alive until stream_sync().
template <typename T>
int h2d_nosync(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return GHIP_OK;
}
template <typename T>
int d2h_nosync(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return GHIP_OK;
}
static int stream_sync(ghip_ctx *ctx) {
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GHIP_OK;
}

template <typename T>
int d2h(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    const size_t bytes = count * sizeof(T);
    if (bytes >= GHIP_PIN_MIN) {
        std::lock_guard<std::mutex> pl(ctx->pin_mu);
        if (void *p = pinned_bounce(ctx, std::min(bytes, GHIP_PIN_MAX))) {
            for (size_t at = 0; at < bytes; at += GHIP_PIN_MAX) {
                const size_t m = std::min(GHIP_PIN_MAX, bytes - at);
                GHIP_HIP_CHECK(ctx, hipMemcpyAsync(p, reinterpret_cast<const char *>(src) + at, m, hipMemcpyDeviceToHost, ctx->stream));
                GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                memcpy(reinterpret_cast<char *>(dst) + at, p, m);
            }
            return GHIP_OK;
        }
    }
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GHIP_OK;
}

// Deterministic ANI index layout (shards must agree): Poisson(L/c) seeds, +10 % + 256 slack.
// CPUs' worth of time the process may use per scheduling period (cgroup v2 cpu.max; v1 cfs quota), 0 = unlimited / unknown
double ghip_cpu_quota() {
    static const double q = [] {
        double quota = 0, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[64] = {0};
            if (fscanf(f, "%63s %lf", a, &period) == 2 && strcmp(a, "max") != 0) quota = atof(a);
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &quota) != 1) quota = 0; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = 0; fclose(g); }
        }
        return (quota > 0 && period > 0) ? quota / period : 0.0;
    }();
    return q;
}

// a multiple of the segment count: the unordered list is GHIP_ANI_SEGMENTS equal parts (seed_common.h)
// Room for the expected seeds of each segment (m = len / (SEGMENTS c), about Poisson), six standard deviations, a quarter
// more for repeats -- all copies of a repeated k-mer land in ONE segment: an insertion-sequence family of 300 copies adds
// a few hundred seeds to two or three segments of a 5 Mb genome -- and a constant.  The list is 8 B per seed, ~6 % of the
// bases it indexes, so room is cheap; an overflow is not: it re-seeds the whole batch with exact capacities (with four
// standard deviations and 10 %, the first form, 50 000 genomes of 1 Mb overflowed a handful of their 400 000 segments
// every time: +33 ms).
}  // namespace

// the same for the other translation units (comm.cpp): bytes between pageable host memory and the device, synchronous
int ghip_copy_to_device(ghip_ctx *ctx, void *d_dst, const void *src, size_t bytes) { return h2d(ctx, static_cast<char *>(d_dst), static_cast<const char *>(src), bytes); }
int ghip_copy_to_host(ghip_ctx *ctx, void *dst, const void *d_src, size_t bytes) { return d2h(ctx, static_cast<char *>(dst), static_cast<const char *>(d_src), bytes); }
// Seed density of a genome (oracle: go_ani_density): the base density c while the genome holds ~8192 seeds at it, else four
// times denser, and so on down to every k-mer (125 -> 31 -> 7 -> 1): a 200 kb plasmid or a 5 kb contig carries as many
// seeds as a genome does, and its ANI is as little noisy (skani's own remedy is --small-genomes = -c 30, chosen by the
// user for the whole run: src/skani.rs:152-153).
uint32_t ghip_ani_density(uint64_t len, uint32_t c) {
    uint32_t t = c ? c : 1;
    while (t > 1 && len < (uint64_t)GHIP_ANI_SEEDS_WANTED * t) t = t / 4 ? t / 4 : 1;
    return t;
}
namespace {
uint64_t ghip_ani_seed_capacity(uint64_t len, uint32_t c) {
    const double m = (double)len / ((double)GHIP_ANI_SEGMENTS * (double)c);
    const uint64_t segcap = (uint64_t)(m + 6.0 * std::sqrt(m) + m / 4.0) + 24;
    return segcap * GHIP_ANI_SEGMENTS;
}

// fn(begin, end) over [0, n) on up to max_threads threads of at least min_per_thread items each (the caller's thread
// takes the last range); spawning costs ~30 us per thread, so short loops stay serial (callers pass min_per_thread = 10 000)
template <typename F>
void parallel_ranges(size_t n, size_t min_per_thread, size_t max_threads, F &&fn) {
    size_t t = std::min<size_t>({max_threads, n / std::max<size_t>(min_per_thread, 1), std::max<size_t>(1, std::thread::hardware_concurrency())});
    if (t <= 1) { fn((size_t)0, n); return; }
    const size_t per = (n + t - 1) / t;
    std::vector<std::thread> pool;
    for (size_t x = 0; x + 1 < t; x++) pool.emplace_back([&fn, x, per, n] { fn(std::min(n, x * per), std::min(n, (x + 1) * per)); });
    fn(std::min(n, (t - 1) * per), n);
    for (auto &th : pool) th.join();
}

uint32_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return (uint32_t)p;
}

// 1 - mash_distance exactly as the reference computes it (src/finch.rs:78-86 with finch's
// jaccard = common/total, mash = -ln(2j/(1+j))/k clamped to [0,1]; Rust f64::max/min drop NaN).
double finch_ani(uint64_t common, uint64_t total, uint32_t k) {
    double j = (double)common / (double)total;
    double mash = -1.0 * std::log((2.0 * j) / (1.0 + j)) / (double)k;
    double lo = std::isnan(mash) ? 0.0 : (mash > 0.0 ? mash : 0.0);  // f64::max(0, mash)
    double cl = lo < 1.0 ? lo : 1.0;                                   // f64::min(1, .)
    return 1.0 - cl;
}

// strtof(sprintf("%.2f", x)) for x in [0, 100]: hundredths -> f32 from a table built with strtof itself;
// the hundredth is found arithmetically unless x*100 is within 1e-6 of a rounding tie, where printf's exact
// decimal rounding is consulted.
float two_decimals_as_f32(double x) {
    static std::vector<float> table = [] {
        std::vector<float> t(10001);
        char txt[32];
        for (int k = 0; k <= 10000; k++) { snprintf(txt, sizeof txt, "%d.%02d", k / 100, k % 100); t[k] = strtof(txt, nullptr); }
        return t;
    }();
    const double y = x * 100.0;
    const double fl = std::floor(y);
    const double frac = y - fl;
    if (!(x >= 0.0 && x <= 100.0) || std::fabs(frac - 0.5) < 1e-6) {
        char txt[64];
        snprintf(txt, sizeof txt, "%.2f", x);
        return strtof(txt, nullptr);
    }
    return table[(int)fl + (frac > 0.5 ? 1 : 0)];
}

struct DeviceFree {  // scratch buffers go back to the pool; ctx->mu is held by the caller
    ghip_ctx *ctx;
    std::vector<void *> ptrs;
    explicit DeviceFree(ghip_ctx *c) : ctx(c) {}
    ~DeviceFree() { for (void *p : ptrs) ghip_pool_free(ctx, p); }
    template <typename T> void add(T *p) { ptrs.push_back((void *)p); }
};

// h2d on a given stream (the ingest keeps off the context's compute stream)
template <typename T>
int h2d_on(ghip_ctx *ctx, hipStream_t st, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, st));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(st));
    return GHIP_OK;
}

int build_work(ghip_ctx *ctx, ghip_genomes *g, hipStream_t st = nullptr) {
    if (!st) st = ctx->stream;
    std::vector<ghip_sketch_work> work;
    for (size_t i = 0; i < g->n; i++) {
        uint64_t chunks = (g->lens[i] + GHIP_SKETCH_CHUNK - 1) / GHIP_SKETCH_CHUNK;
        for (uint64_t c = 0; c < chunks; c++) work.push_back({(uint32_t)i, (uint32_t)c});
    }
    g->n_work = work.size();
    int rc = dmalloc(ctx, &g->d_work, work.size());
    if (rc) return rc;
    rc = h2d_on(ctx, st, g->d_work, work.data(), work.size());
    if (rc) return rc;
    std::vector<uint32_t> ident(g->n);
    for (size_t i = 0; i < g->n; i++) ident[i] = (uint32_t)i;
    rc = dmalloc(ctx, &g->d_identity, g->n);
    if (rc) return rc;
    return h2d_on(ctx, st, g->d_identity, ident.data(), g->n);
}

// base offset of the genome after one of `len` bases that starts at `off`: room for the bases and GHIP_TAIL_PAD invalid
// positions, rounded up to the alignment of a genome's first base
inline uint64_t next_genome_offset(uint64_t off, uint64_t len) { return off + (len + GHIP_TAIL_PAD + GHIP_BASE_ALIGN - 1) / GHIP_BASE_ALIGN * GHIP_BASE_ALIGN; }

// allocates the resident arrays for g->total_alloc base positions, all invalid (the validity bitmap is zero-filled; the
// 2-bit codes of invalid positions are never looked at)
int alloc_bases(ghip_ctx *ctx, ghip_genomes *g, hipStream_t st) {
    int rc = dmalloc(ctx, &g->d_packed, g->total_alloc / 16);
    if (rc) return rc;
    if ((rc = dmalloc(ctx, &g->d_valid, g->total_alloc / 32))) return rc;
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(g->d_valid, 0, g->total_alloc / 32 * sizeof(uint32_t), st));
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(g->d_packed, 0, g->total_alloc / 16 * sizeof(uint32_t), st));   // (tidy: keeps saved / compared images deterministic)
    return GHIP_OK;
}

// lays genomes out (first base at a multiple of GHIP_BASE_ALIGN, GHIP_TAIL_PAD invalid positions after each)
int layout_genomes(ghip_ctx *ctx, ghip_genomes *g, const std::vector<uint64_t> &lens) {
    g->n = lens.size();
    g->lens = lens;
    g->starts.resize(g->n);
    uint64_t off = 0;
    g->total_bases = 0;
    for (size_t i = 0; i < g->n; i++) {
        g->starts[i] = off;
        off = next_genome_offset(off, lens[i]);
        g->total_bases += lens[i];
    }
    g->total_alloc = off + 256;
    int rc = alloc_bases(ctx, g, ctx->stream);
    if (rc) return rc;
    if ((rc = dmalloc(ctx, &g->d_starts, g->n))) return rc;
    if ((rc = dmalloc(ctx, &g->d_lens, g->n))) return rc;
    if ((rc = h2d(ctx, g->d_starts, g->starts.data(), g->n))) return rc;
    if ((rc = h2d(ctx, g->d_lens, g->lens.data(), g->n))) return rc;
    return build_work(ctx, g);
}

// Stream bytes on the host -> the resident form, genome by genome through ONE device staging buffer (the copies and the
// pack kernels are ordered by `st`; pageable sources are staged by the runtime, so the host buffers are free on return).
int upload_streams(ghip_ctx *ctx, ghip_genomes *g, hipStream_t st, const std::function<const uint8_t *(size_t)> &bytes_of) {
    uint64_t longest = 0;
    for (uint64_t l : g->lens) longest = std::max(longest, l);
    if (longest == 0) return GHIP_OK;
    uint8_t *d_stage = nullptr;
    int rc = dmalloc(ctx, &d_stage, longest + 64);
    if (rc) return rc;
    for (size_t i = 0; i < g->n && rc == GHIP_OK; i++) {
        if (!g->lens[i]) continue;
        if (hipMemcpyAsync(d_stage, bytes_of(i), g->lens[i], hipMemcpyHostToDevice, st) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "base upload failed");
        else ghip_launch_pack_bases(st, d_stage, g->lens[i], g->starts[i], g->d_packed, g->d_valid);
    }
    if (rc == GHIP_OK && (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)) rc = ghip_set_error(ctx, GHIP_EHIP, "base upload failed");
    else if (rc != GHIP_OK) hipStreamSynchronize(st);
    ghip_pool_free(ctx, d_stage);
    return rc;
}

}  // namespace

// ------------------------------------------------------------------------------------ options
namespace {
std::mutex g_opt_mu;
bool g_opt_ready = false;
ghip_options g_opt{};

ghip_options options_from_environment() {
    ghip_options o{};
    o.struct_size = sizeof(ghip_options);
    auto is = [](const char *v, const char *w) { return v && !strcmp(v, w); };
    auto num = [](const char *name, uint32_t dflt) { const char *e = getenv(name); return e && *e ? (uint32_t)strtoul(e, nullptr, 10) : dflt; };
    auto set = [](const char *name) { const char *e = getenv(name); return e != nullptr; };
    const char *pk = getenv("GHIP_PAIR_KERNEL"), *jr = getenv("GHIP_JOIN_RANKS"), *ing = getenv("GHIP_INGEST"), *pl = getenv("GHIP_PIPELINE");
    const char *nl = getenv("GHIP_NO_LIBDEFLATE");
    o.pair_form = is(pk, "join") ? GHIP_PAIR_JOIN : is(pk, "probe") ? GHIP_PAIR_PROBE : is(pk, "merge") ? GHIP_PAIR_MERGE : GHIP_PAIR_AUTO;
    o.join_ranks = is(jr, "records") ? GHIP_JOIN_RECORDS : is(jr, "replicate") ? GHIP_JOIN_REPLICATE : GHIP_JOIN_HASH;
    o.ingest_form = is(ing, "ascii") ? GHIP_INGEST_ASCII : is(ing, "pageable") ? GHIP_INGEST_PAGEABLE : is(ing, "two-phase") ? GHIP_INGEST_TWO_PHASE : GHIP_INGEST_PACKED;
    o.ingest_groups = set("GHIP_INGEST_NO_GROUPS") ? 0 : 1;
    o.io_threads_plain = num("GHIP_INGEST_THREADS_PLAIN", 0);
    o.io_threads_gz = num("GHIP_INGEST_THREADS_GZ", 0);
    o.copy_streams = std::min(4u, std::max(1u, num("GHIP_COPY_STREAMS", 2)));
    o.use_libdeflate = (nl && *nl && *nl != '0') ? 0 : 1;
    o.pipeline_pieces = is(pl, "0") ? 0 : 1;
    o.overlap_binning = set("GHIP_NO_OVERLAP") ? 0 : 1;
    o.lazy_flush_below = num("GHIP_LAZY_FLUSH_BELOW", 512);
    o.cluster_threads = num("GHIP_CLUSTER_THREADS", 0);
    o.ani_force_general = set("GHIP_ANI_FORCE_GENERAL") ? 1 : 0;
    o.ani_tall_below = num("GHIP_ANI_TALL_BELOW", 200);
    o.debug = (set("GHIP_INGEST_DEBUG") ? GHIP_DEBUG_INGEST : 0) | (set("GHIP_PRECLUSTER_DEBUG") ? GHIP_DEBUG_PRECLUSTER : 0) |
              (set("GHIP_COMM_DEBUG") ? GHIP_DEBUG_COMM : 0) | (set("GHIP_CLUSTER_DEBUG") ? GHIP_DEBUG_CLUSTER : 0) | (set("GHIP_ANI_DEBUG") ? GHIP_DEBUG_ANI : 0);
    o.pair_debug = num("GHIP_PAIR_DEBUG", 0);
    o.probe_arranged = num("GHIP_PROBE_ARRANGED", 0);   // (likewise off until measured)
    o.join_fused = num("GHIP_JOIN_FUSED", 0);   // (off until a GPU run has shown it byte-identical and faster: profiles/r04*)
    return o;
}

// the fields a caller's (possibly shorter, older) struct holds are taken; out-of-range values are refused
int merge_options(ghip_options &dst, const ghip_options *src) {
    if (!src || src->struct_size < 2 * sizeof(uint32_t) || src->struct_size % sizeof(uint32_t)) return GHIP_EINVAL;
    ghip_options o = dst;
    memcpy(&o, src, std::min<size_t>(src->struct_size, sizeof(ghip_options)));
    o.struct_size = sizeof(ghip_options);
    if (o.pair_form > GHIP_PAIR_MERGE || o.join_ranks > GHIP_JOIN_REPLICATE || o.ingest_form > GHIP_INGEST_TWO_PHASE || o.copy_streams < 1 ||
        o.copy_streams > 4 || o.fault_stage > GHIP_FAULT_ANI_ROUND)
        return GHIP_EINVAL;
    dst = o;
    return GHIP_OK;
}
}  // namespace

ghip_options ghip_process_options() {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (!g_opt_ready) { g_opt = options_from_environment(); g_opt_ready = true; }
    return g_opt;
}

extern "C" int ghip_get_options(const ghip_ctx *ctx, ghip_options *out) {
    if (!out) return GHIP_EINVAL;
    *out = ctx ? ctx->opt : ghip_process_options();
    return GHIP_OK;
}

extern "C" int ghip_set_options(ghip_ctx *ctx, const ghip_options *opt) {
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        const int rc = merge_options(ctx->opt, opt);
        return rc ? ghip_set_error(ctx, rc, "ghip_set_options: a field is out of range or struct_size is wrong") : GHIP_OK;
    }
    (void)ghip_process_options();   // the environment first
    std::lock_guard<std::mutex> lk(g_opt_mu);
    return merge_options(g_opt, opt);
}

// ------------------------------------------------------------------------------------ context
extern "C" int ghip_abi_version(void) { return GHIP_ABI_VERSION; }

extern "C" int ghip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ghip_init(int device, ghip_ctx **out) {
    if (!out) return GHIP_EINVAL;
    *out = nullptr;
    int n = ghip_device_count();
    if (n <= 0) return ghip_set_error(nullptr, GHIP_EHIP, "no HIP device visible: the galah HIP back-end needs an AMD GPU");
    if (device < 0 || device >= n) return ghip_set_error(nullptr, GHIP_EINVAL, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return ghip_set_error(nullptr, GHIP_EHIP, "hipSetDevice failed");
    ghip_ctx *ctx = new ghip_ctx();
    ctx->device = device;
    ctx->opt = ghip_process_options();
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ghip_set_error(nullptr, GHIP_EHIP, "hipStreamCreate failed");
    }
    ctx->stream = ctx->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    *out = ctx;
    return GHIP_OK;
}

extern "C" void ghip_destroy(ghip_ctx *ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
        drain_events(ctx);
        ctx->destroyed = true;
    }
    ctx_release(ctx);  // deferred until the last genomes/sketches/index handle is freed
}

// the text is copied out under the error lock into a per-thread buffer: another thread of the same context (the ingest
// producer) may be writing a new message meanwhile
extern "C" const char *ghip_last_error(const ghip_ctx *ctx) {
    if (!ctx) return g_init_error.c_str();
    static thread_local std::string copy;
    ghip_ctx *c = const_cast<ghip_ctx *>(ctx);
    { std::lock_guard<std::mutex> l(c->err_mu); copy = c->err; }
    return copy.c_str();
}

extern "C" int ghip_set_stream(ghip_ctx *ctx, void *hip_stream) {
    if (!ctx) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return GHIP_OK;
}

extern "C" int ghip_synchronize(ghip_ctx *ctx) {
    if (!ctx) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GHIP_OK;
}

extern "C" int ghip_memcpy_d2d(ghip_ctx *ctx, void *d_dst, const void *d_src, size_t nbytes) {
    if (!ctx || (nbytes && (!d_dst || !d_src))) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (nbytes) GHIP_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, d_src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return GHIP_OK;
}

extern "C" int ghip_profile_enable(ghip_ctx *ctx, int enable) {
    if (!ctx) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!enable) drain_events(ctx);
    ctx->profile = enable != 0;
    return GHIP_OK;
}

extern "C" int ghip_profile_reset(ghip_ctx *ctx) {
    if (!ctx) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    drain_events(ctx);
    ctx->stats.clear();
    return GHIP_OK;
}

extern "C" int ghip_kernel_stats(ghip_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms) {
    if (!ctx || !kernel) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    drain_events(ctx);
    auto it = ctx->stats.find(kernel);
    if (launches) *launches = it == ctx->stats.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == ctx->stats.end() ? 0.0 : it->second.total_ms;
    return GHIP_OK;
}

extern "C" int ghip_selftest_hash_floor(ghip_ctx *ctx, uint64_t wave_positions, double *out_ms) {
    if (!ctx || !out_ms || wave_positions == 0) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return ghip_launch_hash_floor(ctx, wave_positions, out_ms);
}

extern "C" void ghip_free(void *p) { free(p); }

// ------------------------------------------------------------------------------------ genomes
static void free_genomes_locked(ghip_genomes *g) {  // ctx->mu held
    ghip_ctx *ctx = g->ctx;
    ghip_pool_free(ctx, g->d_packed); ghip_pool_free(ctx, g->d_valid); ghip_pool_free(ctx, g->d_starts); ghip_pool_free(ctx, g->d_lens);
    ghip_pool_free(ctx, g->d_work); ghip_pool_free(ctx, g->d_identity);
    ctx->live_handles--;
    delete g;
}

extern "C" void ghip_genomes_free(ghip_genomes *g) {
    if (!g) return;
    ghip_ctx *ctx = g->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStreamSynchronize(ctx->stream);  // nothing in flight may still read the recycled blocks
        free_genomes_locked(g);
    }
    ctx_release(ctx);
}

extern "C" int ghip_genomes_from_host(ghip_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, size_t n,
                                      ghip_genomes **out) {
    if (!ctx || !out || (n && (!bytes || !offsets))) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> lens(n);
    for (size_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) return ghip_set_error(ctx, GHIP_EINVAL, "offsets must be non-decreasing");
        lens[i] = offsets[i + 1] - offsets[i];
    }
    ghip_genomes *g = new ghip_genomes();
    g->ctx = ctx;
    ctx->live_handles++;
    int rc = layout_genomes(ctx, g, lens);
    // caller-supplied bytes are arbitrary: anything but A,C,G,T becomes an invalid position ('N' when read back)
    if (rc == GHIP_OK) rc = upload_streams(ctx, g, ctx->stream, [&](size_t i) { return bytes + offsets[i]; });
    if (rc != GHIP_OK) { free_genomes_locked(g); return rc; }
    *out = g;
    return GHIP_OK;
}

int ghip_read_fasta_stream(const char *path, std::vector<uint8_t> &out, ghip_genome_stats &st, std::string &err);

extern "C" int ghip_fasta_stream(const char *path, uint8_t **out_stream, size_t *out_len, uint64_t out_stats[3]) {
    if (!path || !out_stream || !out_len) return GHIP_EINVAL;
    std::vector<uint8_t> v;
    ghip_genome_stats st;
    std::string err;
    const int rc = ghip_read_fasta_stream(path, v, st, err);
    if (rc != GHIP_OK) return ghip_set_error(nullptr, rc, err);
    uint8_t *p = (uint8_t *)malloc(std::max<size_t>(v.size(), 1));
    if (!p) return GHIP_ENOMEM;
    memcpy(p, v.data(), v.size());
    *out_stream = p;
    *out_len = v.size();
    if (out_stats) { out_stats[0] = st.num_contigs; out_stats[1] = st.num_ambiguous_bases; out_stats[2] = st.n50; }
    return GHIP_OK;
}

// Two-phase form: parse every file into host vectors, then lay out by the exact lengths and copy.  Used when a
// stream outgrows its capacity hint (multi-member gzip) and as the reference point of the pipelined form below.
static int genomes_from_files_two_phase(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads, ghip_genomes **out) {
    std::vector<std::vector<uint8_t>> streams;
    std::vector<ghip_genome_stats> stats;
    std::string err;
    int rc = ghip_read_fasta_streams(paths, n, io_threads, streams, stats, err);
    if (rc != GHIP_OK) return ghip_set_error(ctx, rc, err);
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> lens(n);
    for (size_t i = 0; i < n; i++) lens[i] = streams[i].size();
    ghip_genomes *g = new ghip_genomes();
    g->ctx = ctx;
    g->stats = stats;
    ctx->live_handles++;
    rc = layout_genomes(ctx, g, lens);
    if (rc == GHIP_OK) rc = upload_streams(ctx, g, ctx->stream, [&](size_t i) { return (const uint8_t *)streams[i].data(); });
    if (rc != GHIP_OK) { free_genomes_locked(g); return rc; }
    *out = g;
    return GHIP_OK;
}

// Pipelined ingest: the device layout is fixed up front from per-file capacity hints (a stream is never longer
// than its plain file; a gzip trailer holds the uncompressed size), so every worker thread parses a file and ships
// it straight to its final place in HBM while the other threads are still parsing -- parsing (~12 GB/s per
// thread on clean lines) and PCIe (~55 GB/s) overlap instead of adding up.  Gaps between capacity and actual length stay 'N'
// (the buffer is 'N'-filled).
static int genomes_from_files_impl(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads, const uint64_t *known_caps,
                                   ghip_genomes **out);

extern "C" int ghip_genomes_from_files(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads,
                                       ghip_genomes **out) {
    return genomes_from_files_impl(ctx, paths, n, io_threads, nullptr, out);
}

// known_caps (nullable): the capacity hints of the files, already looked up by the caller (one stat per file is 0.1-0.2 s
// for 100 000 contig files: not twice)
static int genomes_from_files_impl(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads, const uint64_t *known_caps,
                                   ghip_genomes **out) {
    if (!ctx || !out || (n && !paths)) return GHIP_EINVAL;
    const ghip_options opt = ctx->opt;
    if (opt.ingest_form == GHIP_INGEST_TWO_PHASE)
        return genomes_from_files_two_phase(ctx, paths, n, io_threads, out);
    const bool dbg = ghip_dbg(opt, GHIP_DEBUG_INGEST);
    const auto w0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    std::vector<uint64_t> cap(n);
    uint64_t max_cap = 0;
    if (known_caps) std::copy(known_caps, known_caps + n, cap.begin());
    else parallel_ranges(n, 64, (size_t)std::min(std::max(1, io_threads), 8), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) cap[i] = ghip_stream_capacity_hint(paths[i]);
    });
    for (size_t i = 0; i < n; i++) max_cap = std::max(max_cap, cap[i]);
    size_t n_gz = 0;
    for (size_t i = 0; i < n; i++) { const size_t l = strlen(paths[i]); n_gz += (l > 3 && !strcmp(paths[i] + l - 3, ".gz")) ? 1 : 0; }
    const double w_hint = since(w0);
    bool overflow = false;
    {
        // The ingest never takes `mu` and never touches the context's compute stream: its device memory comes from the
        // (internally locked) pool, its copies -- the small layout arrays too -- go over the copy streams.  Another thread
        // may therefore hold `mu` for the length of its kernels meanwhile: ghip_sketch_and_index_files overlaps the
        // sketch pass of batch b with the ingest of batch b + 1.  (While the set-up still took `mu`, batch b + 1 could
        // not START before the kernels of batch b had finished: the two alternated instead of overlapping.)
        std::lock_guard<std::mutex> ingest_lk(ctx->ingest_mu);
        GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        if (ctx->n_copy_streams == 0) ctx->n_copy_streams = (int)std::min(4u, std::max(1u, opt.copy_streams));
        for (int x = 0; x < ctx->n_copy_streams; x++) {
            hipStream_t &cs = ctx->copy_stream[x];
            if (!cs && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { cs = nullptr; (void)hipGetLastError(); }
        }
        hipStream_t fill_stream = ctx->copy_stream[0] ? ctx->copy_stream[0] : ctx->stream;
        ghip_genomes *g = new ghip_genomes();
        g->ctx = ctx;
        g->n = n;
        g->stats.assign(n, ghip_genome_stats());
        g->lens.assign(n, 0);
        g->starts.resize(n);
        ctx->live_handles++;
        uint64_t off = 0;
        for (size_t i = 0; i < n; i++) { g->starts[i] = off; off = next_genome_offset(off, cap[i]); }
        g->total_alloc = off + 256;
        int rc = alloc_bases(ctx, g, fill_stream);
        if (rc == GHIP_OK && hipStreamSynchronize(fill_stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "sync failed");
        if (rc != GHIP_OK) { free_genomes_locked(g); return rc; }
        const double w_alloc = since(w0) - w_hint;

        int threads = std::max(1, io_threads);
        threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(n, 1));
        // Plain files need ~0.2 thread-seconds per GB (read + parse) against 19 ms per GB of PCIe time: a dozen threads
        // keep the copy streams busy, and many more concurrent readers of the page cache only slow each other down
        // (64 threads: 4 thread-seconds of read() per 640 MB instead of 0.1).  Measured files -> clusters, 1 000 x 5 Mb:
        // 154 / 123 / 142 / 150 ms with 8 / 12 / 16 / 24 threads.  gzip input is inflate-bound (CPU): every thread the
        // caller offers is used, up to ~1.5x the CPUs the process may actually use (below).
        {
            const uint32_t mt = opt.io_threads_plain;
            // (the ASCII form is PCIe-bound: a dozen readers; the packed form ships a quarter of the bytes and is bound
            // by the CPUs the process may use -- 1 000 x 5 Mb on the 16-CPU-quota boxes: 127 / 90 / 78 / 74 / 86 / 94 ms
            // with 8 / 12 / 16 / 20 / 24 / 32 readers)
            const bool ascii_form = opt.ingest_form == GHIP_INGEST_ASCII || opt.ingest_form == GHIP_INGEST_PAGEABLE;
            const double q = ghip_cpu_quota();
            const int plain_cap = mt ? (int)mt : (ascii_form ? 12 : (q > 0 ? std::max(8, (int)(q * 1.25 + 0.5)) : 16));
            if (n_gz == 0) threads = std::min(threads, plain_cap);
            else {
                // ... of the CPUs the process may actually use: under a cgroup quota (the GPU boxes: 256 logical CPUs, 16
                // CPUs' worth of time) threads beyond ~1.5x the quota only get throttled -- 1 000 gzip files: 0.58 s with
                // 64 threads, 0.42-0.47 s with 16-32
                const uint32_t gt = opt.io_threads_gz;
                const double quota = ghip_cpu_quota();
                const int gz_cap = gt ? (int)gt : (quota > 0 ? std::max(8, (int)(quota * 1.5 + 0.5)) : threads);
                threads = std::min(threads, gz_cap);
            }
        }
        // two heap buffers of the largest file per thread: keep their total below 8 GiB
        threads = (int)std::min<uint64_t>((uint64_t)threads, std::max<uint64_t>(1, (8ull << 30) / (2 * (max_cap + 64))));
        std::atomic<size_t> next{0};
        std::atomic<int> status{GHIP_OK};
        std::atomic<bool> over{false};
        std::mutex emu;
        std::string err;
        struct AtomicD { std::atomic<double> v{0}; void operator+=(double d) { double o = v.load(); while (!v.compare_exchange_weak(o, o + d)) {} } double load() const { return v.load(); } };
        AtomicD t_alloc, t_read, t_parse;
        auto fail = [&](int code, const std::string &msg) {
            std::lock_guard<std::mutex> l2(emu);
            if (status.load() == GHIP_OK) { status = code; err = msg; }
        };
        // Staging.  Measured on the MI355X host (scripts/ingest_probe.py): ONE pinned 5 GB copy runs at 53 GB/s, one
        // pageable copy at 16-21 GB/s, blocking pageable copies from 16-128 threads level off at 36 GB/s whatever the
        // thread count (the runtime stages them through its own pinned buffers) -- the 0.14 s floor of the previous
        // ingest -- and pageable copies issued NEXT TO pinned asynchronous ones collapse to 3-10 GB/s.  So every copy
        // leaves from a pinned buffer: the threads share a pool of 32 pinned slots (kept in the context: hipHostMalloc
        // costs 0.16 ms/MB); a thread reads / inflates its file into its own heap buffer first (the CPU-heavy part, all
        // io_threads at once), then takes a slot, parses into it, queues the asynchronous copy on one of two copy streams
        // and hands the slot back "in flight"; the next taker waits for its event.  Files above GHIP_PINNED_SLOT_MAX (and
        // everything when GHIP_INGEST=pageable) take blocking pageable copies.
        constexpr size_t GHIP_PINNED_SLOT_MAX = 24u << 20;
        constexpr size_t GHIP_PINNED_SLOTS = 32;
        bool use_pinned = opt.ingest_form != GHIP_INGEST_PAGEABLE;
        for (int x = 0; x < ctx->n_copy_streams; x++) use_pinned = use_pinned && ctx->copy_stream[x];
        // (the packed form stages a quarter of the bytes: streams of up to four times the size go through the slots)
        const bool packed_wanted = use_pinned && opt.ingest_form != GHIP_INGEST_ASCII;
        // (a slot holds the largest file -- or, where the files are small, a group of them: 4 Mbases, see `units` below)
        uint64_t sum_cap = 0;
        for (size_t i = 0; i < n; i++) sum_cap += cap[i] + 2 * GHIP_TAIL_PAD;
        const size_t slot_want = std::max<size_t>((size_t)max_cap + 64, packed_wanted ? (size_t)std::min<uint64_t>(4u << 20, sum_cap) : 0);
        const size_t slot_bytes = std::min<size_t>((slot_want + 4095) / 4096 * 4096, packed_wanted ? 4 * GHIP_PINNED_SLOT_MAX : GHIP_PINNED_SLOT_MAX);
        // Packed form (the default; GHIP_INGEST=ascii turns it off): the stream crosses PCIe as 2-bit codes plus the runs
        // of its other bytes -- a quarter of the bytes, and PCIe is what bounds files -> clusters (5 GB: 88 ms).  The
        // codes ARE the resident form: they are copied straight to their place; the run table goes to a small device
        // staging area that belongs to the slot, and a kernel queued behind the copy on the same copy stream turns it
        // into the genome's validity bits (sketch.hip: ghip_launch_valid_from_runs).  The ASCII forms stage the stream
        // bytes on the device and pack them there (ghip_launch_pack_bases).
        const bool packed_mode = packed_wanted;
        const size_t packed_slot_bytes = (slot_bytes / 4 + (64u << 10) + 4095) / 4096 * 4096;   // codes + room for ~5 000 runs
        const size_t host_slot_bytes = packed_mode ? packed_slot_bytes : slot_bytes;   // what a pinned slot has to hold
        const size_t stage_bytes = packed_mode ? (64u << 10) + 4096 : slot_bytes;      // ... and its device staging area: the runs / the stream bytes
        size_t n_slots = 0;
        if (use_pinned) {
            const size_t want = std::min<size_t>(GHIP_PINNED_SLOTS, (size_t)2 * threads);
            if (ctx->ingest_slots.size() < want) ctx->ingest_slots.resize(want);
            for (size_t x = 0; x < ctx->ingest_slots.size(); x++) {
                ghip_ctx::pinned_slot &sl = ctx->ingest_slots[x];
                if (x < want && sl.bytes < host_slot_bytes) {
                    if (sl.p) hipHostFree(sl.p);
                    sl.p = nullptr; sl.bytes = 0;
                    if (hipHostMalloc((void **)&sl.p, host_slot_bytes, hipHostMallocDefault) == hipSuccess) sl.bytes = host_slot_bytes;
                    else { sl.p = nullptr; (void)hipGetLastError(); }
                }
                if (!sl.ev && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) sl.ev = nullptr;
                sl.state = 0;
            }
            {   // one device allocation for all the staging areas (32 hipMalloc calls cost ~20 ms of a first call)
                const size_t need = want * stage_bytes;
                if (ctx->ingest_stage_bytes < need) {
                    if (ctx->ingest_stage) hipFree(ctx->ingest_stage);
                    ctx->ingest_stage = nullptr; ctx->ingest_stage_bytes = 0;
                    if (hipMalloc((void **)&ctx->ingest_stage, need) == hipSuccess) ctx->ingest_stage_bytes = need;
                    else { ctx->ingest_stage = nullptr; (void)hipGetLastError(); }
                    for (auto &sl : ctx->ingest_slots) { sl.d = nullptr; sl.dbytes = 0; }
                }
                // (slots are re-ordered below: areas are handed out afresh every call, nothing is in flight between calls)
            }
            // usable slots first
            auto usable = [&](const ghip_ctx::pinned_slot &sl) { return sl.p && sl.ev && sl.bytes >= host_slot_bytes; };
            std::stable_partition(ctx->ingest_slots.begin(), ctx->ingest_slots.end(), usable);
            for (auto &sl : ctx->ingest_slots) if (usable(sl)) n_slots++;
            n_slots = std::min(n_slots, want);
            if (!ctx->ingest_stage) n_slots = 0;   // no staging memory: every file takes the plain blocking path
            for (size_t x = 0; x < n_slots; x++) { ctx->ingest_slots[x].d = ctx->ingest_stage + x * stage_bytes; ctx->ingest_slots[x].dbytes = stage_bytes; }
        }
        const double w_pin = since(w0) - w_hint - w_alloc;
        std::mutex smu;
        std::condition_variable scv;
        // a slot for the calling thread: a free one, else the first one in flight (after its copy has finished)
        auto take_slot = [&]() -> ghip_ctx::pinned_slot * {
            std::unique_lock<std::mutex> l(smu);
            for (;;) {
                ghip_ctx::pinned_slot *inflight = nullptr;   // the copy queued longest ago finishes first
                for (size_t x = 0; x < n_slots; x++) {
                    ghip_ctx::pinned_slot &sl = ctx->ingest_slots[x];
                    if (sl.state == 0) { sl.state = 1; return &sl; }
                    if (sl.state == 2 && (!inflight || sl.seq < inflight->seq)) inflight = &sl;
                }
                if (inflight) {
                    inflight->state = 1;   // mine; nobody else waits for it
                    l.unlock();
                    if (hipEventSynchronize(inflight->ev) != hipSuccess) { (void)hipGetLastError(); }
                    return inflight;
                }
                scv.wait(l);   // every slot is being parsed into: wait for one to be handed back
            }
        };
        uint64_t slot_seq = 0;
        auto give_slot = [&](ghip_ctx::pinned_slot *sl, int state) {
            { std::lock_guard<std::mutex> l(smu); sl->state = state; sl->seq = ++slot_seq; }
            scv.notify_one();
        };
        // work units: a file, or a run of consecutive SMALL files shipped as one group (packed form only)
        std::vector<std::pair<size_t, size_t>> units;
        {
            constexpr uint64_t SMALL_FILE = 256u << 10, GROUP_BASES = 4u << 20;
            constexpr size_t GROUP_FILES = 512;
            const bool grouping = packed_mode && n_slots && opt.ingest_groups;
            for (size_t i = 0; i < n;) {
                size_t j = i + 1;
                if (grouping && cap[i] <= SMALL_FILE) {
                    uint64_t bases = next_genome_offset(0, cap[i]);
                    while (j < n && j - i < GROUP_FILES && cap[j] <= SMALL_FILE && bases + next_genome_offset(0, cap[j]) <= std::min<uint64_t>(GROUP_BASES, slot_bytes)) {
                        bases += next_genome_offset(0, cap[j]);
                        j++;
                    }
                }
                units.push_back({i, j});
                i = j;
            }
        }
        auto worker = [&](int me) {
            if (hipSetDevice(ctx->device) != hipSuccess) { fail(GHIP_EHIP, "hipSetDevice failed in an ingest thread"); return; }
            hipStream_t cs = ctx->copy_stream[me % ctx->n_copy_streams];
            std::vector<uint8_t> &raw = ctx->io.raw[me];
            std::vector<uint8_t> heap_buf;
            // one file: parse, ship, build its validity bits.  Returns true when the worker must stop.
            auto ingest_one = [&](const size_t i) -> bool {
                const auto t0 = std::chrono::steady_clock::now();
                if (!ghip_slurp(paths[i], raw)) { fail(GHIP_EIO, std::string("Failed to open fasta file ") + paths[i]); return true; }
                const auto t1 = std::chrono::steady_clock::now();
                size_t len = 0;
                std::string e;
                const bool pack_this = packed_mode && n_slots && cap[i] + 64 <= slot_bytes;
                bool stop = false, inflight = false;
                int r = GHIP_OK;
                ghip_ctx::pinned_slot *slot = nullptr;
                std::chrono::steady_clock::time_point t1b = t1, t2 = t1;
                bool done = false;   // this file has been shipped (or failed) by the packed path
                if (pack_this) {
                    // Parse and pack in one pass, straight into a pinned slot (ingest.cpp: ghip_parse_fasta_packed): the
                    // normalised bytes never leave the L1.  A stream that outgrows its hint or the slot's run table comes
                    // back "does not fit" and takes the plain path below.
                    slot = take_slot();
                    t1b = std::chrono::steady_clock::now();
                    size_t used = 0, runs_off = 0;
                    uint32_t n_runs = 0;
                    bool fit = false;
                    // (the run table is bounded by the slot's device staging area: a stream with more runs "does not fit")
                    const size_t table_at = (((size_t)cap[i] + 3) / 4 + 15) / 16 * 16;
                    r = ghip_parse_fasta_packed(raw.data(), raw.size(), paths[i], slot->p, std::min(slot->bytes, table_at + slot->dbytes) - 8, (size_t)cap[i],
                                                &len, g->stats[i], e, &used, &runs_off, &n_runs, &fit);   // (- 8: the one-entry genome table goes behind the runs)
                    t2 = std::chrono::steady_clock::now();
                    if (r != GHIP_OK) { fail(r, e); stop = true; done = true; }
                    else if (fit) {
                        g->lens[i] = len;
                        hipError_t ce = hipSuccess;
                        if (len) {
                            // the codes to their place (whole 16-byte groups: the slot is zero-padded, the genome's room is longer)
                            ce = hipMemcpyAsync(reinterpret_cast<uint8_t *>(g->d_packed) + g->starts[i] / 4, slot->p, ((len + 3) / 4 + 15) / 16 * 16, hipMemcpyHostToDevice, cs);
                            uint32_t gtab[2] = {0u, (uint32_t)len};   // (len < 2^32: the packed parser refuses longer streams)
                            memcpy(slot->p + runs_off + (size_t)12 * n_runs, gtab, 8);
                            if (ce == hipSuccess) ce = hipMemcpyAsync(slot->d, slot->p + runs_off, (size_t)12 * n_runs + 8, hipMemcpyHostToDevice, cs);
                            if (ce == hipSuccess) {
                                ghip_launch_valid_from_runs(cs, reinterpret_cast<const uint32_t *>(slot->d) + 3 * (size_t)n_runs, 1, len,
                                                            reinterpret_cast<const uint32_t *>(slot->d), n_runs, g->d_valid + g->starts[i] / 32, true);
                                ce = hipGetLastError();   // (per thread: the two launches just made)
                                if (ce == hipSuccess) ce = hipEventRecord(slot->ev, cs);
                                inflight = true;
                            }
                        }
                        if (ce != hipSuccess) { fail(GHIP_EHIP, "ingest copy failed"); stop = true; }
                        done = true;
                    } else {   // goes as it is
                        give_slot(slot, 0);
                        slot = nullptr;
                    }
                }
                if (!done) {
                    slot = (!packed_mode && n_slots && cap[i] + 64 <= slot_bytes) ? take_slot() : nullptr;
                    uint8_t *stream_buf;
                    if (slot) stream_buf = slot->p;
                    else {
                        std::vector<uint8_t> &hb = packed_mode ? ctx->io.ascii[me] : heap_buf;
                        if (hb.size() < (size_t)cap[i] + 64) hb.resize((size_t)cap[i] + 64);
                        stream_buf = hb.data();
                    }
                    t1b = std::chrono::steady_clock::now();
                    r = ghip_parse_fasta(raw.data(), raw.size(), paths[i], stream_buf, (size_t)cap[i], &len, g->stats[i], e);
                    t2 = std::chrono::steady_clock::now();
                    if (r != GHIP_OK) { fail(r, e); stop = true; }
                    else if (len > cap[i]) { over = true; stop = true; }  // capacity hint too small (multi-member gzip): two-phase form
                    else {
                        g->lens[i] = len;
                        hipError_t ce = hipSuccess;
                        if (len && slot) {   // stream bytes to the slot's device staging area, packed into place behind the copy
                            ce = hipMemcpyAsync(slot->d, stream_buf, len, hipMemcpyHostToDevice, cs);
                            if (ce == hipSuccess) {
                                ghip_launch_pack_bases(cs, slot->d, len, g->starts[i], g->d_packed, g->d_valid);
                                ce = hipGetLastError();
                                if (ce == hipSuccess) ce = hipEventRecord(slot->ev, cs);
                                inflight = true;
                            }
                        } else if (len) {    // no slot (a file larger than the slots, or no pinned memory): a staging block of its own
                            uint8_t *d_tmp = (uint8_t *)ghip_pool_alloc(ctx, len + 64);
                            if (!d_tmp) ce = hipErrorOutOfMemory;
                            else {
                                ce = hipMemcpy(d_tmp, stream_buf, len, hipMemcpyHostToDevice);
                                if (ce == hipSuccess) {
                                    ghip_launch_pack_bases(cs, d_tmp, len, g->starts[i], g->d_packed, g->d_valid);
                                    ce = hipStreamSynchronize(cs);
                                }
                                ghip_pool_free(ctx, d_tmp);
                            }
                        }
                        if (ce != hipSuccess) { fail(GHIP_EHIP, "ingest copy failed"); stop = true; }
                    }
                }
                t_read += std::chrono::duration<double>(t1 - t0).count();
                t_parse += std::chrono::duration<double>(t2 - t1b).count();
                if (slot) give_slot(slot, inflight ? 2 : 0);
                if (stop) return true;
                t_alloc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count() + std::chrono::duration<double>(t1b - t1).count();
                return false;
            };
            // a GROUP of small files (consecutive in the list, hence in the device layout): every member parsed and packed
            // into ONE slot image laid out exactly like the group's region of the device arrays, shipped with one copy, its
            // members' validity bits built by one pair of launches -- a file of a few kilobases otherwise costs the copy
            // streams five operations of its own (100 000 contigs: 1.1 s of a 1.8 s run).  A member that does not fit the
            // packed form (too many runs, outgrown hint) sends the whole group through the one-file path.
            auto ingest_group = [&](const size_t first, const size_t last) -> bool {
                const uint64_t base = g->starts[first], end = next_genome_offset(g->starts[last - 1], cap[last - 1]);
                const size_t codes_bytes = (size_t)((end - base) / 4), m = last - first;
                ghip_ctx::pinned_slot *slot = take_slot();
                uint32_t *gtab = reinterpret_cast<uint32_t *>(slot->p + codes_bytes), *runs = gtab + 2 * m;
                const size_t run_cap = std::min((slot->bytes - codes_bytes) / 4, slot->dbytes / 4) > 2 * m ? (std::min((slot->bytes - codes_bytes) / 4, slot->dbytes / 4) - 2 * m) / 3 : 0;
                std::vector<uint8_t> &scratch = ctx->io.ascii[me];
                size_t n_runs_all = 0;
                uint64_t max_len = 0;
                bool ok = true, stop = false;
                for (size_t i = first; i < last && ok; i++) {
                    const auto t0 = std::chrono::steady_clock::now();
                    if (!ghip_slurp(paths[i], raw)) { fail(GHIP_EIO, std::string("Failed to open fasta file ") + paths[i]); stop = true; break; }
                    const auto t1 = std::chrono::steady_clock::now();
                    const size_t need = (((size_t)cap[i] + 3) / 4 + 15) / 16 * 16 + 12 * 1024 + 64;   // codes + room for 1 024 runs
                    if (scratch.size() < need) scratch.resize(need);
                    size_t len = 0, used = 0, runs_off = 0;
                    uint32_t n_runs = 0;
                    bool fit = false;
                    std::string e;
                    const int r = ghip_parse_fasta_packed(raw.data(), raw.size(), paths[i], scratch.data(), need, (size_t)cap[i], &len, g->stats[i], e, &used, &runs_off, &n_runs, &fit);
                    const auto t2 = std::chrono::steady_clock::now();
                    t_read += std::chrono::duration<double>(t1 - t0).count();
                    t_parse += std::chrono::duration<double>(t2 - t1).count();
                    if (r != GHIP_OK) { fail(r, e); stop = true; break; }
                    if (!fit || n_runs_all + n_runs > run_cap) { ok = false; break; }
                    g->lens[i] = len;
                    max_len = std::max<uint64_t>(max_len, len);
                    memcpy(slot->p + (size_t)((g->starts[i] - base) / 4), scratch.data(), ((len + 3) / 4 + 15) / 16 * 16);
                    gtab[2 * (i - first)] = (uint32_t)((g->starts[i] - base) / 32);
                    gtab[2 * (i - first) + 1] = (uint32_t)len;
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(scratch.data() + runs_off);
                    for (uint32_t x = 0; x < n_runs; x++) {
                        runs[3 * (n_runs_all + x)] = src[3 * x]; runs[3 * (n_runs_all + x) + 1] = src[3 * x + 1]; runs[3 * (n_runs_all + x) + 2] = (uint32_t)(i - first);
                    }
                    n_runs_all += n_runs;
                }
                if (stop || !ok) {
                    give_slot(slot, 0);
                    if (stop) return true;
                    for (size_t i = first; i < last; i++) if (ingest_one(i)) return true;   // the careful way, file by file
                    return false;
                }
                const auto t3 = std::chrono::steady_clock::now();
                hipError_t ce = hipMemcpyAsync(reinterpret_cast<uint8_t *>(g->d_packed) + base / 4, slot->p, codes_bytes, hipMemcpyHostToDevice, cs);
                if (ce == hipSuccess) ce = hipMemcpyAsync(slot->d, gtab, (2 * m + 3 * n_runs_all) * sizeof(uint32_t), hipMemcpyHostToDevice, cs);
                if (ce == hipSuccess) {
                    ghip_launch_valid_from_runs(cs, reinterpret_cast<const uint32_t *>(slot->d), (uint32_t)m, max_len,
                                                reinterpret_cast<const uint32_t *>(slot->d) + 2 * m, (uint32_t)n_runs_all, g->d_valid + base / 32, false);
                    ce = hipGetLastError();
                    if (ce == hipSuccess) ce = hipEventRecord(slot->ev, cs);
                }
                give_slot(slot, ce == hipSuccess ? 2 : 0);
                t_alloc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t3).count();
                if (ce != hipSuccess) { fail(GHIP_EHIP, "ingest copy failed"); return true; }
                return false;
            };
            for (;;) {
                if (status.load() != GHIP_OK || over.load()) break;
                const size_t u = next.fetch_add(1);
                if (u >= units.size()) break;
                const size_t first = units[u].first, last = units[u].second;
                if (last - first > 1 ? ingest_group(first, last) : ingest_one(first)) break;
            }
        };
        ctx->io.run(threads, worker);
        // every queued copy has landed before the buffers are handed to the next call and the genomes to the kernels
        for (hipStream_t cs : ctx->copy_stream)
            if (cs && hipStreamSynchronize(cs) != hipSuccess) fail(GHIP_EHIP, "ingest copy failed");
        if (dbg)
            fprintf(stderr, "[ingest] %d threads, %d pinned slots, thread-seconds: read %.3f parse %.3f copy/wait %.3f; wall: hints %.3f alloc+fill %.3f pinned setup %.3f workers %.3f\n",
                    threads, (int)n_slots, t_read.load(), t_parse.load(), t_alloc.load(), w_hint, w_alloc, w_pin, since(w0) - w_hint - w_alloc - w_pin);
        overflow = over.load();
        rc = status.load();
        if (rc != GHIP_OK) ghip_set_error(ctx, rc, err);
        if (rc == GHIP_OK && !overflow) {
            g->total_bases = 0;
            for (uint64_t l : g->lens) g->total_bases += l;
            if ((rc = dmalloc(ctx, &g->d_starts, n)) == GHIP_OK && (rc = dmalloc(ctx, &g->d_lens, n)) == GHIP_OK &&
                (rc = h2d_on(ctx, fill_stream, g->d_starts, g->starts.data(), n)) == GHIP_OK &&
                (rc = h2d_on(ctx, fill_stream, g->d_lens, g->lens.data(), n)) == GHIP_OK)
                rc = build_work(ctx, g, fill_stream);
        }
        if (rc != GHIP_OK || overflow) { free_genomes_locked(g); if (rc != GHIP_OK) return rc; }
        else { *out = g; return GHIP_OK; }
    }
    return genomes_from_files_two_phase(ctx, paths, n, io_threads, out);  // overflow: exact lengths first
}

extern "C" int ghip_genomes_synthetic_range(ghip_ctx *ctx, uint64_t seed, uint32_t members, uint64_t first,
                                            uint64_t count, uint64_t length, double sub_rate, ghip_genomes **out);

extern "C" int ghip_genomes_synthetic(ghip_ctx *ctx, uint64_t seed, uint32_t n_species, uint32_t members,
                                      uint64_t length, double sub_rate, ghip_genomes **out) {
    if (n_species == 0) return GHIP_EINVAL;
    return ghip_genomes_synthetic_range(ctx, seed, members, 0, (uint64_t)n_species * members, length, sub_rate, out);
}

extern "C" int ghip_genomes_synthetic_range(ghip_ctx *ctx, uint64_t seed, uint32_t members, uint64_t first,
                                            uint64_t count, uint64_t length, double sub_rate, ghip_genomes **out) {
    if (!ctx || !out || members == 0 || length == 0) return GHIP_EINVAL;
    if (count > 65535) return ghip_set_error(ctx, GHIP_EINVAL, "at most 65535 synthetic genomes per call");
    if (!(sub_rate >= 0.0 && sub_rate < 1.0)) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> lens((size_t)count, length);
    ghip_genomes *g = new ghip_genomes();
    g->ctx = ctx;
    ctx->live_handles++;
    int rc = layout_genomes(ctx, g, lens);
    if (rc != GHIP_OK) { free_genomes_locked(g); return rc; }
    if (count) ghip_launch_synth(ctx, g->d_packed, g->d_valid, g->d_starts, length, (uint32_t)first, (uint32_t)count, members, seed, (uint32_t)(sub_rate * 4294967296.0));
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { free_genomes_locked(g); return ghip_set_error(ctx, GHIP_EHIP, "synth_genomes failed"); }
    *out = g;
    return GHIP_OK;
}

extern "C" size_t ghip_genomes_count(const ghip_genomes *g) { return g ? g->n : 0; }
extern "C" uint64_t ghip_genomes_total_bases(const ghip_genomes *g) { return g ? g->total_bases : 0; }
extern "C" uint64_t ghip_genomes_length(const ghip_genomes *g, size_t idx) { return (g && idx < g->n) ? g->lens[idx] : 0; }

extern "C" int ghip_genomes_stats(const ghip_genomes *g, size_t idx, uint64_t *num_contigs, uint64_t *num_ambiguous_bases,
                                  uint64_t *n50) {
    if (!g || idx >= g->n) return GHIP_EINVAL;
    if (g->stats.size() != g->n) return GHIP_EUNSUPPORTED;  // only genomes read from FASTA files carry statistics
    if (num_contigs) *num_contigs = g->stats[idx].num_contigs;
    if (num_ambiguous_bases) *num_ambiguous_bases = g->stats[idx].num_ambiguous_bases;
    if (n50) *n50 = g->stats[idx].n50;
    return GHIP_OK;
}

extern "C" int ghip_genomes_to_host(ghip_ctx *ctx, const ghip_genomes *g, size_t idx, uint8_t *outp) {
    if (!ctx || !g || idx >= g->n || !outp) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t len = g->lens[idx];
    if (len == 0) return GHIP_OK;
    uint8_t *d_tmp = nullptr;   // the stream as bytes ('N' at every invalid position)
    int rc = dmalloc(ctx, &d_tmp, (len + 15) / 16 * 16);
    if (rc) return rc;
    ghip_launch_unpack_bases(ctx->stream, g->d_packed, g->d_valid, g->starts[idx], d_tmp, len);
    rc = d2h(ctx, outp, d_tmp, len);
    ghip_pool_free(ctx, d_tmp);
    return rc;
}

// ------------------------------------------------------------------------------------ sketches
static void free_sketches_locked(ghip_sketches *sk) {  // ctx->mu held
    ghip_ctx *ctx = sk->ctx;
    if (sk->owned) { ghip_pool_free(ctx, sk->d_hashes); ghip_pool_free(ctx, sk->d_lens); }
    ghip_pool_free(ctx, sk->d_tables); ghip_pool_free(ctx, sk->d_tags); ghip_pool_free(ctx, sk->d_row_start); ghip_pool_free(ctx, sk->d_arranged);
    ctx->live_handles--;
    delete sk;
}

extern "C" void ghip_sketches_free(ghip_sketches *sk) {
    if (!sk) return;
    ghip_ctx *ctx = sk->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStreamSynchronize(ctx->stream);
        free_sketches_locked(sk);
    }
    ctx_release(ctx);
}

extern "C" size_t ghip_sketches_count(const ghip_sketches *sk) { return sk ? sk->n : 0; }
extern "C" uint32_t ghip_sketches_size(const ghip_sketches *sk) { return sk ? sk->s : 0; }
extern "C" uint32_t ghip_sketches_kmer(const ghip_sketches *sk) { return sk ? sk->k : 0; }
extern "C" void *ghip_sketches_device_hashes(const ghip_sketches *sk) { return sk ? sk->d_hashes : nullptr; }
extern "C" void *ghip_sketches_device_lens(const ghip_sketches *sk) { return sk ? sk->d_lens : nullptr; }

extern "C" int ghip_sketches_from_host(ghip_ctx *ctx, const uint64_t *hashes, const uint32_t *lens, size_t n,
                                       uint32_t s, uint32_t k, ghip_sketches **out) {
    if (!ctx || !out || s == 0 || (n && (!hashes || !lens))) return GHIP_EINVAL;
    for (size_t i = 0; i < n; i++) {
        if (lens[i] > s) return ghip_set_error(ctx, GHIP_EINVAL, "sketch length exceeds sketch size");
        for (uint32_t e = 1; e < lens[i]; e++)
            if (hashes[i * s + e] <= hashes[i * s + e - 1]) return ghip_set_error(ctx, GHIP_EINVAL, "sketch hashes must be strictly ascending");
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = n; sk->s = s; sk->k = k;
    std::vector<uint64_t> padded(hashes, hashes + n * (size_t)s);
    for (size_t i = 0; i < n; i++) for (uint32_t e = lens[i]; e < s; e++) padded[i * s + e] = ~0ull;
    int rc = dmalloc(ctx, &sk->d_hashes, n * (size_t)s);
    if (!rc) rc = dmalloc(ctx, &sk->d_lens, n);
    if (!rc) rc = h2d(ctx, sk->d_hashes, padded.data(), n * (size_t)s);
    if (!rc) rc = h2d(ctx, sk->d_lens, lens, n);
    if (rc) { free_sketches_locked(sk); return rc; }
    *out = sk;
    return GHIP_OK;
}

extern "C" int ghip_sketches_wrap_device(ghip_ctx *ctx, void *d_hashes, void *d_lens, size_t n, uint32_t s,
                                         uint32_t k, ghip_sketches **out) {
    if (!ctx || !out || s == 0 || (n && (!d_hashes || !d_lens))) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = n; sk->s = s; sk->k = k;
    sk->d_hashes = (uint64_t *)d_hashes; sk->d_lens = (uint32_t *)d_lens; sk->owned = false;
    *out = sk;
    return GHIP_OK;
}

extern "C" int ghip_sketches_to_host(ghip_ctx *ctx, const ghip_sketches *sk, uint64_t *hashes, uint32_t *lens) {
    if (!ctx || !sk) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = GHIP_OK;
    if (hashes) rc = d2h(ctx, hashes, sk->d_hashes, sk->n * (size_t)sk->s);
    if (!rc && lens) rc = d2h(ctx, lens, sk->d_lens, sk->n);
    return rc;
}

extern "C" int ghip_sketches_copy_into(ghip_ctx *ctx, const ghip_sketches *sk, void *d_hashes_dst, void *d_lens_dst) {
    if (!ctx || !sk || !d_hashes_dst || !d_lens_dst) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sk->n) {
        GHIP_HIP_CHECK(ctx, hipMemcpyAsync(d_hashes_dst, sk->d_hashes, sk->n * (size_t)sk->s * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        GHIP_HIP_CHECK(ctx, hipMemcpyAsync(d_lens_dst, sk->d_lens, sk->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return GHIP_OK;
}

// On-disk sketch matrix (the reference has no finch counterpart -- skani's --low-memory db dir,
// src/skani.rs:266-304, and the sketched reference set of its --reference-genomes mode, src/skani.rs:502-565, are the
// closest).  "GHIPSK02", little-endian:
//   char[8] magic; u32 k; u32 s; u64 hash seed; u64 n; u64 names_bytes; u32 len[n]; u64 hashes[n][s];
//   char names[names_bytes] (n NUL-terminated genome names, in row order); u64 FNV-1a-64 of every byte before it.
// "GHIPSK01" (round 1-2: no seed, no names, no checksum) still loads.
namespace {
struct Fnv {
    uint64_t h = 0xcbf29ce484222325ull;
    void add(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; } }
};
bool put(FILE *f, Fnv &c, const void *p, size_t n) { c.add(p, n); return n == 0 || fwrite(p, 1, n, f) == n; }
bool get(FILE *f, Fnv &c, void *p, size_t n) { if (n && fread(p, 1, n, f) != n) return false; c.add(p, n); return true; }
}  // namespace

extern "C" int ghip_sketches_save_named(ghip_ctx *ctx, const ghip_sketches *sk, const char *const *names, uint64_t seed, const char *path) {
    if (!ctx || !sk || !path) return GHIP_EINVAL;
    std::vector<uint64_t> h(sk->n * (size_t)sk->s);
    std::vector<uint32_t> l(sk->n);
    int rc = ghip_sketches_to_host(ctx, sk, h.data(), l.data());
    if (rc) return rc;
    std::string blob;
    for (size_t i = 0; i < sk->n; i++) { if (names && names[i]) blob += names[i]; blob.push_back('\0'); }
    FILE *f = fopen(path, "wb");
    if (!f) return ghip_set_error(ctx, GHIP_EIO, std::string("cannot write ") + path);
    const uint64_t n = sk->n, nb = blob.size();
    Fnv c;
    bool ok = put(f, c, "GHIPSK02", 8) && put(f, c, &sk->k, 4) && put(f, c, &sk->s, 4) && put(f, c, &seed, 8) && put(f, c, &n, 8) &&
              put(f, c, &nb, 8) && put(f, c, l.data(), 4 * l.size()) && put(f, c, h.data(), 8 * h.size()) && put(f, c, blob.data(), blob.size());
    ok = ok && fwrite(&c.h, 8, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    return ok ? GHIP_OK : ghip_set_error(ctx, GHIP_EIO, std::string("short write to ") + path);
}

extern "C" int ghip_sketches_save(ghip_ctx *ctx, const ghip_sketches *sk, const char *path) {
    return ghip_sketches_save_named(ctx, sk, nullptr, 0, path);
}

extern "C" int ghip_sketches_load_named(ghip_ctx *ctx, const char *path, ghip_sketches **out, char **out_names, size_t *out_names_bytes, uint64_t *out_seed) {
    if (!ctx || !path || !out) return GHIP_EINVAL;
    if (out_names) *out_names = nullptr;
    if (out_names_bytes) *out_names_bytes = 0;
    if (out_seed) *out_seed = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return ghip_set_error(ctx, GHIP_EIO, std::string("cannot read ") + path);
    char magic[8];
    uint32_t k = 0, s = 0;
    uint64_t n = 0, seed = 0, nb = 0;
    Fnv c;
    bool ok = get(f, c, magic, 8);
    const bool v2 = ok && !memcmp(magic, "GHIPSK02", 8);
    ok = ok && (v2 || !memcmp(magic, "GHIPSK01", 8)) && get(f, c, &k, 4) && get(f, c, &s, 4);
    if (ok && v2) ok = get(f, c, &seed, 8);
    ok = ok && get(f, c, &n, 8);
    if (ok && v2) ok = get(f, c, &nb, 8);
    ok = ok && s >= 1 && s <= GHIP_MAX_SKETCH_SIZE && n < (1ull << 32) && nb < (1ull << 40);
    std::vector<uint32_t> l;
    std::vector<uint64_t> h;
    std::string blob;
    std::string why = "not a sketch matrix file: ";
    if (ok) {
        // the header's counts are believed only when the FILE is exactly as long as they say (a truncated or
        // corrupt header could ask for terabytes before the checksum was ever looked at), and an allocation that still fails
        // is an error code, not an exception through the C boundary
        struct stat st;
        const uint64_t header = 8 + 4 + 4 + (v2 ? 8 : 0) + 8 + (v2 ? 8 : 0);
        const uint64_t want = header + 4 * n + 8 * n * (uint64_t)s + nb + (v2 ? 8 : 0);
        if (fstat(fileno(f), &st) != 0 || (uint64_t)st.st_size != want) { ok = false; why = "sketch matrix file is truncated or its header is damaged (size): "; }
    }
    if (ok) {
        try { l.resize(n); h.resize(n * (size_t)s); blob.resize(nb); }
        catch (const std::exception &) { fclose(f); return ghip_set_error(ctx, GHIP_ENOMEM, std::string("out of host memory loading ") + path); }
        ok = get(f, c, l.data(), 4 * l.size()) && get(f, c, h.data(), 8 * h.size()) && get(f, c, &blob[0], nb);
    }
    if (ok && v2) {
        uint64_t sum = 0;
        ok = fread(&sum, 8, 1, f) == 1 && sum == c.h;
        if (!ok) why = "sketch matrix file is damaged (checksum): ";
        else if ((size_t)std::count(blob.begin(), blob.end(), '\0') != n) { ok = false; why = "sketch matrix file: name table does not match the row count: "; }
    }
    fclose(f);
    if (!ok) return ghip_set_error(ctx, GHIP_EIO, why + path);
    const int rc = ghip_sketches_from_host(ctx, h.data(), l.data(), n, s, k, out);  // validates order and lengths
    if (rc) return rc;
    if (out_names) {
        if (!v2) blob.assign(n, '\0');   // a GHIPSK01 file holds no names
        char *p = (char *)malloc(std::max<size_t>(blob.size(), 1));
        if (!p) { ghip_sketches_free(*out); *out = nullptr; return GHIP_ENOMEM; }
        memcpy(p, blob.data(), blob.size());
        *out_names = p;
        if (out_names_bytes) *out_names_bytes = blob.size();
    }
    if (out_seed) *out_seed = seed;
    return GHIP_OK;
}

extern "C" int ghip_sketches_load(ghip_ctx *ctx, const char *path, ghip_sketches **out) {
    return ghip_sketches_load_named(ctx, path, out, nullptr, nullptr, nullptr);
}

// rows of a followed by the rows of b (same k and s): the matrix of an incremental run = the saved one + the new genomes'
extern "C" int ghip_sketches_concat(ghip_ctx *ctx, const ghip_sketches *a, const ghip_sketches *b, ghip_sketches **out) {
    if (!ctx || !a || !b || !out) return GHIP_EINVAL;
    if (a->s != b->s || a->k != b->k) return ghip_set_error(ctx, GHIP_EINVAL, "sketch matrices differ in sketch size or k-mer length");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = a->n + b->n; sk->s = a->s; sk->k = a->k;
    int rc = dmalloc(ctx, &sk->d_hashes, sk->n * (size_t)sk->s);
    if (!rc) rc = dmalloc(ctx, &sk->d_lens, sk->n);
    const ghip_sketches *parts[2] = {a, b};
    size_t at = 0;
    for (int x = 0; x < 2 && !rc; x++) {
        const ghip_sketches *p = parts[x];
        if (p->n && (hipMemcpyAsync(sk->d_hashes + at * sk->s, p->d_hashes, p->n * (size_t)sk->s * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                     hipMemcpyAsync(sk->d_lens + at, p->d_lens, p->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess))
            rc = ghip_set_error(ctx, GHIP_EHIP, "sketch concatenation failed");
        at += p->n;
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "sketch concatenation failed");
    if (rc) { free_sketches_locked(sk); return rc; }
    *out = sk;
    return GHIP_OK;
}

// finch::sketch_files replacement.  Exactness: a genome is accepted only when its candidate
// list did not overflow and held >= s distinct hashes (or the threshold was already 2^64-1).
static int sketch_genomes_locked(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s, uint64_t seed,
                                 const ghip_seed_args *seeds, ghip_sketches **out) {
    if (k < 1 || k > 32) return ghip_set_error(ctx, GHIP_EINVAL, "kmer_length must be in 1..=32");
    if (s < 1 || s > GHIP_MAX_SKETCH_SIZE) return ghip_set_error(ctx, GHIP_EINVAL, "num_kmers must be in 1..=65535");
    if (seed > 0xffffffffull) return ghip_set_error(ctx, GHIP_EINVAL, "hash_seed must fit 32 bits (murmurhash3 seed)");
    const size_t n = g->n;
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = n; sk->s = s; sk->k = k;
    int rc = dmalloc(ctx, &sk->d_hashes, n * (size_t)s);
    if (!rc) rc = dmalloc(ctx, &sk->d_lens, n);
    if (rc) { free_sketches_locked(sk); return rc; }

    // pending slots: initially every genome, with a threshold that lets ~1.5*s hashes survive (for s = 1000:
    // 1500 +- 39, so fewer than s survive only on repetitive or tiny genomes, which the retry loop widens)
    std::vector<uint32_t> slot_genome(n);
    std::vector<uint64_t> slot_thr(n);
    std::vector<uint32_t> slot_cap(n);
    const uint32_t cap0 = next_pow2(2ull * s + 64);
    for (size_t i = 0; i < n; i++) {
        slot_genome[i] = (uint32_t)i;
        uint64_t nk = g->lens[i] >= k ? g->lens[i] - k + 1 : 1;
        unsigned __int128 want = ((unsigned __int128)(3ull * s / 2 + 8)) << 64;
        unsigned __int128 t = want / nk;
        slot_thr[i] = (t >> 64) ? ~0ull : (uint64_t)t;
        slot_cap[i] = cap0;
    }
    bool first = true;
    int iter = 0;
    while (!slot_genome.empty()) {
        if (++iter > 80) { free_sketches_locked(sk); return ghip_set_error(ctx, GHIP_EHIP, "sketch selection did not converge"); }
        const size_t ns = slot_genome.size();
        std::vector<uint64_t> cand_start(ns);
        uint64_t total_cand = 0;
        for (size_t i = 0; i < ns; i++) { cand_start[i] = total_cand; total_cand += slot_cap[i]; }
        DeviceFree tmp(ctx);
        uint32_t *d_slot_genome = nullptr, *d_cap = nullptr, *d_count = nullptr, *d_status = nullptr;
        uint64_t *d_thr = nullptr, *d_cstart = nullptr, *d_cand = nullptr;
        ghip_sketch_work *d_work = nullptr;
        size_t n_work = 0;
        if ((rc = dmalloc(ctx, &d_thr, ns))) break; tmp.add(d_thr);
        if ((rc = dmalloc(ctx, &d_cstart, ns))) break; tmp.add(d_cstart);
        if ((rc = dmalloc(ctx, &d_cap, ns))) break; tmp.add(d_cap);
        if ((rc = dmalloc(ctx, &d_count, ns))) break; tmp.add(d_count);
        if ((rc = dmalloc(ctx, &d_status, ns))) break; tmp.add(d_status);
        if ((rc = dmalloc(ctx, &d_cand, total_cand))) break; tmp.add(d_cand);
        if (first) {
            d_slot_genome = g->d_identity;
            d_work = g->d_work;
            n_work = g->n_work;
        } else {
            std::vector<ghip_sketch_work> work;
            for (size_t i = 0; i < ns; i++) {
                uint64_t chunks = (g->lens[slot_genome[i]] + GHIP_SKETCH_CHUNK - 1) / GHIP_SKETCH_CHUNK;
                for (uint64_t c = 0; c < chunks; c++) work.push_back({(uint32_t)i, (uint32_t)c});
            }
            n_work = work.size();
            if ((rc = dmalloc(ctx, &d_slot_genome, ns))) break; tmp.add(d_slot_genome);
            if ((rc = dmalloc(ctx, &d_work, n_work))) break; tmp.add(d_work);
            if ((rc = h2d(ctx, d_slot_genome, slot_genome.data(), ns))) break;
            if ((rc = h2d(ctx, d_work, work.data(), n_work))) break;
        }
        if ((rc = h2d_nosync(ctx, d_thr, slot_thr.data(), ns))) break;
        if ((rc = h2d_nosync(ctx, d_cstart, cand_start.data(), ns))) break;
        if ((rc = h2d_nosync(ctx, d_cap, slot_cap.data(), ns))) break;
        if (hipMemsetAsync(d_count, 0, ns * sizeof(uint32_t), ctx->stream) != hipSuccess) { rc = ghip_set_error(ctx, GHIP_EHIP, "memset failed"); break; }
        // (no synchronisation here: the three host vectors stay as they are until the one after the kernels)
        ghip_launch_sketch_kmers(ctx, g->d_packed, g->d_valid, g->d_starts, g->d_lens, d_slot_genome, d_thr, d_cstart, d_cap,
                                 d_work, n_work, k, (uint32_t)seed, d_cand, d_count, first ? seeds : nullptr);
        ghip_launch_sketch_select(ctx, d_slot_genome, ns, d_cand, d_count, d_cstart, d_cap, *std::max_element(slot_cap.begin(), slot_cap.end()), s, sk->d_hashes,
                                  sk->d_lens, d_status);
        std::vector<uint32_t> status(ns), count(ns);
        if ((rc = d2h_nosync(ctx, status.data(), d_status, ns))) break;
        if ((rc = d2h_nosync(ctx, count.data(), d_count, ns))) break;
        if ((rc = stream_sync(ctx))) break;
        { hipError_t e = hipGetLastError(); if (e != hipSuccess) { rc = ghip_set_error(ctx, GHIP_EHIP, std::string("sketch kernels: ") + hipGetErrorString(e)); break; } }
        std::vector<uint32_t> ng; std::vector<uint64_t> nthr; std::vector<uint32_t> ncap;
        for (size_t i = 0; i < ns; i++) {
            if (status[i] & 1u) {  // overflow: same threshold, list as large as the survivor count
                ng.push_back(slot_genome[i]); nthr.push_back(slot_thr[i]); ncap.push_back(next_pow2(count[i]));
            } else if ((status[i] & 2u) && slot_thr[i] != ~0ull) {  // too few distinct: widen 8x
                uint64_t t = slot_thr[i];
                ng.push_back(slot_genome[i]);
                nthr.push_back(t > (~0ull >> 3) ? ~0ull : t << 3);
                ncap.push_back(std::max<uint32_t>(slot_cap[i], next_pow2(8ull * count[i] + 64)));
            }
        }
        slot_genome.swap(ng); slot_thr.swap(nthr); slot_cap.swap(ncap);
        first = false;
    }
    if (rc) { free_sketches_locked(sk); return rc; }
    *out = sk;
    return GHIP_OK;
}

extern "C" int ghip_sketch_genomes(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s, uint64_t seed,
                                   ghip_sketches **out) {
    if (!ctx || !g || !out) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return sketch_genomes_locked(ctx, g, k, s, seed, nullptr, out);
}

extern "C" int ghip_sketch_and_index_files(ghip_ctx *ctx, const char *const *paths, size_t n, uint32_t k, uint32_t s, uint64_t seed,
                                           uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, int io_threads, uint64_t batch_bytes,
                                           ghip_sketches **out_sk, ghip_ani_index **out_idx, uint64_t *out_stats);

// finch::sketch_files for a file list (src/finch.rs:55-69); inputs larger than HBM are sketched in batches
extern "C" int ghip_sketch_files(ghip_ctx *ctx, const char *const *paths, size_t n, uint32_t k, uint32_t s,
                                 uint64_t seed, int io_threads, ghip_sketches **out) {
    return ghip_sketch_and_index_files(ctx, paths, n, k, s, seed, 0, 0, 0, io_threads, 0, out, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------ precluster
extern "C" uint64_t ghip_last_pairs_compared(const ghip_ctx *ctx) { return ctx ? ctx->last_pairs : 0; }

// Builds the cuckoo sets and the work rows of the probe-form pair kernel once per sketch matrix.
static int prepare_probe(ghip_ctx *ctx, ghip_sketches *sk) {
    if (sk->probe_ready) return GHIP_OK;
    const size_t slots = ghip_probe_table_slots(sk->s);
    int rc;
    uint32_t *d_flags = nullptr;
    if ((rc = dmalloc(ctx, &sk->d_tables, sk->n * slots))) return rc;
    if ((rc = dmalloc(ctx, &sk->d_tags, sk->n * slots))) return rc;
    if ((rc = dmalloc(ctx, &d_flags, 1))) return rc;
    DeviceFree tmp(ctx); tmp.add(d_flags);
    // the arranged form (ghip_options.probe_arranged, fixed for the life of the matrix's tables): constrained second bucket,
    // B rows dealt to the lanes by bucket residue
    if (ctx->opt.probe_arranged && (rc = dmalloc(ctx, &sk->d_arranged, sk->n * ghip_probe_arranged_slots(sk->s)))) return rc;
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_flags, 0, sizeof(uint32_t), ctx->stream));
    ghip_launch_pair_tables(ctx, sk->d_hashes, sk->d_lens, sk->n, sk->s, sk->d_tables, sk->d_tags, d_flags, sk->d_arranged);
    sk->n_work = ghip_probe_work_rows(sk->n, ctx->num_cus, &sk->probe_cb, sk->row_start);
    if ((rc = dmalloc(ctx, &sk->d_row_start, sk->row_start.size()))) return rc;
    if ((rc = h2d(ctx, sk->d_row_start, sk->row_start.data(), sk->row_start.size()))) return rc;
    if ((rc = d2h(ctx, &sk->probe_flags, d_flags, 1))) return rc;
    sk->probe_ready = true;
    return GHIP_OK;
}

// replicate_join: a multi-rank caller that prefers the whole list on every rank whenever the join form runs (its
// cost, one pass over all N*s hashes, does not shard) to a share it then has to exchange.
// cmin[total] = smallest common whose reference ANI clears the threshold (src/finch.rs:91:
// `distance >= min_ani as f64`).  The device filter only has to be a superset; the exact
// test is repeated on the host for every emitted pair.  Cached in the context per (min_ani, s, k); ctx->mu held.
int ghip_pair_filter_prepare(ghip_ctx *ctx, uint32_t s, uint32_t k, float min_ani) {
    const double thr = (double)min_ani;
    uint32_t ani_bits;
    memcpy(&ani_bits, &min_ani, 4);
    if (ctx->cmin.valid && ctx->cmin.ani_bits == ani_bits && ctx->cmin.s == s && ctx->cmin.k == k) return GHIP_OK;
    const uint32_t max_total = 2 * s;
    std::vector<uint16_t> cmin(max_total + 2, 0xffff);
    for (uint32_t total = 0; total <= max_total; total++) {
        // finch_ani is non-decreasing in common for a fixed total: the smallest passing common by bisection (s = 10 000
        // would cost 2e8 logarithms the linear way); total = 0 is the NaN corner (ANI 1.0 whatever common is)
        const uint32_t cmax = std::min(total, s);
        if (!(finch_ani(cmax, total, k) >= thr)) continue;   // nothing passes: 0xffff
        uint32_t lo = 0, hi = cmax;                            // invariant: hi passes
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (finch_ani(mid, total, k) >= thr) hi = mid; else lo = mid + 1; }
        cmin[total] = (uint16_t)hi;
    }
    if (ctx->cmin.d_cmin) ghip_pool_free(ctx, ctx->cmin.d_cmin);
    ctx->cmin.valid = false;
    int rc;
    if ((rc = dmalloc(ctx, &ctx->cmin.d_cmin, cmin.size()))) return rc;
    if ((rc = h2d(ctx, ctx->cmin.d_cmin, cmin.data(), cmin.size()))) return rc;
    ctx->cmin.ani_bits = ani_bits; ctx->cmin.s = s; ctx->cmin.k = k; ctx->cmin.valid = true;
    ctx->cmin.floor = ghip_cmin_floor(cmin);
    return GHIP_OK;
}

// The host end of the pair stage: exact reference arithmetic (f64 ANI, threshold, `as f32`) on every candidate the device
// filter let through, then (i, j) order.  filter_share: keep only the pairs with (i + j) mod world == rank.
// (serial on purpose: one f64 log per emitted pair is 1 ms per 45 000 pairs, and spawning threads on the 256-core
// host cost more than that -- measured 4.7 -> 6.6 ms for the stage at 10 000 genomes)
int ghip_pairs_finalize(ghip_ctx *ctx, std::vector<ghip_pair> &host, uint32_t k, float min_ani, size_t n, bool filter_share,
                        uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n) {
    const double thr = (double)min_ani;
    size_t m = 0;
    // finch_ani is a pure function of (common, total, k): its f64 results are kept per context in a table indexed by
    // (total, common) -- one logarithm per distinct value instead of one per candidate and call (45 000 candidates cost
    // 0.9 ms at 10 000 genomes, every step; the values a run meets cluster in a few hundred KB of the table).  ctx->mu held.
    const uint32_t s_max = ctx && ctx->cmin.valid ? ctx->cmin.s : 0;
    const bool memo = ctx && s_max >= 1 && s_max <= 2048;
    if (memo && (ctx->ani_memo_k != k || ctx->ani_memo_s != s_max)) {
        ctx->ani_memo.assign((size_t)(2 * s_max + 1) * (s_max + 1), std::nan(""));
        ctx->ani_memo_k = k; ctx->ani_memo_s = s_max;
    }
    if (host.size() >= 200000 && ctx && ctx->ingest_mu.try_lock()) {
        // long lists (a large collection, a very large family): the f64 values on the context's workers, each over its own
        // range and without the table (its slots are written on first use: not from several threads); -1.0f marks a pair
        // that fails (an ANI is never negative)
        const size_t total_n = host.size(), workers = std::min<size_t>(16, total_n / 50000), per = (total_n + workers - 1) / workers;
        ctx->io.run((int)workers, [&](int w) {
            for (size_t i = std::min(total_n, (size_t)w * per), e = std::min(total_n, ((size_t)w + 1) * per); i < e; i++) {
                host[i].ani = -1.0f;
                if (filter_share && (host[i].i + host[i].j) % world != rank) continue;
                const double ani = finch_ani(host[i].common, host[i].total, k);
                if (ani >= thr) host[i].ani = (float)ani;
            }
        });
        ctx->ingest_mu.unlock();
        for (size_t i = 0; i < total_n; i++) if (host[i].ani != -1.0f) host[m++] = host[i];
    } else
    for (size_t i = 0; i < host.size(); i++) {
        if (filter_share && (host[i].i + host[i].j) % world != rank) continue;
        double ani;
        if (memo && host[i].total <= 2 * s_max && host[i].common <= s_max) {
            double &slot = ctx->ani_memo[(size_t)host[i].total * (s_max + 1) + host[i].common];
            if (std::isnan(slot)) slot = finch_ani(host[i].common, host[i].total, k);
            ani = slot;
        } else ani = finch_ani(host[i].common, host[i].total, k);
        if (ani >= thr) { host[i].ani = (float)ani; host[m++] = host[i]; }
    }
    host.resize(m);
    // (i, j) order: counting sort by i straight into the result (O(m + n)), then the few entries of each i by j --
    // a comparison sort of the whole list costs 0.24 ms at 4 500 hits and 3.8 ms at 45 000
    ghip_pair *res = (ghip_pair *)malloc(std::max<size_t>(m, 1) * sizeof(ghip_pair));
    if (!res) return ghip_set_error(ctx, GHIP_ENOMEM, "out of host memory");
    {
        std::vector<size_t> at(n + 1, 0);
        for (size_t x = 0; x < m; x++) at[host[x].i + 1]++;
        for (size_t g = 0; g < n; g++) at[g + 1] += at[g];
        std::vector<size_t> fill(at.begin(), at.end() - 1);
        for (size_t x = 0; x < m; x++) res[fill[host[x].i]++] = host[x];
        auto sort_rows = [&](size_t g0, size_t g1) {
            for (size_t g = g0; g < g1; g++)
                if (at[g + 1] - at[g] > 1)
                    std::sort(res + at[g], res + at[g + 1], [](const ghip_pair &a, const ghip_pair &b) { return a.j < b.j; });
        };
        if (m >= 200000 && ctx && ctx->ingest_mu.try_lock()) {   // long lists: the rows on the context's workers, equal shares of the ENTRIES
            const size_t workers = std::min<size_t>(16, m / 50000);
            std::vector<size_t> cut(workers + 1, n);
            cut[0] = 0;
            for (size_t w = 1; w < workers; w++) cut[w] = std::lower_bound(at.begin(), at.end(), m * w / workers) - at.begin();
            for (size_t w = 1; w <= workers; w++) cut[w] = std::min(std::max(cut[w], cut[w - 1]), n);
            ctx->io.run((int)workers, [&](int w) { sort_rows(cut[w], cut[w + 1]); });
            ctx->ingest_mu.unlock();
        } else sort_rows(0, n);
    }
    *out_pairs = res; *out_n = m;
    return GHIP_OK;
}

// row_lo > 0: the (new x all) rectangle of an incremental run -- only the pairs (i, j), i < j, with j >= row_lo.
// dense_share: no join; a dense pass over EVERY pair, of which the (i + j) mod world == rank share is returned (what a
// rank of the hash-sharded join owes when its own second stage had to give up).
static int precluster_impl(ghip_ctx *ctx, const ghip_sketches *sk_in, float min_ani, uint32_t rank, uint32_t world,
                           bool replicate_join, size_t row_lo, ghip_pair **out_pairs, size_t *out_n, int *out_replicated,
                           bool dense_share = false) {
    ghip_sketches *sk = const_cast<ghip_sketches *>(sk_in);  // lazily caches the probe-form tables
    if (out_replicated) *out_replicated = 0;
    if (!ctx || !sk || !out_pairs || !out_n || world == 0 || rank >= world) return GHIP_EINVAL;
    if (sk->s > GHIP_MAX_SKETCH_SIZE) return ghip_set_error(ctx, GHIP_EINVAL, "sketch size above 65535 is not supported");
    if (row_lo > sk->n || (row_lo && world > 1)) return ghip_set_error(ctx, GHIP_EINVAL, "row_lo must not exceed the sketch count (single rank only)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    *out_pairs = nullptr; *out_n = 0;
    const size_t n = sk->n;
    const uint32_t s = sk->s, k = sk->k;
    ctx->last_pairs = 0;
    if (n < 2) return GHIP_OK;

    DeviceFree tmp(ctx);
    int rc = ghip_pair_filter_prepare(ctx, s, k, min_ani);
    if (rc) return rc;
    uint16_t *d_cmin = ctx->cmin.d_cmin;

    // Three forms of the pair stage, identical results (tests/test_gpu_parity.py runs all three against the oracle):
    //   join  (pairs_join.hip)   inverted index over all N*s hashes; N >= GHIP_JOIN_MIN_N, declines dense inputs
    //   probe (pairs_probe.hip)  dense, cuckoo sets in LDS; s <= 1024
    //   merge (pairs.hip)        dense, 64-way merge path; s > 1024, or a sketch holds 2^64-1 / a cuckoo insertion failed
    // GHIP_PAIR_KERNEL=join|probe|merge forces a form (join still declines what it cannot do).
    const uint32_t force = ctx->opt.pair_form;
    // (sketches too long for LDS tiles, s > 4096, go to the join whatever n is: the dense form left for them reads global memory)
    const bool want_join = !dense_share && (force != GHIP_PAIR_AUTO ? force == GHIP_PAIR_JOIN : (n >= GHIP_JOIN_MIN_N || s > 4096));
    bool use_probe = s <= 1024 && force != GHIP_PAIR_MERGE;
    bool probe_checked = false;

    const bool dbg_laps = ghip_dbg(ctx->opt, GHIP_DEBUG_PRECLUSTER);   // host laps of the stage on stderr
    auto lap_t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg_laps) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[precluster] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - lap_t0).count());
        lap_t0 = t;
    };
    const uint64_t P = (uint64_t)n * (n - 1) / 2;
    uint64_t cap = std::min<uint64_t>(P, std::max<uint64_t>(1u << 20, 64ull * n));
    unsigned long long *d_count = nullptr;
    if ((rc = dmalloc(ctx, &d_count, 1))) return rc;
    tmp.add(d_count);
    std::vector<ghip_pair> host;
    std::vector<uint32_t> empties;   // empty sketches of a joined run: their pairs are added on the host
    uint8_t *d_big = nullptr;        // genomes of element buckets too large for the join (their mutual pairs: dense, below)
    bool big_pending = false;
    bool filter_share = dense_share;  // dense pass over ALL pairs, this rank's (i + j) mod world share picked on the host
    bool listed = false;  // an attempt whose candidate list held every hit
    for (int attempt = 0; attempt < 4 && !listed; attempt++) {
        ghip_pair *d_out = nullptr;
        if ((rc = dmalloc(ctx, &d_out, cap))) return rc;
        DeviceFree t2(ctx); t2.add(d_out);
        GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
        uint64_t compared = 0;
        bool joined = false, late = false;
        empties.clear();
        bool has_big = false;
        if (want_join && !filter_share) {
            if (!d_big) { if ((rc = dmalloc(ctx, &d_big, n))) return rc; tmp.add(d_big); }
            GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_big, 0, n, ctx->stream));
            if ((rc = ghip_pairs_join(ctx, sk->d_hashes, sk->d_lens, n, s, d_cmin, ctx->cmin.floor,
                                      replicate_join ? 0 : rank, replicate_join ? 1 : world, (uint32_t)row_lo,
                                      d_out, d_count, cap, &compared, &joined, &late, &empties, d_big, &has_big))) return rc;
        }
        if (!joined) { empties.clear(); has_big = false; }   // a dense form enumerates every pair itself
        if (joined && replicate_join) {  // every rank holds every pair; book an equal share of the comparisons
            compared = P / world + (rank < P % world ? 1 : 0);
            if (out_replicated) *out_replicated = 1;
        }
        // The sharded join declined AFTER the point where all ranks decide alike (this rank's own records overflowed a
        // table): the other ranks deliver their (i + j) mod world shares, so this one must deliver exactly its own --
        // a dense pass over every pair, filtered on the host below.
        if (late && world > 1 && !replicate_join) filter_share = true;
        const uint32_t drank = filter_share ? 0 : rank, dworld = filter_share ? 1 : world;
        if (!joined && use_probe && !probe_checked) {
            if ((rc = prepare_probe(ctx, sk))) return rc;
            use_probe = sk->probe_flags == 0;
            probe_checked = true;
        }
        if (joined) {
        } else if (use_probe) {
            ghip_launch_pairs_probe(ctx, sk->d_hashes, sk->d_lens, sk->d_tables, sk->d_tags, n, s, sk->probe_cb, sk->d_row_start,
                                    (uint32_t)(sk->row_start.size() - 1), sk->n_work, d_cmin, drank, dworld, (uint32_t)row_lo, d_out, d_count, cap, sk->d_arranged);
            compared = ghip_probe_pairs_of_rank(n, sk->probe_cb, sk->row_start, drank, dworld);
        } else if (s <= 4096) {
            ghip_launch_pairs(ctx, sk->d_hashes, sk->d_lens, n, s, d_cmin, drank, dworld, (uint32_t)row_lo, d_out, d_count, cap, &compared);
        } else {
            ghip_launch_pairs_global(ctx, sk->d_hashes, sk->d_lens, n, s, d_cmin, drank, dworld, (uint32_t)row_lo, d_out, d_count, cap, &compared);
        }
        if (row_lo) compared = P - (uint64_t)row_lo * (row_lo - 1) / 2;   // the rectangle
        ctx->last_pairs = compared;
        unsigned long long cnt = 0;
        lap("kernels issued");
        if ((rc = d2h(ctx, &cnt, d_count, 1))) return rc;
        lap("count back (kernels done)");
        { hipError_t e = hipGetLastError(); if (e != hipSuccess) return ghip_set_error(ctx, GHIP_EHIP, std::string("pair_intersect_tile: ") + hipGetErrorString(e)); }
        if (cnt > cap) { cap = cnt; continue; }  // list overflowed: rerun with room for every hit
        host.resize(cnt);
        if ((rc = d2h(ctx, host.data(), d_out, cnt))) return rc;
        lap("candidates back");
        listed = true;
        big_pending = has_big;
        // the pairs of the empty sketches, which share no hash with anybody and pair with everybody (ANI 1.0 by the
        // reference's NaN arithmetic, common = total = 0): N - 1 each, in this rank's share of a sharded join
        if (!empties.empty()) {
            std::vector<uint8_t> is_empty(n, 0);
            for (uint32_t e : empties) is_empty[e] = 1;
            const uint32_t jrank = replicate_join ? 0 : rank, jworld = replicate_join ? 1 : world;
            for (uint32_t e : empties)
                for (size_t x = 0; x < n; x++) {
                    if (x == e || (is_empty[x] && x < e)) continue;   // two empty sketches: once
                    const uint32_t i = (uint32_t)std::min<size_t>(e, x), j = (uint32_t)std::max<size_t>(e, x);
                    if (j < row_lo) continue;
                    if (jworld > 1 && (i + j) % jworld != jrank) continue;
                    ghip_pair r; r.i = i; r.j = j; r.common = 0; r.total = 0; r.ani = 0.0f;
                    host.push_back(r);
                }
        }
    }
    if (!listed) return ghip_set_error(ctx, GHIP_EHIP, "precluster candidate list overflowed on every attempt");
    if (big_pending) {
        // The join left out the pairs of two genomes that both sit in an element bucket too large for it (a hash shared by
        // more than J_ELEM_CAP genomes: one very large family).  Those genomes' rows are gathered into a compact matrix, a
        // dense form runs over it, and the pairs come back under their own indices: the family costs |G|^2 / 2 probes, the
        // rest of the collection stays with the join (a dense pass over everything is 0.9 s at 50 000 genomes).
        std::vector<uint8_t> big(n);
        if ((rc = d2h(ctx, big.data(), d_big, n))) return rc;
        std::vector<uint32_t> G;
        for (size_t g = 0; g < n; g++) if (big[g]) G.push_back((uint32_t)g);
        if (G.size() >= 2) {
            ghip_sketches sub;
            sub.ctx = ctx; sub.n = G.size(); sub.s = s; sub.k = k; sub.owned = false;
            uint32_t *d_G = nullptr;
            if ((rc = dmalloc(ctx, &d_G, G.size())) || (tmp.add(d_G), false) || (rc = h2d(ctx, d_G, G.data(), G.size())) ||
                (rc = dmalloc(ctx, &sub.d_hashes, G.size() * (size_t)s)) || (tmp.add(sub.d_hashes), false) ||
                (rc = dmalloc(ctx, &sub.d_lens, G.size())) || (tmp.add(sub.d_lens), false)) return rc;
            ghip_launch_gather_rows(ctx, sk->d_hashes, sk->d_lens, d_G, G.size(), s, sub.d_hashes, sub.d_lens);
            bool sub_probe = s <= 1024 && force != GHIP_PAIR_MERGE;
            if (sub_probe) {
                if ((rc = prepare_probe(ctx, &sub))) return rc;
                tmp.add(sub.d_tables); tmp.add(sub.d_tags); tmp.add(sub.d_row_start); if (sub.d_arranged) tmp.add(sub.d_arranged);
                sub_probe = sub.probe_flags == 0;
            }
            const uint64_t Ps = (uint64_t)G.size() * (G.size() - 1) / 2;
            uint64_t scap = std::min<uint64_t>(Ps, std::max<uint64_t>(1u << 20, 64ull * G.size()));
            bool sub_listed = false;
            for (int attempt = 0; attempt < 4 && !sub_listed; attempt++) {
                ghip_pair *d_sub = nullptr;
                if ((rc = dmalloc(ctx, &d_sub, scap))) return rc;
                DeviceFree t3(ctx); t3.add(d_sub);
                GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
                uint64_t unused = 0;
                if (sub_probe)
                    ghip_launch_pairs_probe(ctx, sub.d_hashes, sub.d_lens, sub.d_tables, sub.d_tags, sub.n, s, sub.probe_cb, sub.d_row_start,
                                            (uint32_t)(sub.row_start.size() - 1), sub.n_work, d_cmin, 0, 1, 0, d_sub, d_count, scap, sub.d_arranged);
                else if (s <= 4096) ghip_launch_pairs(ctx, sub.d_hashes, sub.d_lens, sub.n, s, d_cmin, 0, 1, 0, d_sub, d_count, scap, &unused);
                else ghip_launch_pairs_global(ctx, sub.d_hashes, sub.d_lens, sub.n, s, d_cmin, 0, 1, 0, d_sub, d_count, scap, &unused);
                unsigned long long cnt = 0;
                if ((rc = d2h(ctx, &cnt, d_count, 1))) return rc;
                if (cnt > scap) { scap = cnt; continue; }
                std::vector<ghip_pair> part(cnt);
                if ((rc = d2h(ctx, part.data(), d_sub, cnt))) return rc;
                const uint32_t jrank = replicate_join ? 0 : rank, jworld = replicate_join ? 1 : world;
                for (ghip_pair &r : part) {
                    r.i = G[r.i]; r.j = G[r.j];   // (G ascends: i < j stays)
                    if (r.j < row_lo) continue;
                    if (jworld > 1 && (r.i + r.j) % jworld != jrank) continue;
                    host.push_back(r);
                }
                sub_listed = true;
            }
            if (!sub_listed) return ghip_set_error(ctx, GHIP_EHIP, "precluster candidate list overflowed on every attempt");
            lap("dense pass over the genomes of oversized buckets");
        }
    }
    rc = ghip_pairs_finalize(ctx, host, k, min_ani, n, filter_share, rank, world, out_pairs, out_n);
    lap("finalize (f64 recheck, (i, j) order)");
    return rc;
}

int ghip_precluster_dense_share(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, rank, world, false, 0, out_pairs, out_n, nullptr, true);
}

extern "C" int ghip_precluster_shard(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, uint32_t rank,
                                     uint32_t world, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, rank, world, false, 0, out_pairs, out_n, nullptr);
}

extern "C" int ghip_precluster_ranks(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, uint32_t rank,
                                     uint32_t world, ghip_pair **out_pairs, size_t *out_n, int *out_replicated) {
    if (!out_replicated) return GHIP_EINVAL;
    // Default: the pair work is SHARDED -- dense forms by tile, the join form by (i + j) mod world at record emission
    // (its element stage, one pass over all N*s hashes, runs on every rank).  GHIP_JOIN_RANKS=replicate makes every
    // rank run the whole join and keep the whole list instead (no candidate exchange; DESIGN.md section 6 has both timings).
    const bool replicate = world > 1 && ctx->opt.join_ranks == GHIP_JOIN_REPLICATE;
    return precluster_impl(ctx, sk, min_ani, rank, world, replicate, 0, out_pairs, out_n, out_replicated);
}

extern "C" int ghip_precluster(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, 0, 1, false, 0, out_pairs, out_n, nullptr);
}

extern "C" int ghip_precluster_from(ghip_ctx *ctx, const ghip_sketches *sk, size_t row_lo, float min_ani, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, 0, 1, false, row_lo, out_pairs, out_n, nullptr);
}

// ------------------------------------------------------------------------------------ ANI
int ghip_index_wait(ghip_ctx *ctx, const ghip_ani_index *idx) {
    if (!idx->bin_done) return GHIP_OK;
    const hipError_t e = hipEventSynchronize(idx->bin_done);
    hipEventDestroy(idx->bin_done);
    idx->bin_done = nullptr;
    for (void *p : idx->bin_scratch) ghip_pool_free(ctx, p);   // the unordered lists, the segment counts
    idx->bin_scratch.clear();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return ghip_set_error(ctx, GHIP_EHIP, "ANI index kernels failed");
    return GHIP_OK;
}

static void free_index_arrays_locked(ghip_ani_index *idx) {
    ghip_ctx *ctx = idx->ctx;
    (void)ghip_index_wait(ctx, idx);   // the side stream may still be writing them
    if (idx->owned) {
        ghip_pool_free(ctx, idx->d_seed_code); ghip_pool_free(ctx, idx->d_seed_loc);
        ghip_pool_free(ctx, idx->d_bin_start); ghip_pool_free(ctx, idx->d_chunk_total);
    }
    ghip_pool_free(ctx, idx->d_seed_start); ghip_pool_free(ctx, idx->d_seed_count); ghip_pool_free(ctx, idx->d_seg_count);
    ghip_pool_free(ctx, idx->d_chunk_start); ghip_pool_free(ctx, idx->d_glen); ghip_pool_free(ctx, idx->d_seed_thr);
    idx->d_seed_thr = nullptr;
    idx->d_seed_code = nullptr; idx->d_seed_loc = nullptr; idx->d_bin_start = nullptr; idx->d_chunk_total = nullptr;
    idx->d_seed_start = nullptr; idx->d_seed_count = nullptr; idx->d_seg_count = nullptr; idx->d_chunk_start = nullptr;
    idx->d_glen = nullptr;
}

static void free_index_locked(ghip_ani_index *idx) {  // ctx->mu held
    ghip_ctx *ctx = idx->ctx;
    free_index_arrays_locked(idx);
    ctx->live_handles--;
    delete idx;
}

extern "C" void ghip_ani_index_free(ghip_ani_index *idx) {
    if (!idx) return;
    ghip_ctx *ctx = idx->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStreamSynchronize(ctx->stream);
        free_index_locked(idx);
    }
    ctx_release(ctx);
}

// ---- ANI index construction, in steps so that the seeding pass can be the standalone ani_seeds
// kernel or ride along with the MinHash pass (ghip_sketch_and_index)
static int index_new(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t c, uint32_t chunk, ghip_ani_index **out,
                     std::vector<uint64_t> &cap) {
    if (k < 1 || k > 16 || c < 1 || chunk < 1 || chunk > GHIP_ANI_MAX_CHUNK_LEN)
        return ghip_set_error(ctx, GHIP_EINVAL, "bad ANI sketch parameters (k must be 1..=16, chunk 1..=32768)");
    const size_t n = g->n;
    ghip_ani_index *idx = new ghip_ani_index();
    ctx->live_handles++;
    idx->ctx = ctx; idx->n = n; idx->k = k; idx->c = c; idx->chunk = chunk;
    idx->glen = g->lens;
    idx->chunk_start.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
        uint64_t nch = (g->lens[i] + chunk - 1) / chunk;
        idx->max_chunks = (uint32_t)std::max<uint64_t>(idx->max_chunks, nch);
        idx->chunk_start[i + 1] = idx->chunk_start[i] + nch;
    }
    if (idx->max_chunks > GHIP_ANI_MAX_CHUNKS) {
        free_index_locked(idx);
        return ghip_set_error(ctx, GHIP_EINVAL, "genome too long for the ANI index (a seed's chunk is a 16-bit field: at most 65535 chunks per genome, 1.3 Gb at the default 20 kb chunk)");
    }
    cap.resize(n);
    idx->seed_thr.resize(n);
    for (size_t i = 0; i < n; i++) {
        const uint32_t cg = ghip_ani_density(g->lens[i], c);
        idx->seed_thr[i] = ~0u / cg;
        cap[i] = ghip_ani_seed_capacity(g->lens[i], cg);
    }
    *out = idx;
    return GHIP_OK;
}

// (re)allocates the seed arrays for the given capacities and zeroes the counters
static int index_alloc_seeds(ghip_ctx *ctx, ghip_ani_index *idx, const std::vector<uint64_t> &cap) {
    const size_t n = idx->n;
    int rc;
    idx->seed_start.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) idx->seed_start[i + 1] = idx->seed_start[i] + cap[i];
    hipStreamSynchronize(ctx->stream);
    free_index_arrays_locked(idx);
    if ((rc = dmalloc(ctx, &idx->d_seed_code, idx->seed_start[n]))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_seed_loc, idx->seed_start[n]))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_seed_start, n + 1))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_seed_count, n))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_seg_count, n * GHIP_ANI_SEGMENTS))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_chunk_total, idx->chunk_start[n]))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_chunk_start, n + 1))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_glen, n))) return rc;
    if ((rc = dmalloc(ctx, &idx->d_seed_thr, n))) return rc;
    if ((rc = h2d_nosync(ctx, idx->d_seed_thr, idx->seed_thr.data(), n))) return rc;
    if ((rc = h2d_nosync(ctx, idx->d_seed_start, idx->seed_start.data(), n + 1))) return rc;
    if ((rc = h2d_nosync(ctx, idx->d_chunk_start, idx->chunk_start.data(), n + 1))) return rc;
    if ((rc = h2d_nosync(ctx, idx->d_glen, idx->glen.data(), n))) return rc;
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(idx->d_seg_count, 0, std::max<size_t>(n, 1) * GHIP_ANI_SEGMENTS * sizeof(uint32_t), ctx->stream));
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(idx->d_chunk_total, 0, std::max<uint64_t>(idx->chunk_start[n], 1) * sizeof(uint32_t), ctx->stream));
    return GHIP_OK;  // (the host vectors just queued for upload are only rewritten by a retry, i.e. after index_check_seeds has synchronised)
}

static ghip_seed_args index_seed_args(const ghip_ani_index *idx) {
    return ghip_seed_args{idx->k, idx->chunk, idx->d_seed_thr, idx->d_seed_code, idx->d_seed_loc, idx->d_seed_start,
                          idx->d_seg_count, idx->d_chunk_total, idx->d_chunk_start};
}

// reads the per-segment seed counts; *overflow = some segment was too small (the genome's capacity is raised to
// GHIP_ANI_SEGMENTS x its fullest segment's exact count)
static int index_check_seeds(ghip_ctx *ctx, ghip_ani_index *idx, std::vector<uint64_t> &cap, bool *overflow) {
    const size_t n = idx->n;
    std::vector<uint32_t> seg(n * GHIP_ANI_SEGMENTS);
    idx->seed_count.assign(n, 0);
    int rc = d2h(ctx, seg.data(), idx->d_seg_count, n * GHIP_ANI_SEGMENTS);
    if (rc) return rc;
    *overflow = false;
    for (size_t i = 0; i < n; i++) {
        uint64_t tot = 0, mx = 0;
        for (size_t s = 0; s < GHIP_ANI_SEGMENTS; s++) { tot += seg[i * GHIP_ANI_SEGMENTS + s]; mx = std::max<uint64_t>(mx, seg[i * GHIP_ANI_SEGMENTS + s]); }
        idx->seed_count[i] = (uint32_t)tot;
        if (mx * GHIP_ANI_SEGMENTS > cap[i]) { *overflow = true; cap[i] = mx * GHIP_ANI_SEGMENTS; }
    }
    if (!*overflow) rc = h2d_nosync(ctx, idx->d_seed_count, idx->seed_count.data(), n);   // (the host vector lives as long as the index)
    return rc;
}

// standalone seeding with retry on overflow
static int index_seed_standalone(ghip_ctx *ctx, const ghip_genomes *g, ghip_ani_index *idx, std::vector<uint64_t> &cap) {
    for (int attempt = 0; attempt < 2; attempt++) {
        int rc = index_alloc_seeds(ctx, idx, cap);
        if (rc) return rc;
        ghip_launch_ani_seeds(ctx, g, idx->k, idx->d_seed_thr, idx->chunk, idx->d_seed_code, idx->d_seed_loc, idx->d_seed_start,
                              idx->d_seg_count, idx->d_chunk_total, idx->d_chunk_start, g->d_work, g->n_work);
        bool overflow = false;
        if ((rc = index_check_seeds(ctx, idx, cap, &overflow))) return rc;
        if (!overflow) return GHIP_OK;
    }
    return ghip_set_error(ctx, GHIP_EHIP, "ANI seed list overflowed twice");
}

// reorder every genome's seed list by hash bin and record the bin offsets (the join index)
static int index_finish(ghip_ctx *ctx, ghip_ani_index *idx, bool defer = false) {
    const size_t n = idx->n;
    int rc;
    uint32_t *d_code2 = nullptr, *d_pos = nullptr, *d_chunk2 = nullptr;
    if (!(rc = dmalloc(ctx, &idx->d_bin_start, n * (size_t)(GHIP_ANI_BIN_COUNT + 1))) &&
        !(rc = dmalloc(ctx, &d_code2, idx->seed_start[n])) && !(rc = dmalloc(ctx, &d_chunk2, idx->seed_start[n])) &&
        !(rc = dmalloc(ctx, &d_pos, idx->seed_start[n]))) {
        hipEvent_t seeded = nullptr;
        if (defer && !ctx->side_stream && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) { ctx->side_stream = nullptr; defer = false; (void)hipGetLastError(); }
        if (defer && (hipEventCreateWithFlags(&seeded, hipEventDisableTiming) != hipSuccess || hipEventRecord(seeded, ctx->stream) != hipSuccess ||
                      hipStreamWaitEvent(ctx->side_stream, seeded, 0) != hipSuccess ||
                      hipEventCreateWithFlags(&idx->bin_done, hipEventDisableTiming) != hipSuccess)) {
            if (idx->bin_done) { hipEventDestroy(idx->bin_done); idx->bin_done = nullptr; }
            defer = false; (void)hipGetLastError();
        }
        if (seeded) hipEventDestroy(seeded);   // (the wait it feeds is already enqueued)
        // deferred: the binning goes to the side stream, behind the seeding kernels of the main one, and this call returns
        // without waiting for it -- the main stream's next stage (the pair stage: sketches only) runs next to it
        hipStream_t main_stream = ctx->stream;
        if (defer) ctx->stream = ctx->side_stream;   // (ctx->mu held: the launcher and its profiling events follow ctx->stream)
        ghip_launch_ani_bin(ctx, n, idx->d_seed_code, idx->d_seed_loc, d_code2, d_chunk2, idx->d_seed_start,
                            idx->d_seg_count, idx->d_bin_start, d_pos);
        ctx->stream = main_stream;
        std::swap(idx->d_seed_code, d_code2);
        std::swap(idx->d_seed_loc, d_chunk2);
        if (defer) {
            if (hipEventRecord(idx->bin_done, ctx->side_stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "ANI index kernels failed");
            idx->bin_scratch = {d_code2, d_chunk2, d_pos, idx->d_seg_count};
            idx->d_seg_count = nullptr;
            if (rc) (void)ghip_index_wait(ctx, idx);
            return rc;
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
            rc = ghip_set_error(ctx, GHIP_EHIP, "ANI index kernels failed");
    }
    ghip_pool_free(ctx, d_code2); ghip_pool_free(ctx, d_chunk2); ghip_pool_free(ctx, d_pos);  // the unordered lists
    ghip_pool_free(ctx, idx->d_seg_count);
    idx->d_seg_count = nullptr;
    return rc;
}

extern "C" int ghip_ani_index_build(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t c, uint32_t chunk,
                                    ghip_ani_index **out) {
    if (!ctx || !g || !out) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ghip_ani_index *idx = nullptr;
    std::vector<uint64_t> cap;
    int rc = index_new(ctx, g, k, c, chunk, &idx, cap);
    if (rc) return rc;
    if (!(rc = index_seed_standalone(ctx, g, idx, cap))) rc = index_finish(ctx, idx);
    if (rc) { free_index_locked(idx); return rc; }
    *out = idx;
    return GHIP_OK;
}

// One pass over the bases for both sketches: the MinHash k-mer pass also emits the ANI seeds
// (sketch.hip: sketch_kmers<21, true>).  Same results as ghip_sketch_genomes + ghip_ani_index_build.
extern "C" int ghip_sketch_and_index(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s, uint64_t seed,
                                     uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, ghip_sketches **out_sk,
                                     ghip_ani_index **out_idx) {
    if (!ctx || !g || !out_sk || !out_idx) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ghip_ani_index *idx = nullptr;
    ghip_sketches *sk = nullptr;
    std::vector<uint64_t> cap;
    int rc = index_new(ctx, g, ani_k, ani_c, ani_chunk, &idx, cap);
    if (rc) return rc;
    const bool fuse = (k == 21 && ani_k <= k);  // the fused kernel exists for finch's k = 21
    if (fuse) {
        if (!(rc = index_alloc_seeds(ctx, idx, cap))) {
            const ghip_seed_args sa = index_seed_args(idx);
            rc = sketch_genomes_locked(ctx, g, k, s, seed, &sa, &sk);
        }
        bool overflow = false;
        if (!rc) rc = index_check_seeds(ctx, idx, cap, &overflow);
        if (!rc && overflow) rc = index_seed_standalone(ctx, g, idx, cap);  // exact counts now known
    } else {
        rc = sketch_genomes_locked(ctx, g, k, s, seed, nullptr, &sk);
        if (!rc) rc = index_seed_standalone(ctx, g, idx, cap);
    }
    if (!rc) rc = index_finish(ctx, idx, ctx->opt.overlap_binning != 0);   // the binning overlaps the caller's pair stage
    if (rc) { if (sk) free_sketches_locked(sk); free_index_locked(idx); return rc; }
    *out_sk = sk;
    *out_idx = idx;
    return GHIP_OK;
}

// Files in -> MinHash sketches (+ ANI index, + assembly statistics), with at most `batch_bytes` of bases resident in
// HBM at a time: the files are ingested, sketched and seeded batch by batch and the per-batch results -- packed
// sketch rows and the flat, genome-relative index arrays -- are concatenated on the device.  One batch (the common
// case) returns its handles as they are.
extern "C" int ghip_sketch_and_index_files(ghip_ctx *ctx, const char *const *paths, size_t n, uint32_t k, uint32_t s, uint64_t seed,
                                           uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, int io_threads, uint64_t batch_bytes,
                                           ghip_sketches **out_sk, ghip_ani_index **out_idx, uint64_t *out_stats) {
    if (!ctx || !out_sk || (n && !paths)) return GHIP_EINVAL;
    if (batch_bytes == 0) batch_bytes = 96ull << 30;  // a third of the 288 GB of an MI355X
    *out_sk = nullptr;
    if (out_idx) *out_idx = nullptr;
    std::vector<ghip_sketches *> sks;
    std::vector<ghip_ani_index *> idxs;
    auto drop = [&]() {
        for (auto *x : sks) ghip_sketches_free(x);
        for (auto *x : idxs) ghip_ani_index_free(x);
    };
    // ---- plan the batches.  Two reasons to split: the HBM bound (batch_bytes), and OVERLAP -- while the kernels of batch b
    // run (sketch + seed pass, ~2.2 ms per GB), the ingest threads already fill batch b + 1 over PCIe (~19 ms per GB):
    // inputs above 1 GiB go in 4 pieces (at least 512 MiB each; every piece costs ~2 ms of set-up, synchronisations and
    // concatenation, so 8 pieces gave back what they hid), ingested by a producer thread one piece ahead.
    std::vector<uint64_t> caps(n), hints(n);
    parallel_ranges(n, 64, (size_t)std::min(std::max(1, io_threads), 8), [&](size_t b0, size_t e0) {
        for (size_t i = b0; i < e0; i++) { hints[i] = ghip_stream_capacity_hint(paths[i]); caps[i] = hints[i] + GHIP_TAIL_PAD + GHIP_BASE_ALIGN; }   // base positions
    });
    uint64_t total_bytes = 0;
    for (uint64_t c : caps) total_bytes += c;
    const uint64_t batch_bases = batch_bytes / 3 * 8;   // the resident form takes 3 bits per base (2-bit code + validity bit)
    uint64_t piece = batch_bases;
    {
        // (gzip input is inflate-bound on the host: cutting it into pieces only adds eight load-imbalanced tails)
        size_t n_gz = 0;
        for (size_t i = 0; i < n; i++) { const size_t l = strlen(paths[i]); n_gz += (l > 3 && !strcmp(paths[i] + l - 3, ".gz")) ? 1 : 0; }
        if (ctx->opt.pipeline_pieces && n_gz == 0 && total_bytes > (1ull << 30))
            piece = std::min<uint64_t>(batch_bases, std::max<uint64_t>(total_bytes / 4 + 1, 512ull << 20));
    }
    std::vector<std::pair<size_t, size_t>> ranges;
    for (size_t first = 0; first < n || ranges.empty();) {  // at least one (possibly empty) batch, so that n == 0 yields empty handles
        size_t last = first;
        uint64_t bytes = 0;
        while (last < n) {
            if (last > first && bytes + caps[last] > piece) break;
            bytes += caps[last];
            last++;
        }
        ranges.push_back({first, last});
        first = last;
        if (n == 0) break;
    }
    // producer: ingests the batches in order, at most two ahead of the consumer
    struct Ingested { ghip_genomes *g; int rc; };
    std::vector<Ingested> ready(ranges.size(), Ingested{nullptr, GHIP_OK});
    std::mutex qmu;
    std::condition_variable qcv;
    size_t produced = 0, consumed = 0;
    bool abort_producer = false;
    auto ingest = [&](size_t b) {
        ghip_genomes *g = nullptr;
        const int rc = genomes_from_files_impl(ctx, paths + ranges[b].first, ranges[b].second - ranges[b].first, io_threads, hints.data() + ranges[b].first, &g);
        std::lock_guard<std::mutex> l(qmu);
        ready[b] = Ingested{g, rc};
        produced = b + 1;
        qcv.notify_all();
    };
    std::thread producer;
    if (ranges.size() > 1)
        producer = std::thread([&] {
            for (size_t b = 0; b < ranges.size(); b++) {
                {
                    std::unique_lock<std::mutex> l(qmu);
                    qcv.wait(l, [&] { return abort_producer || b < consumed + 2; });
                    if (abort_producer) return;
                }
                ingest(b);
                if (ready[b].rc != GHIP_OK) return;
            }
        });
    auto stop_producer = [&]() {
        if (!producer.joinable()) return;
        { std::lock_guard<std::mutex> l(qmu); abort_producer = true; qcv.notify_all(); }
        producer.join();
        for (auto &r : ready) if (r.g) { ghip_genomes_free(r.g); r.g = nullptr; }
    };
    for (size_t b = 0; b < ranges.size(); b++) {
        if (ranges.size() == 1) ingest(0);
        else {
            std::unique_lock<std::mutex> l(qmu);
            qcv.wait(l, [&] { return produced > b; });
        }
        ghip_genomes *g = ready[b].g;
        ready[b].g = nullptr;
        int rc = ready[b].rc;
        const size_t first = ranges[b].first, last = ranges[b].second;
        ghip_sketches *sk = nullptr;
        ghip_ani_index *idx = nullptr;
        if (!rc) rc = out_idx ? ghip_sketch_and_index(ctx, g, k, s, seed, ani_k, ani_c, ani_chunk, &sk, &idx)
                              : ghip_sketch_genomes(ctx, g, k, s, seed, &sk);
        if (!rc && out_stats)
            for (size_t i = first; i < last; i++) {
                const ghip_genome_stats &st = g->stats[i - first];
                out_stats[3 * i] = st.num_contigs; out_stats[3 * i + 1] = st.num_ambiguous_bases; out_stats[3 * i + 2] = st.n50;
            }
        if (g) ghip_genomes_free(g);
        { std::lock_guard<std::mutex> l(qmu); consumed = b + 1; qcv.notify_all(); }
        if (rc) {
            const std::string msg = ghip_last_error(ctx);   // the producer's later calls must not overwrite the cause
            stop_producer();
            if (sk) ghip_sketches_free(sk);
            if (idx) ghip_ani_index_free(idx);
            drop();
            return ghip_set_error(ctx, rc, msg);
        }
        sks.push_back(sk);
        if (out_idx) idxs.push_back(idx);
    }
    if (producer.joinable()) producer.join();
    if (sks.size() == 1) {
        *out_sk = sks[0];
        if (out_idx) *out_idx = idxs[0];
        return GHIP_OK;
    }
    // ---- concatenate the batches on the device
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = GHIP_OK;
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = n; sk->s = s; sk->k = k;
    if (!(rc = dmalloc(ctx, &sk->d_hashes, n * (size_t)s)) && !(rc = dmalloc(ctx, &sk->d_lens, n))) {
        size_t at = 0;
        for (auto *b : sks) {
            if (b->n && (hipMemcpyAsync(sk->d_hashes + at * s, b->d_hashes, b->n * (size_t)s * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                         hipMemcpyAsync(sk->d_lens + at, b->d_lens, b->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess))
                rc = ghip_set_error(ctx, GHIP_EHIP, "sketch concatenation failed");
            at += b->n;
        }
    }
    ghip_ani_index *idx = nullptr;
    if (!rc && out_idx) {
        idx = new ghip_ani_index();
        ctx->live_handles++;
        idx->ctx = ctx; idx->n = n; idx->k = ani_k; idx->c = ani_c; idx->chunk = ani_chunk;
        idx->seed_start.assign(1, 0); idx->chunk_start.assign(1, 0);
        for (auto *b : idxs) {
            idx->max_chunks = std::max(idx->max_chunks, b->max_chunks);
            idx->glen.insert(idx->glen.end(), b->glen.begin(), b->glen.end());
            idx->seed_count.insert(idx->seed_count.end(), b->seed_count.begin(), b->seed_count.end());
            idx->seed_thr.insert(idx->seed_thr.end(), b->seed_thr.begin(), b->seed_thr.end());
            for (size_t i = 0; i < b->n; i++) {
                idx->seed_start.push_back(idx->seed_start.back() + (b->seed_start[i + 1] - b->seed_start[i]));
                idx->chunk_start.push_back(idx->chunk_start.back() + (b->chunk_start[i + 1] - b->chunk_start[i]));
            }
        }
        const uint64_t n_seed = idx->seed_start[n], n_chunk = idx->chunk_start[n];
        if (!(rc = dmalloc(ctx, &idx->d_seed_code, n_seed)) && !(rc = dmalloc(ctx, &idx->d_seed_loc, n_seed)) &&
            !(rc = dmalloc(ctx, &idx->d_bin_start, n * (size_t)(GHIP_ANI_BIN_COUNT + 1))) && !(rc = dmalloc(ctx, &idx->d_chunk_total, n_chunk)) &&
            !(rc = dmalloc(ctx, &idx->d_seed_start, n + 1)) && !(rc = dmalloc(ctx, &idx->d_seed_count, n)) &&
            !(rc = dmalloc(ctx, &idx->d_chunk_start, n + 1)) && !(rc = dmalloc(ctx, &idx->d_glen, n)) &&
            !(rc = dmalloc(ctx, &idx->d_seed_thr, n)) && !(rc = h2d(ctx, idx->d_seed_thr, idx->seed_thr.data(), n)) &&
            !(rc = h2d(ctx, idx->d_seed_start, idx->seed_start.data(), n + 1)) && !(rc = h2d(ctx, idx->d_seed_count, idx->seed_count.data(), n)) &&
            !(rc = h2d(ctx, idx->d_chunk_start, idx->chunk_start.data(), n + 1)) && !(rc = h2d(ctx, idx->d_glen, idx->glen.data(), n))) {
            uint64_t at_seed = 0, at_chunk = 0;
            size_t at_g = 0;
            for (auto *b : idxs) {
                const uint64_t ns = b->seed_start[b->n], nc = b->chunk_start[b->n];
                hipError_t e = hipSuccess;
                if (ns) e = hipMemcpyAsync(idx->d_seed_code + at_seed, b->d_seed_code, ns * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
                if (ns && e == hipSuccess) e = hipMemcpyAsync(idx->d_seed_loc + at_seed, b->d_seed_loc, ns * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
                if (b->n && e == hipSuccess) e = hipMemcpyAsync(idx->d_bin_start + at_g * (GHIP_ANI_BIN_COUNT + 1), b->d_bin_start, b->n * (size_t)(GHIP_ANI_BIN_COUNT + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
                if (nc && e == hipSuccess) e = hipMemcpyAsync(idx->d_chunk_total + at_chunk, b->d_chunk_total, nc * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
                if (e != hipSuccess) { rc = ghip_set_error(ctx, GHIP_EHIP, "ANI index concatenation failed"); break; }
                at_seed += ns; at_chunk += nc; at_g += b->n;
            }
        }
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "concatenation failed");
    for (auto *b : sks) free_sketches_locked(b);
    for (auto *b : idxs) free_index_locked(b);
    if (rc) { free_sketches_locked(sk); if (idx) free_index_locked(idx); return rc; }
    *out_sk = sk;
    if (out_idx) *out_idx = idx;
    return GHIP_OK;
}

// device part of ghip_ani_pairs: res[6 p ..] = M and T of the median-containment chunk, aligned bases of q, aligned chunks,
// (unused), aligned bases of r
static int ani_pairs_device(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *pairs, size_t n, uint64_t *res) {
    // GHIP_ANI_DEBUG=1: where the wall time of one call goes (stderr)
    const bool dbg = ghip_dbg(ctx->opt, GHIP_DEBUG_ANI);
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ani_pairs %zu] %-22s %8.3f ms\n", n, what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // the context is held for the device part only: the host finish runs next to other callers' launches
    // (calculate_ani arrives from many rayon workers at once, src/clusterer.rs:267-296)
    std::lock_guard<std::mutex> lk(ctx->mu);
    lap("context lock");
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    { const int rcw = ghip_index_wait(ctx, idx); if (rcw) return rcw; }
    lap("index wait (binning)");
    DeviceFree tmp(ctx);
    uint32_t *d_pairs = nullptr;
    uint64_t *d_out = nullptr;
    int rc;
    if ((rc = dmalloc(ctx, &d_pairs, 2 * n))) return rc;
    tmp.add(d_pairs);
    if ((rc = dmalloc(ctx, &d_out, 6 * n))) return rc;
    tmp.add(d_out);
    lap("device buffers");
    if ((rc = h2d(ctx, d_pairs, pairs, 2 * n))) return rc;
    lap("pairs to device");
    const size_t batch = (size_t)1 << 22;  // 512 threads per pair: keep each dispatch below 2^32 work-items
    for (size_t off = 0; off < n; off += batch)
        if ((rc = ghip_launch_ani_pairs(ctx, idx, pairs + 2 * off, d_pairs + 2 * off, std::min(batch, n - off), d_out + 6 * off))) return rc;
    lap("launch");
    if ((rc = d2h(ctx, res, d_out, 6 * n))) return rc;
    lap("kernel + results to host");
    { hipError_t e = hipGetLastError(); if (e != hipSuccess) return ghip_set_error(ctx, GHIP_EHIP, std::string("ani_pairs: ") + hipGetErrorString(e)); }
    return GHIP_OK;
}

extern "C" int ghip_ani_pairs_detail(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *pairs, size_t n, uint64_t *out) {
    if (!ctx || !idx || !out || (n && !pairs)) return GHIP_EINVAL;
    for (size_t i = 0; i < 2 * n; i++) if (pairs[i] >= idx->n) return ghip_set_error(ctx, GHIP_EINVAL, "genome index out of range");
    if (n == 0) return GHIP_OK;
    std::vector<uint64_t> res(6 * n);
    int rc = ani_pairs_device(ctx, idx, pairs, n, res.data());
    if (rc) return rc;
    for (size_t p = 0; p < n; p++) {
        const uint32_t tq = idx->seed_thr[pairs[2 * p]], tr = idx->seed_thr[pairs[2 * p + 1]];
        out[6 * p] = res[6 * p]; out[6 * p + 1] = res[6 * p + 1]; out[6 * p + 2] = res[6 * p + 3];
        out[6 * p + 3] = res[6 * p + 2]; out[6 * p + 4] = res[6 * p + 5];
        out[6 * p + 5] = ~0u / std::min(tq, tr);   // thr = (2^32 - 1) / c exactly inverts for c <= 65535
    }
    return GHIP_OK;
}

extern "C" int ghip_ani_pairs(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *pairs, size_t n,
                              float min_af, float *out_ani, float *out_af) {
    if (!ctx || !idx || !out_ani || (n && !pairs)) return GHIP_EINVAL;
    for (size_t i = 0; i < 2 * n; i++) if (pairs[i] >= idx->n) return ghip_set_error(ctx, GHIP_EINVAL, "genome index out of range");
    if (n == 0) return GHIP_OK;
    const bool dbg = ghip_dbg(ctx->opt, GHIP_DEBUG_ANI);
    const auto t_in = std::chrono::steady_clock::now();
    std::vector<uint64_t> res(6 * n);
    { const int rc = ani_pairs_device(ctx, idx, pairs, n, res.data()); if (rc) return rc; }
    const auto t_dev = std::chrono::steady_clock::now();
    // the f64 pow and the two-decimal rounding stay on the host (glibc's pow is what the oracle's parity is defined
    // by): ~70 ns per pair, spread over threads from 20 000 pairs on (below that spawning costs more than it saves)
    auto finish_range = [&](size_t p0, size_t p1) {
    for (size_t p = p0; p < p1; p++) {
        // res: [0] M and [1] T of the median-containment chunk, [2] aligned bases of q, [3] #aligned chunks, [5] bases of r
        const uint64_t M = res[6 * p], T = res[6 * p + 1], n_aligned = res[6 * p + 3];
        const uint32_t q = pairs[2 * p], r = pairs[2 * p + 1];
        const double afq = idx->glen[q] ? (double)res[6 * p + 2] / (double)idx->glen[q] : 0.0;
        const double afr = idx->glen[r] ? (double)res[6 * p + 5] / (double)idx->glen[r] : 0.0;
        if (out_af) { out_af[2 * p] = (float)afq; out_af[2 * p + 1] = (float)afr; }
        float v = 0.0f;
        if (n_aligned != 0 && T != 0 && !(afq < (double)min_af && afr < (double)min_af)) {
            // skani prints ANI with two decimals and galah parses that text as f32 (src/skani.rs:770)
            const double c = (double)M / (double)T;   // colinear seed matches only (ani.hip): no chance-match term
            double ani = 100.0 * std::pow(c, 1.0 / (double)idx->k);
            v = two_decimals_as_f32(ani);
        }
        out_ani[p] = v;
    }
    };
    // ~70 ns per pair (pow + the two-decimal rounding): 0.3 ms of a 14 ms step at 4 500 pairs.  Fresh threads cost more
    // than they save below ~25 000 pairs (30 us each to spawn), the context's persistent I/O workers do not (a wake-up
    // is ~10 us) -- used when no ingest holds them; otherwise, and for short lists, the calling thread does it all.
    const size_t workers = std::min<size_t>(16, n / 1000);
    if (workers >= 2 && ctx->ingest_mu.try_lock()) {
        const size_t per = (n + workers - 1) / workers;
        ctx->io.run((int)workers, [&](int w) { finish_range(std::min(n, (size_t)w * per), std::min(n, ((size_t)w + 1) * per)); });
        ctx->ingest_mu.unlock();
    } else parallel_ranges(n, 12500, 16, finish_range);
    if (dbg) fprintf(stderr, "[ani_pairs %zu] device part %.3f ms, host finish %.3f ms\n", n, std::chrono::duration<double, std::milli>(t_dev - t_in).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dev).count());
    return GHIP_OK;
}

extern "C" int ghip_ani_index_layout(const ghip_ani_index *idx, ghip_ani_layout *out) {
    if (!idx || !out) return GHIP_EINVAL;
    { std::lock_guard<std::mutex> lk(idx->ctx->mu); const int rcw = ghip_index_wait(idx->ctx, idx); if (rcw) return rcw; }   // the arrays it names must be final
    out->n = idx->n;
    out->n_seed_slots = idx->seed_start[idx->n];
    out->n_bin_slots = (uint64_t)idx->n * (GHIP_ANI_BIN_COUNT + 1);
    out->n_chunk_slots = idx->chunk_start[idx->n];
    out->d_seed_code = idx->d_seed_code; out->d_seed_loc = idx->d_seed_loc;
    out->d_bin_start = idx->d_bin_start; out->d_chunk_total = idx->d_chunk_total;
    return GHIP_OK;
}

extern "C" int ghip_ani_index_meta(const ghip_ani_index *idx, uint64_t *genome_len, uint64_t *seed_cap, uint32_t *seed_count) {
    if (!idx) return GHIP_EINVAL;
    for (size_t i = 0; i < idx->n; i++) {
        if (genome_len) genome_len[i] = idx->glen[i];
        if (seed_cap) seed_cap[i] = idx->seed_start[i + 1] - idx->seed_start[i];
        if (seed_count) seed_count[i] = idx->seed_count[i];
    }
    return GHIP_OK;
}

extern "C" int ghip_ani_index_wrap_device(ghip_ctx *ctx, size_t n, uint32_t k, uint32_t c, uint32_t chunk,
                                          const uint64_t *genome_len, const uint64_t *seed_cap,
                                          const uint32_t *seed_count, void *d_seed_code, void *d_seed_loc,
                                          void *d_bin_start, void *d_chunk_total, ghip_ani_index **out) {
    if (!ctx || !out || chunk == 0 || chunk > GHIP_ANI_MAX_CHUNK_LEN || (n && (!genome_len || !seed_cap || !seed_count))) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ghip_ani_index *idx = new ghip_ani_index();
    ctx->live_handles++;
    idx->ctx = ctx; idx->n = n; idx->k = k; idx->c = c; idx->chunk = chunk; idx->owned = false;
    idx->glen.assign(genome_len, genome_len + n);
    idx->seed_count.assign(seed_count, seed_count + n);
    idx->seed_start.assign(n + 1, 0); idx->chunk_start.assign(n + 1, 0);
    idx->seed_thr.resize(n);
    for (size_t i = 0; i < n; i++) idx->seed_thr[i] = ~0u / ghip_ani_density(genome_len[i], c);   // the density follows from the length
    for (size_t i = 0; i < n; i++) {
        if (seed_count[i] > seed_cap[i]) { free_index_locked(idx); return ghip_set_error(ctx, GHIP_EINVAL, "inconsistent ANI index metadata"); }
        uint64_t nch = (genome_len[i] + chunk - 1) / chunk;
        idx->max_chunks = (uint32_t)std::max<uint64_t>(idx->max_chunks, nch);
        idx->seed_start[i + 1] = idx->seed_start[i] + seed_cap[i];
        idx->chunk_start[i + 1] = idx->chunk_start[i] + nch;
    }
    idx->d_seed_code = (uint32_t *)d_seed_code; idx->d_seed_loc = (uint32_t *)d_seed_loc;
    idx->d_bin_start = (uint32_t *)d_bin_start; idx->d_chunk_total = (uint32_t *)d_chunk_total;
    int rc = GHIP_OK;
    if (idx->max_chunks > GHIP_ANI_MAX_CHUNKS) rc = ghip_set_error(ctx, GHIP_EINVAL, "genome too long for the ANI index (at most 65535 chunks per genome)");
    if (!rc) rc = dmalloc(ctx, &idx->d_seed_thr, n);
    if (!rc) rc = h2d(ctx, idx->d_seed_thr, idx->seed_thr.data(), n);
    if (!rc) rc = dmalloc(ctx, &idx->d_seed_start, n + 1);
    if (!rc) rc = dmalloc(ctx, &idx->d_seed_count, n);
    if (!rc) rc = dmalloc(ctx, &idx->d_chunk_start, n + 1);
    if (!rc) rc = dmalloc(ctx, &idx->d_glen, n);
    if (!rc) rc = h2d(ctx, idx->d_seed_start, idx->seed_start.data(), n + 1);
    if (!rc) rc = h2d(ctx, idx->d_seed_count, idx->seed_count.data(), n);
    if (!rc) rc = h2d(ctx, idx->d_chunk_start, idx->chunk_start.data(), n + 1);
    if (!rc) rc = h2d(ctx, idx->d_glen, idx->glen.data(), n);
    if (rc) { free_index_locked(idx); return rc; }
    *out = idx;
    return GHIP_OK;
}
