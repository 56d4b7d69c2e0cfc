// Shared by the api_*.cpp translation units of libgalah_hip.so (round 4: api.cpp, 2 400 lines, split by stage):
// device-memory and copy helpers, the genome layout helpers, the exact reference arithmetic of the pair stage, and the
// few functions one stage needs of another.  Host orchestration only.
#pragma once
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
#include <functional>


int ghip_read_fasta_streams(const char *const *paths, size_t n, int threads,
                            std::vector<std::vector<uint8_t>> &streams, std::vector<ghip_genome_stats> &stats,
                            std::string &err);
int ghip_parse_fasta(const uint8_t *buf, size_t n, const char *path, uint8_t *out, size_t cap, size_t *out_len,
                     ghip_genome_stats &st, std::string &err);
bool ghip_slurp(const char *path, std::vector<uint8_t> &buf);
uint64_t ghip_stream_capacity_hint(const char *path);

// what one stage's file needs of another's
void ghip_ctx_release(ghip_ctx *ctx);   // api.cpp; called with ctx->mu NOT held: a handle was freed, the context goes with the last one
void ghip_free_sketches_locked(ghip_sketches *sk);   // api_sketches.cpp; ctx->mu held
int ghip_sketch_genomes_locked(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s, uint64_t seed, const ghip_seed_args *seeds,
                               ghip_sketches **out);   // api_sketches.cpp; ctx->mu held
int ghip_genomes_from_files_impl(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads, const uint64_t *known_caps,
                                 ghip_genomes **out);   // api_genomes.cpp

namespace ghip_api {


template <typename T>
inline int dmalloc(ghip_ctx *ctx, T **p, size_t count) {
    *p = (T *)ghip_pool_alloc(ctx, std::max<size_t>(count, 1) * sizeof(T));
    return *p ? GHIP_OK : GHIP_EHIP;
}

// Copies between pageable host memory and the device go through a pinned bounce buffer from 256 KiB on.  The runtime
// stages small pageable copies itself; a larger buffer it pins in place, and the NEXT copy or event wait of the process
// then stalls for 12-30 ms while it is unpinned (measured at 50 000 genomes: the 3.6 MB candidate list coming back made
// the first ANI round's 360 KB upload take 12-32 ms; scripts/ani_round_overhead.py).  One memcpy at host speed instead.
constexpr size_t GHIP_PIN_MIN = 256u << 10, GHIP_PIN_MAX = 32u << 20;
inline void *pinned_bounce(ghip_ctx *ctx, size_t bytes) {   // ctx->pin_mu held; nullptr: no pinned memory to be had (the caller copies directly)
    if (ctx->pin_bytes >= bytes) return ctx->pin_buf;
    if (ctx->pin_buf) { hipHostFree(ctx->pin_buf); ctx->pin_buf = nullptr; ctx->pin_bytes = 0; }
    const size_t want = std::min(bytes + bytes / 2, GHIP_PIN_MAX);
    if (hipHostMalloc(&ctx->pin_buf, want, hipHostMallocDefault) != hipSuccess) { ctx->pin_buf = nullptr; (void)hipGetLastError(); return nullptr; }
    ctx->pin_bytes = want;
    return ctx->pin_buf;
}

template <typename T>
inline int h2d(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
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
inline int h2d_nosync(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return GHIP_OK;
}
template <typename T>
inline int d2h_nosync(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return GHIP_OK;
}
inline int stream_sync(ghip_ctx *ctx) {
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GHIP_OK;
}

template <typename T>
inline int d2h(ghip_ctx *ctx, T *dst, const T *src, size_t count) {
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
inline double ghip_cpu_quota() {
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

inline uint64_t ghip_ani_seed_capacity(uint64_t len, uint32_t c) {
    const double m = (double)len / ((double)GHIP_ANI_SEGMENTS * (double)c);
    const uint64_t segcap = (uint64_t)(m + 6.0 * std::sqrt(m) + m / 4.0) + 24;
    return segcap * GHIP_ANI_SEGMENTS;
}

// fn(begin, end) over [0, n) on up to max_threads threads of at least min_per_thread items each (the caller's thread
// takes the last range); spawning costs ~30 us per thread, so short loops stay serial (callers pass min_per_thread = 10 000)
template <typename F>
inline void parallel_ranges(size_t n, size_t min_per_thread, size_t max_threads, F &&fn) {
    size_t t = std::min<size_t>({max_threads, n / std::max<size_t>(min_per_thread, 1), std::max<size_t>(1, std::thread::hardware_concurrency())});
    if (t <= 1) { fn((size_t)0, n); return; }
    const size_t per = (n + t - 1) / t;
    std::vector<std::thread> pool;
    for (size_t x = 0; x + 1 < t; x++) pool.emplace_back([&fn, x, per, n] { fn(std::min(n, x * per), std::min(n, (x + 1) * per)); });
    fn(std::min(n, (t - 1) * per), n);
    for (auto &th : pool) th.join();
}

inline uint32_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return (uint32_t)p;
}

// 1 - mash_distance exactly as the reference computes it (src/finch.rs:78-86 with finch's
// jaccard = common/total, mash = -ln(2j/(1+j))/k clamped to [0,1]; Rust f64::max/min drop NaN).
inline double finch_ani(uint64_t common, uint64_t total, uint32_t k) {
    double j = (double)common / (double)total;
    double mash = -1.0 * std::log((2.0 * j) / (1.0 + j)) / (double)k;
    double lo = std::isnan(mash) ? 0.0 : (mash > 0.0 ? mash : 0.0);  // f64::max(0, mash)
    double cl = lo < 1.0 ? lo : 1.0;                                   // f64::min(1, .)
    return 1.0 - cl;
}

// strtof(sprintf("%.2f", x)) for x in [0, 100]: hundredths -> f32 from a table built with strtof itself;
// the hundredth is found arithmetically unless x*100 is within 1e-6 of a rounding tie, where printf's exact
// decimal rounding is consulted.
inline float two_decimals_as_f32(double x) {
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
inline int h2d_on(ghip_ctx *ctx, hipStream_t st, T *dst, const T *src, size_t count) {
    if (count == 0) return GHIP_OK;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, st));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(st));
    return GHIP_OK;
}

inline int build_work(ghip_ctx *ctx, ghip_genomes *g, hipStream_t st = nullptr) {
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
inline int alloc_bases(ghip_ctx *ctx, ghip_genomes *g, hipStream_t st) {
    int rc = dmalloc(ctx, &g->d_packed, g->total_alloc / 16);
    if (rc) return rc;
    if ((rc = dmalloc(ctx, &g->d_valid, g->total_alloc / 32))) return rc;
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(g->d_valid, 0, g->total_alloc / 32 * sizeof(uint32_t), st));
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(g->d_packed, 0, g->total_alloc / 16 * sizeof(uint32_t), st));   // (tidy: keeps saved / compared images deterministic)
    return GHIP_OK;
}

// lays genomes out (first base at a multiple of GHIP_BASE_ALIGN, GHIP_TAIL_PAD invalid positions after each)
inline int layout_genomes(ghip_ctx *ctx, ghip_genomes *g, const std::vector<uint64_t> &lens) {
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
inline int upload_streams(ghip_ctx *ctx, ghip_genomes *g, hipStream_t st, const std::function<const uint8_t *(size_t)> &bytes_of) {
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


}  // namespace ghip_api

