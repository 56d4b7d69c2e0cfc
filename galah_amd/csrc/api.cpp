// C-ABI of libgalah_hip.so (include/galah_hip.h), core: errors, the device-memory pool, the I/O worker pool, per-kernel timing,
// options, the context.  The stages live in api_genomes.cpp, api_sketches.cpp, api_pairs.cpp, api_ani.cpp.
// Host orchestration only -- all data-parallel work is in sketch.hip / pairs*.hip / ani.hip.
#include "api_internal.h"

using namespace ghip_api;

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
    const bool exact = ghip_dbg(ctx->opt, GHIP_DEBUG_POOL_EXACT);   // (a memory checker's run: no slack behind any buffer)
    bytes = exact ? std::max<size_t>(bytes, 1) : std::max<size_t>((bytes + 255) / 256 * 256, 256);
    std::lock_guard<std::mutex> pl(ctx->pool_mu);
    ghip_pool_block *best = nullptr;
    for (auto &b : ctx->pool)
        if (!b.used && b.bytes >= bytes && b.bytes <= (exact ? bytes : 2 * bytes + (1u << 20)) && (!best || b.bytes < best->bytes)) best = &b;
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

void ghip_ctx_release(ghip_ctx *ctx) {  // called with ctx->mu NOT held
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
    for (auto &sl : ctx->gz_slots) { if (sl.ev) hipEventDestroy(sl.ev); if (sl.p) hipHostFree(sl.p); }
    for (uint8_t *p : ctx->gz_results) if (p) hipHostFree(p);
    if (ctx->gz_rec_host) hipHostFree(ctx->gz_rec_host);
    for (hipEvent_t e : ctx->gz_ev) if (e) hipEventDestroy(e);
    if (ctx->gz_stream) hipStreamDestroy(ctx->gz_stream);
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
              (set("GHIP_COMM_DEBUG") ? GHIP_DEBUG_COMM : 0) | (set("GHIP_CLUSTER_DEBUG") ? GHIP_DEBUG_CLUSTER : 0) | (set("GHIP_ANI_DEBUG") ? GHIP_DEBUG_ANI : 0) |
              (set("GHIP_POOL_EXACT") ? GHIP_DEBUG_POOL_EXACT : 0);
    o.pair_debug = num("GHIP_PAIR_DEBUG", 0);
    o.probe_arranged = num("GHIP_PROBE_ARRANGED", 0);   // (likewise off until measured)
    o.comm_timeout_ms = num("GHIP_COMM_TIMEOUT_MS", 0);   // (no deadline unless asked for: it also bounds how late a HEALTHY peer may be, comm.cpp rccl_wait)
    o.gz_device = num("GHIP_GZ_DEVICE", 0);   // (likewise)
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
        o.copy_streams > 4 || o.fault_stage > GHIP_FAULT_GZ_SMALL_BATCHES || o.probe_arranged > 4)
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
        if (rc) return ghip_set_error(ctx, rc, "ghip_set_options: a field is out of range or struct_size is wrong");
        if (ctx->opt.gz_device == 0 && !ctx->gz_slots.empty()) {
            // the device-side gzip path is off again: its pinned memory (2 x 12 slots of 16 MiB and the result buffers) goes back
            // to the host; no ingest is under way (this thread holds the context)
            hipSetDevice(ctx->device);
            for (auto &sl : ctx->gz_slots) { if (sl.ev) hipEventDestroy(sl.ev); if (sl.p) hipHostFree(sl.p); }
            ctx->gz_slots.clear();
            for (int x = 0; x < 2; x++) { if (ctx->gz_results[x]) hipHostFree(ctx->gz_results[x]); ctx->gz_results[x] = nullptr; ctx->gz_results_bytes[x] = 0; }
            if (ctx->gz_rec_host) hipHostFree(ctx->gz_rec_host);
            ctx->gz_rec_host = nullptr; ctx->gz_rec_host_bytes = 0;
        }
        return GHIP_OK;
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
    ghip_ctx_release(ctx);  // deferred until the last genomes/sketches/index handle is freed
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

extern "C" int ghip_ingest_counters(ghip_ctx *ctx, uint64_t out[4]) {
    if (!ctx || !out) return GHIP_EINVAL;
    out[0] = ctx->gz_device_files.load();
    out[1] = ctx->gz_host_files.load();
    out[2] = ctx->gz_device_us.load();
    out[3] = ctx->ingest_repeats.load();
    return GHIP_OK;
}

extern "C" int ghip_selftest_hash_floor(ghip_ctx *ctx, uint64_t wave_positions, double *out_ms) {
    if (!ctx || !out_ms || wave_positions == 0) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return ghip_launch_hash_floor(ctx, wave_positions, out_ms);
}

extern "C" void ghip_free(void *p) { free(p); }

