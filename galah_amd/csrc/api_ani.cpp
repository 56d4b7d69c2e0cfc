// C-ABI of libgalah_hip.so, ANI stage: the index (build, fused with sketching, from files in HBM-bounded batches), ghip_ani_pairs.
#include "api_internal.h"

using namespace ghip_api;

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

extern "C" uint32_t ghip_ani_definition_version(void) { return GHIP_ANI_DEFINITION_VERSION; }

extern "C" void ghip_ani_index_free(ghip_ani_index *idx) {
    if (!idx) return;
    ghip_ctx *ctx = idx->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStreamSynchronize(ctx->stream);
        free_index_locked(idx);
    }
    ghip_ctx_release(ctx);
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
    // (exactly what was asked of dmalloc: with no genome the arrays are one element long, not GHIP_ANI_SEGMENTS)
    if (n) GHIP_HIP_CHECK(ctx, hipMemsetAsync(idx->d_seg_count, 0, n * GHIP_ANI_SEGMENTS * sizeof(uint32_t), ctx->stream));
    if (idx->chunk_start[n]) GHIP_HIP_CHECK(ctx, hipMemsetAsync(idx->d_chunk_total, 0, idx->chunk_start[n] * sizeof(uint32_t), ctx->stream));
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
            rc = ghip_sketch_genomes_locked(ctx, g, k, s, seed, &sa, &sk);
        }
        bool overflow = false;
        if (!rc) rc = index_check_seeds(ctx, idx, cap, &overflow);
        if (!rc && overflow) rc = index_seed_standalone(ctx, g, idx, cap);  // exact counts now known
    } else {
        rc = ghip_sketch_genomes_locked(ctx, g, k, s, seed, nullptr, &sk);
        if (!rc) rc = index_seed_standalone(ctx, g, idx, cap);
    }
    if (!rc) rc = index_finish(ctx, idx, ctx->opt.overlap_binning != 0);   // the binning overlaps the caller's pair stage
    if (rc) { if (sk) ghip_free_sketches_locked(sk); free_index_locked(idx); return rc; }
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
        const int rc = ghip_genomes_from_files_impl(ctx, paths + ranges[b].first, ranges[b].second - ranges[b].first, io_threads, hints.data() + ranges[b].first, &g);
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
    for (auto *b : sks) ghip_free_sketches_locked(b);
    for (auto *b : idxs) free_index_locked(b);
    if (rc) { ghip_free_sketches_locked(sk); if (idx) free_index_locked(idx); return rc; }
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

