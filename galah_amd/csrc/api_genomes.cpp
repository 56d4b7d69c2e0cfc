// C-ABI of libgalah_hip.so, genome ingest: files / host streams / synthetic genomes -> the resident form (ghip_genomes).
#include "api_internal.h"

using namespace ghip_api;

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
    ghip_ctx_release(ctx);
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
int ghip_genomes_from_files_impl(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads, const uint64_t *known_caps,
                                   ghip_genomes **out);

extern "C" int ghip_genomes_from_files(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads,
                                       ghip_genomes **out) {
    return ghip_genomes_from_files_impl(ctx, paths, n, io_threads, nullptr, out);
}

// known_caps (nullable): the capacity hints of the files, already looked up by the caller (one stat per file is 0.1-0.2 s
// for 100 000 contig files: not twice)
int ghip_genomes_from_files_impl(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads, const uint64_t *known_caps,
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
        std::vector<uint8_t> outgrown(n, 0);   // files whose stream is longer than their hint promised (a multi-member gzip that is not BGZF)
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
        // The device-side gzip path first (ingest_gz.cpp; ghip_options.gz_device = N): files named *.gz go to the device
        // compressed and are inflated, checked, parsed and packed there -- those of which the call holds N files' worth;
        // gz_done marks the ones it ingested, the workers below take the rest: plain files, and whatever that path declined.
        std::vector<uint8_t> gz_done(n, 0);
        if (opt.gz_device && n_gz >= opt.gz_device) {
            std::vector<size_t> cand;
            for (size_t i = 0; i < n; i++) { const size_t l = strlen(paths[i]); if (l > 3 && !strcmp(paths[i] + l - 3, ".gz")) cand.push_back(i); }
            const int grc = ghip_ingest_gz_device(ctx, g, paths, cap, cand, threads, gz_done);
            if (grc != GHIP_OK) { free_genomes_locked(g); return grc; }
        }
        // work units: a file, or a run of consecutive SMALL files shipped as one group (packed form only)
        std::vector<std::pair<size_t, size_t>> units;
        {
            constexpr uint64_t SMALL_FILE = 256u << 10, GROUP_BASES = 4u << 20;
            constexpr size_t GROUP_FILES = 512;
            const bool grouping = packed_mode && n_slots && opt.ingest_groups;
            for (size_t i = 0; i < n;) {
                if (gz_done[i]) { i++; continue; }
                size_t j = i + 1;
                if (grouping && cap[i] <= SMALL_FILE) {
                    uint64_t bases = next_genome_offset(0, cap[i]);
                    while (j < n && !gz_done[j] && j - i < GROUP_FILES && cap[j] <= SMALL_FILE && bases + next_genome_offset(0, cap[j]) <= std::min<uint64_t>(GROUP_BASES, slot_bytes)) {
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
                    else if (len > cap[i]) { over = true; outgrown[i] = 1; }  // capacity hint too small (multi-member gzip): this file is placed afterwards (below)
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
                if (status.load() != GHIP_OK) break;
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
        if (rc == GHIP_OK && overflow) {
            // Some streams outgrew their places.  Only THOSE files are read again (exact lengths first, as the two-phase form does
            // for every file); the genomes already resident move into a layout made with the exact lengths by device-to-device
            // copies of whole 64-base groups -- not, as until round 5, the whole call repeated through host vectors (one
            // concatenated gzip among 10 000 genomes: 50 GB of them).
            ctx->ingest_repeats++;
            std::vector<size_t> again;
            for (size_t i = 0; i < n; i++) if (outgrown[i]) again.push_back(i);
            // The outgrown files' texts pass through host vectors -- a group of `threads` files at a time, so that a call with
            // many large concatenated gzips never holds all of them at once.  Their exact lengths come first (the layout needs
            // every length); texts are kept for the upload while they fit KEEP_BYTES together, the rest is read a second time.
            const uint64_t KEEP_BYTES = opt.fault_stage == GHIP_FAULT_GZ_SMALL_BATCHES ? 0 : 1ull << 30;   // (tests: every text is read twice)
            const size_t group = (size_t)std::max(1, threads);
            std::vector<std::vector<uint8_t>> kept(again.size());
            std::vector<uint8_t> have(again.size(), 0);
            std::string e2;
            auto read_group = [&](size_t lo, size_t hi, std::vector<std::vector<uint8_t>> &streams, std::vector<ghip_genome_stats> &stats) {
                std::vector<const char *> gp;
                for (size_t x = lo; x < hi; x++) gp.push_back(paths[again[x]]);
                return ghip_read_fasta_streams(gp.data(), gp.size(), threads, streams, stats, e2);
            };
            {
                uint64_t kept_bytes = 0;
                for (size_t lo = 0; lo < again.size() && rc == GHIP_OK; lo += group) {
                    const size_t hi = std::min(again.size(), lo + group);
                    std::vector<std::vector<uint8_t>> streams;
                    std::vector<ghip_genome_stats> stats;
                    if ((rc = read_group(lo, hi, streams, stats)) != GHIP_OK) break;
                    for (size_t x = lo; x < hi; x++) {
                        g->lens[again[x]] = streams[x - lo].size();
                        g->stats[again[x]] = stats[x - lo];
                        if (kept_bytes + streams[x - lo].size() <= KEEP_BYTES) { kept_bytes += streams[x - lo].size(); kept[x] = std::move(streams[x - lo]); have[x] = 1; }
                    }
                }
            }
            if (rc != GHIP_OK) ghip_set_error(ctx, rc, e2);
            uint32_t *old_packed = g->d_packed, *old_valid = g->d_valid;
            std::vector<uint64_t> old_starts = g->starts;
            if (rc == GHIP_OK) {
                uint64_t off = 0;
                for (size_t i = 0; i < n; i++) { g->starts[i] = off; off = next_genome_offset(off, g->lens[i]); }
                g->total_alloc = off + 256;
                g->d_packed = nullptr; g->d_valid = nullptr;
                rc = alloc_bases(ctx, g, fill_stream);
            }
            uint64_t *d_runs = nullptr;
            if (rc == GHIP_OK) {
                std::vector<uint64_t> runs;   // (source word, destination word, words) of the packed array; the validity bitmap's are half of each
                for (size_t i = 0; i < n; i++) {
                    if (outgrown[i] || g->lens[i] == 0) continue;
                    const uint64_t groups = (g->lens[i] + 63) / 64;
                    runs.push_back(old_starts[i] / 16); runs.push_back(g->starts[i] / 16); runs.push_back(groups * 4);
                }
                const size_t n_runs = runs.size() / 3;
                for (size_t x = 0; x < 3 * n_runs; x++) runs.push_back(runs[x] / 2);
                if (n_runs && (rc = dmalloc(ctx, &d_runs, runs.size())) == GHIP_OK && (rc = h2d_on(ctx, fill_stream, d_runs, runs.data(), runs.size())) == GHIP_OK) {
                    ghip_launch_copy_runs(fill_stream, old_packed, g->d_packed, d_runs, n_runs);
                    ghip_launch_copy_runs(fill_stream, old_valid, g->d_valid, d_runs + 3 * n_runs, n_runs);
                    if (hipGetLastError() != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "ingest re-layout failed");
                }
            }
            if (rc == GHIP_OK) {   // the outgrown files into their new places, through one staging buffer
                uint64_t longest = 0;
                for (size_t i : again) longest = std::max<uint64_t>(longest, g->lens[i]);
                uint8_t *d_stage = nullptr;
                auto place = [&](size_t x, const std::vector<uint8_t> &text) {
                    if (text.empty()) return;
                    if (text.size() != g->lens[again[x]]) { rc = ghip_set_error(ctx, GHIP_EIO, std::string(paths[again[x]]) + ": changed while it was being read"); return; }
                    // (the staging buffer is reused: the copy before this one has been packed when the stream gets here, and the host
                    // vector is pageable -- the runtime has taken its bytes when hipMemcpyAsync returns)
                    if (hipMemcpyAsync(d_stage, text.data(), text.size(), hipMemcpyHostToDevice, fill_stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "base upload failed");
                    else ghip_launch_pack_bases(fill_stream, d_stage, text.size(), g->starts[again[x]], g->d_packed, g->d_valid);
                };
                if (longest && (rc = dmalloc(ctx, &d_stage, longest + 64)) == GHIP_OK) {
                    for (size_t lo = 0; lo < again.size() && rc == GHIP_OK; lo += group) {
                        const size_t hi = std::min(again.size(), lo + group);
                        bool all = true;
                        for (size_t x = lo; x < hi; x++) all = all && have[x];
                        std::vector<std::vector<uint8_t>> streams;
                        std::vector<ghip_genome_stats> stats;
                        if (!all && (rc = read_group(lo, hi, streams, stats)) != GHIP_OK) { ghip_set_error(ctx, rc, e2); break; }
                        for (size_t x = lo; x < hi && rc == GHIP_OK; x++) place(x, have[x] ? kept[x] : streams[x - lo]);
                        // a group's re-read vectors die here: their copies must have left the host first
                        if (!all && hipStreamSynchronize(fill_stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "base upload failed");
                    }
                }
                if (hipStreamSynchronize(fill_stream) != hipSuccess || hipGetLastError() != hipSuccess) { if (rc == GHIP_OK) rc = ghip_set_error(ctx, GHIP_EHIP, "ingest re-layout failed"); }
                ghip_pool_free(ctx, d_stage);
            } else hipStreamSynchronize(fill_stream);
            ghip_pool_free(ctx, d_runs);
            if (old_packed != g->d_packed) { ghip_pool_free(ctx, old_packed); ghip_pool_free(ctx, old_valid); }
            overflow = false;
        }
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
    return ghip_set_error(ctx, GHIP_EHIP, "ingest: a stream is still marked outgrown after the re-layout (internal error)");   // (not reached)
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

