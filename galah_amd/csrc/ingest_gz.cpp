// Host driver of the device-side gzip path (gz_inflate.hip; ghip_options.gz_device): the .gz files of a
// ghip_genomes_from_files call go to the device as they are on disk -- a third of the bytes of their text -- in batches of
// up to 4 096 files, and come back as resident genomes plus their assembly statistics.  Per batch:
//
//   reader threads (<= 12, the context's I/O pool)   file -> pinned slot (16 MiB, two per thread) -> hipMemcpyAsync into the
//                                                   batch's input area, a group of consecutive files per copy
//   one stream of its own (ctx->gz_stream)          job table up, the six launches of ghip_launch_gz_batch, job table + record
//                                                   pool back into pinned memory, an event
//   host, once the event has fired                  genome lengths, contig counts, ambiguous bases, N50 from the record table;
//                                                   files with a verdict other than GHIP_GZ_OK stay with the host path
//
// Two batches are in flight: batch b + 1 is read and shipped while the kernels of batch b run -- the readers' copies go over the
// context's copy streams, the kernels over a stream of their own, so that staging b + 1 (which waits for its copies) does not
// wait for b's kernels.  A batch's text is bounded by a quarter of the device memory that is free when the call starts (two in
// flight: half), so that the sketch pass of the previous batch of FILES, which allocates from the same pool on another thread
// (ghip_sketch_and_index_files), still finds room.  Nothing here decides that a
// file is bad: whatever the device declines is ingested by ingest.cpp's inflate + parser afterwards, which words the errors
// (reference behaviour: needletail's reader behind src/finch.rs:69).
#include "api_internal.h"
#include "gz_common.h"

#include <memory>
#include <set>

namespace {

constexpr size_t GZ_SLOT_BYTES = 16u << 20;      // a pinned slot; a compressed image larger than this is left to the host
constexpr size_t GZ_MAX_READERS = 12;   // (read() of ~1.5 MB files from the page cache: ~3 GB/s per thread; two 16 MiB pinned slots each)
constexpr size_t GZ_BATCH_FILES = 4096;          // one wavefront each: 16 per CU x 256 CUs
constexpr uint64_t GZ_BATCH_TEXT = 24ull << 30;  // bytes of text per batch: 4 096 genomes of 5 Mb (a batch the pool has no room for is halved)
constexpr size_t GZ_SPLIT_MIN = 64;              // ... down to this many files, below which the host takes them
constexpr uint64_t GZ_REC_PER_TEXT = 48;         // the record pool of a batch: four entries per file and one per this many bytes of its text (a fragmented
                                                 // assembly has a contig per few kb, a file of 30-base primers one per ~50 bytes; a file whose records do
                                                 // not fit is the host's) -- 8 % of the texts' memory

struct Unit { size_t first, last; uint64_t in_off, bytes; };   // jobs [first, last): their images lie together at in_off

struct Batch {
    std::vector<size_t> files;          // index into the call's paths, per job
    std::vector<ghip_gz_job> jobs;
    std::vector<uint32_t> chunk_start;
    std::vector<Unit> units;
    uint64_t in_bytes = 0, text_bytes = 0, max_text_cap = 0;
    size_t n_chunks = 0;
    uint32_t rec_room = 0;
    uint8_t *d_in = nullptr, *d_text = nullptr;
    ghip_gz_job *d_jobs = nullptr;
    uint32_t *d_chunk_start = nullptr, *d_rec = nullptr;   // d_rec[0] = entries handed out, the pool from d_rec + 4
    void *d_chunks = nullptr;
    int side = 0;                       // which pinned result buffer / event triple
    bool launched = false, no_room = false;
};

void free_batch(ghip_ctx *ctx, Batch &b) {
    ghip_pool_free(ctx, b.d_in); ghip_pool_free(ctx, b.d_text); ghip_pool_free(ctx, b.d_jobs);
    ghip_pool_free(ctx, b.d_chunk_start); ghip_pool_free(ctx, b.d_chunks); ghip_pool_free(ctx, b.d_rec);
    b.d_in = b.d_text = nullptr; b.d_jobs = nullptr; b.d_chunk_start = b.d_rec = nullptr; b.d_chunks = nullptr;
}

inline uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

}  // namespace

int ghip_ingest_gz_device(ghip_ctx *ctx, ghip_genomes *g, const char *const *paths, const std::vector<uint64_t> &cap, const std::vector<size_t> &cand,
                          int io_threads, std::vector<uint8_t> &done) {
    const bool dbg = ghip_dbg(ctx->opt, GHIP_DEBUG_INGEST);
    const auto w0 = std::chrono::steady_clock::now();
    if (cand.empty() || !ctx->copy_stream[0] || ctx->n_copy_streams < 1) return GHIP_OK;
    if (!ctx->gz_stream && hipStreamCreateWithFlags(&ctx->gz_stream, hipStreamNonBlocking) != hipSuccess) { ctx->gz_stream = nullptr; (void)hipGetLastError(); }
    hipStream_t ks = ctx->gz_stream;
    if (!ks) { ctx->gz_host_files += cand.size(); return GHIP_OK; }   // (no stream to be had: the host path takes them all)
    // what a batch may hold: a quarter of what is free now, counting the pool's idle blocks (they are ours to reuse)
    uint64_t batch_text = GZ_BATCH_TEXT;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            uint64_t idle = 0;
            { std::lock_guard<std::mutex> pl(ctx->pool_mu); for (auto &blk : ctx->pool) if (!blk.used) idle += blk.bytes; }
            batch_text = std::min<uint64_t>(GZ_BATCH_TEXT, std::max<uint64_t>(((uint64_t)free_b + idle) / 4, 256ull << 20));
        } else (void)hipGetLastError();
    }
    // ---- what the path takes: a readable image that fits a slot, a text below 4 GiB (the trailer's ISIZE is all there is to go by)
    std::vector<std::pair<size_t, uint64_t>> take;   // (file, bytes on disk)
    for (size_t i : cand) {
        struct stat st;
        // (deflate cannot expand more than 1032-fold: a trailer that promises more is damaged, and sizing buffers by it would cost gigabytes)
        if (stat(paths[i], &st) != 0 || st.st_size < 18 || (uint64_t)st.st_size > GZ_SLOT_BYTES || cap[i] == 0 || cap[i] > 0xffffffffull ||
            cap[i] - 1 > (uint64_t)st.st_size * 1032 + 1024) { ctx->gz_host_files++; continue; }
        take.push_back({i, (uint64_t)st.st_size});
    }
    // One wavefront inflates one file, so a launch takes as long as its largest file (~0.03 us per byte of text, modelled), while the
    // host's threads take the files one after the other: the device is the better place for a file only when the call holds many
    // files' worth of it.  gz_device = N: a file goes to the device when the call's device-bound files together hold at least N times
    // its text (N files of one size: the plain count; a 100 Mb assembly among bacterial genomes: left to the host).
    {
        std::sort(take.begin(), take.end(), [&](const std::pair<size_t, uint64_t> &a, const std::pair<size_t, uint64_t> &b) { return cap[a.first] > cap[b.first]; });
        uint64_t sum = 0;
        for (auto &tk : take) sum += cap[tk.first];
        size_t drop = 0;
        while (drop < take.size() && cap[take[drop].first] * (uint64_t)ctx->opt.gz_device > sum) { sum -= cap[take[drop].first]; drop++; }
        ctx->gz_host_files += drop;
        take.erase(take.begin(), take.begin() + drop);
        std::sort(take.begin(), take.end());   // back to the order of the call (the layout's order: consecutive files, consecutive places)
    }
    if (take.empty()) return GHIP_OK;
    // ---- pinned memory and events, kept by the context
    const size_t readers = std::min<size_t>(std::max(1, io_threads), GZ_MAX_READERS);
    if (ctx->gz_slots.size() < 2 * readers) ctx->gz_slots.resize(2 * readers);
    for (size_t x = 0; x < 2 * readers; x++) {
        ghip_ctx::gz_slot &sl = ctx->gz_slots[x];
        if (!sl.p && hipHostMalloc((void **)&sl.p, GZ_SLOT_BYTES, hipHostMallocDefault) != hipSuccess) { sl.p = nullptr; (void)hipGetLastError(); }
        if (!sl.ev && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) { sl.ev = nullptr; (void)hipGetLastError(); }
        sl.inflight = false;
        if (!sl.p || !sl.ev) { ctx->gz_host_files += take.size(); return GHIP_OK; }   // no pinned memory: the host path takes them all
    }
    for (hipEvent_t &e : ctx->gz_ev)
        if (!e && hipEventCreate(&e) != hipSuccess) { e = nullptr; (void)hipGetLastError(); ctx->gz_host_files += take.size(); return GHIP_OK; }

    // ---- batches: runs [lo, hi) of `take`, laid out when their turn comes (a run the pool has no room for is halved)
    const bool small_pool = ctx->opt.fault_stage == GHIP_FAULT_GZ_SMALL_BATCHES;   // (tests: a record pool that a file of many short records outgrows)
    auto lay_out = [&](size_t lo, size_t hi) {
        Batch b;
        for (size_t t = lo; t < hi; t++) {
            const size_t i = take[t].first;
            const uint64_t text_cap = cap[i] - 1;
            ghip_gz_job j{};
            j.in_off = b.in_bytes;
            j.text_off = b.text_bytes;
            j.gbase = g->starts[i];
            j.in_len = (uint32_t)take[t].second;
            j.text_cap = (uint32_t)text_cap;
            j.stream_cap = (uint32_t)std::min<uint64_t>(cap[i], 0xffffffffull);
            j.status = GHIP_GZ_ENOTRUN;
            b.chunk_start.push_back((uint32_t)b.n_chunks);
            b.n_chunks += ghip_gz_chunks_of(text_cap);
            b.in_bytes += up(j.in_len, 16);
            b.text_bytes += up(text_cap, 64) + 64;
            b.max_text_cap = std::max<uint64_t>(b.max_text_cap, text_cap);
            b.files.push_back(i);
            b.jobs.push_back(j);
        }
        for (size_t f = 0; f < b.jobs.size();) {   // groups of consecutive images that fill a pinned slot
            Unit u{f, f, b.jobs[f].in_off, 0};
            while (u.last < b.jobs.size() && u.bytes + up(b.jobs[u.last].in_len, 16) <= GZ_SLOT_BYTES) { u.bytes += up(b.jobs[u.last].in_len, 16); u.last++; }
            b.units.push_back(u);
            f = u.last;
        }
        uint64_t room = 0;
        for (const ghip_gz_job &j : b.jobs) room += small_pool ? 1 + j.text_cap / 4096 : std::min<uint64_t>(j.text_cap / 2 + 1, 4 + j.text_cap / GZ_REC_PER_TEXT);
        b.rec_room = (uint32_t)std::min<uint64_t>(room, 0x3fffffffull);
        return b;
    };
    const bool small = ctx->opt.fault_stage == GHIP_FAULT_GZ_SMALL_BATCHES;   // (tests: several batches in flight, runs that find no room)
    const size_t batch_files = small ? 3 : GZ_BATCH_FILES, split_min = small ? 1 : GZ_SPLIT_MIN;
    const bool pretend_no_room = small && ctx->opt.fault_rank == 1;
    std::set<std::pair<size_t, size_t>> pretended;
    std::vector<std::pair<size_t, size_t>> todo;   // a stack: the next run on top
    for (size_t t = 0; t < take.size();) {
        size_t hi = t;
        uint64_t text = 0;
        while (hi < take.size() && hi - t < batch_files && (hi == t || text + up(cap[take[hi].first] - 1, 64) + 64 <= batch_text)) { text += up(cap[take[hi].first] - 1, 64) + 64; hi++; }
        todo.push_back({t, hi});
        t = hi;
    }
    std::reverse(todo.begin(), todo.end());
    size_t n_batches = 0;

    std::atomic<int> hip_failed{0};
    double t_read = 0, t_wait = 0;
    uint64_t verdicts[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, diag[6] = {0, 0, 0, 0, 0, 0};   // diag: text bytes, tokens, matches, batches, rounds, blocks

    auto stage = [&](Batch &b) {
        const size_t nj = b.jobs.size(), jobs_bytes = nj * sizeof(ghip_gz_job), rec_bytes = (4 + (size_t)b.rec_room) * sizeof(uint32_t);
        // the pinned landing place of the job table and the record pool's head (the pool's used part follows in finish())
        if (ctx->gz_results_bytes[b.side] < jobs_bytes + 16) {
            if (ctx->gz_results[b.side]) hipHostFree(ctx->gz_results[b.side]);
            ctx->gz_results[b.side] = nullptr; ctx->gz_results_bytes[b.side] = 0;
            const size_t want = std::max(jobs_bytes, GZ_BATCH_FILES * sizeof(ghip_gz_job)) + 16;
            if (hipHostMalloc((void **)&ctx->gz_results[b.side], want, hipHostMallocDefault) == hipSuccess) ctx->gz_results_bytes[b.side] = want;
            else { ctx->gz_results[b.side] = nullptr; (void)hipGetLastError(); return; }
        }
        b.d_in = (uint8_t *)ghip_pool_alloc(ctx, b.in_bytes + 1024);
        b.d_text = (uint8_t *)ghip_pool_alloc(ctx, b.text_bytes + 64);
        b.d_jobs = (ghip_gz_job *)ghip_pool_alloc(ctx, jobs_bytes);
        b.d_chunk_start = (uint32_t *)ghip_pool_alloc(ctx, nj * sizeof(uint32_t));
        b.d_chunks = ghip_pool_alloc(ctx, std::max<size_t>(b.n_chunks, 1) * ghip_gz_chunk_bytes());
        b.d_rec = (uint32_t *)ghip_pool_alloc(ctx, rec_bytes);
        if (!b.d_in || !b.d_text || !b.d_jobs || !b.d_chunk_start || !b.d_chunks || !b.d_rec) { free_batch(ctx, b); b.no_room = true; return; }
        // ---- the images: disk -> pinned slot -> input area
        std::atomic<size_t> next{0};
        std::mutex tmu;
        ctx->io.run((int)readers, [&](int me) {
            if (hipSetDevice(ctx->device) != hipSuccess) { hip_failed = 1; return; }
            hipStream_t cs = ctx->copy_stream[me % ctx->n_copy_streams];
            double rd = 0, wt = 0;
            for (size_t turn = 0;; turn++) {
                const size_t u = next.fetch_add(1);
                if (u >= b.units.size()) break;
                const Unit &un = b.units[u];
                ghip_ctx::gz_slot &sl = ctx->gz_slots[2 * (size_t)me + (turn & 1)];
                const auto t0 = std::chrono::steady_clock::now();
                if (sl.inflight && hipEventSynchronize(sl.ev) != hipSuccess) { (void)hipGetLastError(); hip_failed = 1; }
                sl.inflight = false;
                const auto t1 = std::chrono::steady_clock::now();
                for (size_t j = un.first; j < un.last; j++) {
                    ghip_gz_job &job = b.jobs[j];
                    uint8_t *dst = sl.p + (job.in_off - un.in_off);
                    const size_t want = job.in_len;
                    size_t got = 0;
                    if (FILE *f = fopen(paths[b.files[j]], "rb")) { got = fread(dst, 1, want, f); fclose(f); }
                    if (got != want) job.in_len = 0;   // unreadable or changed under us: the kernel declines it, the host path reports it
                    memset(dst + got, 0, (size_t)up(want, 16) - got);
                }
                const auto t2 = std::chrono::steady_clock::now();
                if (hipMemcpyAsync(b.d_in + un.in_off, sl.p, un.bytes, hipMemcpyHostToDevice, cs) != hipSuccess || hipEventRecord(sl.ev, cs) != hipSuccess) {
                    (void)hipGetLastError(); hip_failed = 1;
                } else sl.inflight = true;
                rd += std::chrono::duration<double>(t2 - t1).count();
                wt += std::chrono::duration<double>(t1 - t0).count();
            }
            std::lock_guard<std::mutex> l(tmu);
            t_read += rd; t_wait += wt;
        });
        for (int x = 0; x < ctx->n_copy_streams; x++)
            if (ctx->copy_stream[x] && hipStreamSynchronize(ctx->copy_stream[x]) != hipSuccess) { (void)hipGetLastError(); hip_failed = 1; }
        for (auto &sl : ctx->gz_slots) sl.inflight = false;
        if (hip_failed.load()) { free_batch(ctx, b); return; }
        // ---- the device's part, and its results on their way back
        uint8_t *res = ctx->gz_results[b.side];
        hipEvent_t ev_begin = ctx->gz_ev[3 * b.side], ev_end = ctx->gz_ev[3 * b.side + 1], ev_landed = ctx->gz_ev[3 * b.side + 2];
        bool ok = hipMemcpyAsync(b.d_jobs, b.jobs.data(), jobs_bytes, hipMemcpyHostToDevice, ks) == hipSuccess &&
                  hipMemcpyAsync(b.d_chunk_start, b.chunk_start.data(), nj * sizeof(uint32_t), hipMemcpyHostToDevice, ks) == hipSuccess &&
                  hipMemsetAsync(b.d_rec, 0, 16, ks) == hipSuccess && hipEventRecord(ev_begin, ks) == hipSuccess;
        if (ok) {
            ghip_launch_gz_batch(ks, b.d_in, b.d_text, b.d_jobs, (uint32_t)nj, b.max_text_cap, b.d_chunk_start, b.d_chunks, b.d_rec, b.d_rec + 4, b.rec_room, g->d_packed,
                                 g->d_valid);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(ev_end, ks) == hipSuccess &&
                 hipMemcpyAsync(res, b.d_jobs, jobs_bytes, hipMemcpyDeviceToHost, ks) == hipSuccess &&
                 hipMemcpyAsync(res + jobs_bytes, b.d_rec, 16, hipMemcpyDeviceToHost, ks) == hipSuccess && hipEventRecord(ev_landed, ks) == hipSuccess;
        }
        if (!ok) { (void)hipGetLastError(); hip_failed = 1; hipStreamSynchronize(ks); free_batch(ctx, b); return; }
        b.launched = true;
    };

    auto finish = [&](Batch &b) {
        if (!b.launched) { ctx->gz_host_files += b.files.size(); return; }
        if (hipEventSynchronize(ctx->gz_ev[3 * b.side + 2]) != hipSuccess) { (void)hipGetLastError(); hip_failed = 1; hipStreamSynchronize(ks); free_batch(ctx, b); return; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->gz_ev[3 * b.side], ctx->gz_ev[3 * b.side + 1]) == hipSuccess) ctx->gz_device_us += (uint64_t)(ms * 1000.0f);
        const size_t nj = b.jobs.size();
        const ghip_gz_job *jobs = reinterpret_cast<const ghip_gz_job *>(ctx->gz_results[b.side]);
        // the used part of the record pool, over the stream the next batch's kernels are NOT queued on (they may be running)
        const uint32_t rec_used = std::min(reinterpret_cast<const uint32_t *>(ctx->gz_results[b.side] + nj * sizeof(ghip_gz_job))[0], b.rec_room);
        if (rec_used) {
            if (ctx->gz_rec_host_bytes < (size_t)rec_used * 4) {
                if (ctx->gz_rec_host) hipHostFree(ctx->gz_rec_host);
                ctx->gz_rec_host = nullptr; ctx->gz_rec_host_bytes = 0;
                const size_t want = std::max<size_t>((size_t)rec_used * 4 * 2, 4u << 20);
                if (hipHostMalloc((void **)&ctx->gz_rec_host, want, hipHostMallocDefault) == hipSuccess) ctx->gz_rec_host_bytes = want;
                else { ctx->gz_rec_host = nullptr; (void)hipGetLastError(); }
            }
            hipStream_t rs = ctx->copy_stream[0];
            if (!ctx->gz_rec_host || hipMemcpyAsync(ctx->gz_rec_host, b.d_rec + 4, (size_t)rec_used * 4, hipMemcpyDeviceToHost, rs) != hipSuccess || hipStreamSynchronize(rs) != hipSuccess) {
                (void)hipGetLastError(); hip_failed = 1; hipStreamSynchronize(ks); free_batch(ctx, b); return;
            }
        }
        const uint32_t *rec = reinterpret_cast<const uint32_t *>(ctx->gz_rec_host);
        std::vector<uint64_t> lengths;
        for (size_t j = 0; j < nj; j++) {
            const ghip_gz_job &r = jobs[j];
            verdicts[std::min<uint32_t>(r.status, 8)]++;
            diag[0] += r.text_len; diag[1] += r.tokens; diag[2] += r.matches; diag[3] += r.batches; diag[4] += r.rounds; diag[5] += r.blocks;
            if (r.status != GHIP_GZ_OK || (uint64_t)r.rec_off + r.records > rec_used) { ctx->gz_host_files++; continue; }
            const size_t i = b.files[j];
            g->lens[i] = r.stream_len;
            ghip_genome_stats &st = g->stats[i];
            st = ghip_genome_stats();
            st.num_contigs = r.records;
            st.num_ambiguous_bases = r.ambiguous;
            // reference src/genome_stats.rs:33-45: ascending record lengths, the first whose running sum reaches half the total
            lengths.resize(r.records);
            for (uint32_t q = 0; q < r.records; q++) lengths[q] = (uint64_t)(q + 1 < r.records ? rec[r.rec_off + q + 1] : r.seq_bytes) - rec[r.rec_off + q];
            std::sort(lengths.begin(), lengths.end());
            uint64_t total = 0, run = 0;
            for (uint64_t l : lengths) total += l;
            for (uint64_t l : lengths) {
                run += l;
                if (run >= total / 2) { st.n50 = l; break; }
            }
            done[i] = 1;
            ctx->gz_device_files++;
        }
        free_batch(ctx, b);   // (the event behind the results has fired: nothing in flight reads these)
    };

    // two batches in flight: the next one is read and shipped while the kernels of the one before it run
    std::unique_ptr<Batch> flying;
    int side = 0;
    while (!todo.empty()) {
        const std::pair<size_t, size_t> run = todo.back();
        todo.pop_back();
        std::unique_ptr<Batch> b(new Batch(lay_out(run.first, run.second)));
        b->side = side;
        if (pretend_no_room && run.second - run.first >= 2 && pretended.insert(run).second) b->no_room = true;   // (once per run)
        else stage(*b);
        if (b->no_room) {   // (ghip_pool_alloc has left its message in the context: not an error here)
            if (flying) { finish(*flying); flying.reset(); todo.push_back(run); continue; }   // once more with the other batch's memory back
            if (run.second - run.first >= 2 * split_min) {
                const size_t mid = run.first + (run.second - run.first) / 2;
                todo.push_back({mid, run.second});
                todo.push_back({run.first, mid});
                continue;
            }
        }
        if (flying) finish(*flying);
        flying = std::move(b);
        side ^= 1;
        n_batches++;
    }
    if (flying) finish(*flying);
    if (dbg)
        fprintf(stderr, "[ingest gz-device] text %llu bytes in %llu deflate blocks: %llu tokens (%llu matches) in %llu batches, %llu copy rounds\n", (unsigned long long)diag[0],
                (unsigned long long)diag[5], (unsigned long long)diag[1], (unsigned long long)diag[2], (unsigned long long)diag[3], (unsigned long long)diag[4]);
    if (dbg)
        fprintf(stderr, "[ingest gz-device] %zu files in %zu batches, %zu readers: ok %llu, format %llu, data %llu, unusual %llu, overflow %llu, multi %llu, crc %llu, fasta %llu, "
                        "not run %llu; thread-seconds read %.3f slot wait %.3f; device %.3f s; wall %.3f s\n",
                take.size(), n_batches, readers, (unsigned long long)verdicts[0], (unsigned long long)verdicts[1], (unsigned long long)verdicts[2],
                (unsigned long long)verdicts[3], (unsigned long long)verdicts[4], (unsigned long long)verdicts[5], (unsigned long long)verdicts[6],
                (unsigned long long)verdicts[7], (unsigned long long)verdicts[8], t_read, t_wait, ctx->gz_device_us.load() * 1e-6,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count());
    if (hip_failed.load()) return ghip_set_error(ctx, GHIP_EHIP, "the device-side gzip path failed (HIP error)");
    return GHIP_OK;
}
