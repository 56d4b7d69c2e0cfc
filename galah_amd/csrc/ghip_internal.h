// Internal declarations shared by the host API and the HIP kernel launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/galah_hip.h"

struct ghip_kstat {
    uint64_t launches = 0;
    double total_ms = 0.0;
};

struct ghip_pending_event {
    std::string name;
    hipEvent_t start, stop;
};

struct ghip_pool_block {
    void *p;
    size_t bytes;
    bool used;
};

struct ghip_cmin_cache {  // device filter table of ghip_precluster, keyed by (min_ani bits, s, k)
    uint32_t ani_bits = 0, s = 0, k = 0;
    uint16_t *d_cmin = nullptr;
    uint32_t floor = 0;  // smallest common that passes the filter for any total >= 1 (0: min_ani <= 0)
    bool valid = false;
};

// Persistent I/O workers of a context (file ingest).  Threads and their read buffers outlive the calls: spawning 64
// threads costs ~2 ms per call and, worse, every fresh thread page-faults a fresh multi-megabyte buffer under the
// process-wide mmap lock -- with 8 pipelined batches per run that was 2-4 thread-seconds of "read" per batch.
struct ghip_io_pool {
    std::vector<std::thread> threads;
    std::vector<std::vector<uint8_t>> raw;   // one read / inflate buffer per worker, kept mapped
    std::vector<std::vector<uint8_t>> ascii; // one parse buffer per worker (packed ingest: parse here, pack into the pinned slot)
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> job;
    uint64_t generation = 0;
    int active = 0, pending = 0;
    bool stop = false;
    void run(int n, std::function<void(int)> fn);   // fn(worker) on workers 0..n-1; returns when all are done
    void shutdown();
};

// process-wide default options: the environment, read once (api.cpp); ghip_set_options(NULL, ..) replaces them
ghip_options ghip_process_options();
inline bool ghip_dbg(const ghip_options &o, uint32_t bit) { return (o.debug & bit) != 0; }

struct ghip_ctx {
    int device = 0;
    ghip_options opt{};   // copied from the process-wide defaults at ghip_init; ghip_set_options(ctx, ..)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // work that overlaps the main stream's next stage (the deferred ani_bin); made on first use
    std::string err;
    std::mutex err_mu;  // `err` alone: the file ingest runs on a producer thread next to the consumer's kernels, both may fail
    std::mutex mu;  // serialises API calls on this ctx (ghip_ani_pairs is called from rayon workers)
    // hipFuncAttributeMaxDynamicSharedMemorySize is per DEVICE: what this context's device has been told so far, per kernel
    // (a process-wide flag left devices 1..7 of a one-process multi-GPU run without the allowance)
    std::unordered_map<const void *, size_t> dyn_lds;
    std::mutex dyn_lds_mu;
    bool profile = false;
    std::unordered_map<std::string, ghip_kstat> stats;
    std::vector<ghip_pending_event> pending;
    uint64_t last_pairs = 0;
    int num_cus = 256;
    // device-memory pool: hipMalloc/hipFree synchronise the device, so blocks are recycled
    std::vector<ghip_pool_block> pool;
    std::mutex pool_mu;  // the pool itself: taken inside ghip_pool_alloc/free, so that the file ingest can allocate while
                         // another call holds `mu` for the length of its kernels (lock order: mu, then pool_mu)
    ghip_cmin_cache cmin;
    std::vector<double> ani_memo;   // finch_ani(common, total) of the pair stage's host recheck, [total][common] for (k, s) below; NaN = not yet (under mu)
    uint32_t ani_memo_k = 0, ani_memo_s = 0;
    uint64_t *d_kmer_luts = nullptr;  // MurmurHash3 first-stage tables of sketch_kmers21 (12 KiB, built once; pool-owned)
    // ingest staging (ghip_genomes_from_files): pinned double buffers of the worker threads and two copy streams, kept
    // for the life of the context (hipHostMalloc costs ~0.16 ms/MB, hipStreamCreate ~3 ms)
    struct pinned_slot { uint8_t *p = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; int state = 0; uint64_t seq = 0;
                         uint8_t *d = nullptr; size_t dbytes = 0; };   // d: device staging of the packed form (same life as p)   // state: 0 free, 1 owned by a thread, 2 copy in flight (seq = issue order)
    std::vector<pinned_slot> ingest_slots;
    uint8_t *ingest_stage = nullptr;   // one device allocation behind all the slots' staging areas (pinned_slot::d point into it)
    size_t ingest_stage_bytes = 0;
    // pinned bounce buffer of the larger host<->device copies (api_internal.h h2d / d2h), kept and grown on demand
    void *pin_buf = nullptr;
    size_t pin_bytes = 0;
    std::mutex pin_mu;
    ghip_io_pool io;
    std::mutex ingest_mu;  // one file ingest at a time per context (it runs with `mu` released, next to kernels of other calls)
    hipStream_t copy_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    int n_copy_streams = 0;   // streams in use (GHIP_COPY_STREAMS, default 2)
    // device-side gzip path (ingest_gz.cpp): pinned staging of the compressed images (two slots per reader thread) and of
    // the results of the two batches in flight, kept for the life of the context; counters for ghip_ingest_counters
    struct gz_slot { uint8_t *p = nullptr; hipEvent_t ev = nullptr; bool inflight = false; };
    std::vector<gz_slot> gz_slots;
    uint8_t *gz_results[2] = {nullptr, nullptr};
    size_t gz_results_bytes[2] = {0, 0};
    uint8_t *gz_rec_host = nullptr;   // pinned landing place of a batch's record table (grown on demand)
    size_t gz_rec_host_bytes = 0;
    hipEvent_t gz_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // per batch in flight: kernels begin, end, results landed
    hipStream_t gz_stream = nullptr;   // the gz kernels' own stream: the readers' copies of batch b + 1 (copy_stream[]) do not queue behind batch b's kernels
    std::atomic<uint64_t> gz_device_files{0}, gz_host_files{0}, gz_device_us{0};
    std::atomic<uint64_t> ingest_repeats{0};   // calls that outgrew a capacity hint and were repeated in the two-phase form
    std::atomic<int> live_handles{0};   // genomes / sketches / ani indexes still alive
    bool destroyed = false; // ghip_destroy called; the struct is deleted with the last handle
};

// synchronous copies between pageable host memory and the device on the context's stream (pinned bounce buffer from 256 KiB: api.cpp)
int ghip_copy_to_device(ghip_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int ghip_copy_to_host(ghip_ctx *ctx, void *dst, const void *d_src, size_t bytes);
void *ghip_pool_alloc(ghip_ctx *ctx, size_t bytes);  // nullptr on failure (ctx->err set); thread-safe
void ghip_pool_free(ghip_ctx *ctx, void *p);         // thread-safe; the caller has made sure nothing in flight uses p

struct ghip_sketch_work {  // one block of the k-mer pass
    uint32_t slot;   // index into the pending-genome arrays
    uint32_t chunk;  // chunk index within the genome
};

struct ghip_genome_stats {  // reference src/genome_stats.rs:5-9
    uint64_t num_contigs = 0, num_ambiguous_bases = 0, n50 = 0;
};

struct ghip_genomes {
    ghip_ctx *ctx = nullptr;
    size_t n = 0;
    // resident bases: 2-bit codes (base i in bits 2 (i % 16).. of word i / 16; A0 C1 G2 T3) and a validity bitmap (bit i % 32
    // of word i / 32: the stream byte is A/C/G/T); every genome starts at a base offset that is a multiple of 64 and is
    // followed by >= GHIP_TAIL_PAD invalid positions
    uint32_t *d_packed = nullptr;      // [total_alloc / 16]
    uint32_t *d_valid = nullptr;       // [total_alloc / 32]
    uint64_t *d_starts = nullptr;      // [n] base offset of genome g
    uint64_t *d_lens = nullptr;        // [n] stream length L_g
    std::vector<uint64_t> starts, lens;
    std::vector<ghip_genome_stats> stats;  // filled by ghip_genomes_from_files
    uint64_t total_alloc = 0;
    uint64_t total_bases = 0;
    // cached full-pass work list (one entry per 16384-position chunk of every genome)
    ghip_sketch_work *d_work = nullptr;
    size_t n_work = 0;
    uint32_t *d_identity = nullptr;  // [n] 0..n-1
};

struct ghip_sketches {
    ghip_ctx *ctx = nullptr;
    size_t n = 0;
    uint32_t s = 0, k = 0;
    uint64_t *d_hashes = nullptr;  // [n][s]
    uint32_t *d_lens = nullptr;    // [n]
    bool owned = true;
    // probe-form pair stage (pairs_probe.hip): cuckoo sets + work rows, built on first use
    bool probe_ready = false;
    uint32_t probe_flags = 0;          // != 0: fall back to the merge-path kernel
    uint64_t *d_tables = nullptr;      // [n][2*next_pow2(s)]
    uint32_t *d_tags = nullptr;        // [n][2*next_pow2(s)]: 31-bit tags of the same slots (what the probe kernel keeps in LDS)
    uint64_t *d_arranged = nullptr;    // [n][ghip_probe_arranged_slots(s)]: the rows dealt to lanes by bucket residue (arranged form; else null)
    uint32_t probe_cbits = 0;          // bucket bits the tables' second cuckoo choice shares with the first (arranged form; 0: free)
    uint64_t *d_row_start = nullptr;   // [nta+1]
    std::vector<uint64_t> row_start;
    uint64_t n_work = 0;
    uint32_t probe_cb = 64;            // B-sketches per work item
};

struct ghip_ani_index {
    ghip_ctx *ctx = nullptr;
    size_t n = 0;
    uint32_t k = 0, c = 0, chunk = 0;
    // per-genome seed lists (unordered) and open-addressing membership tables
    uint32_t *d_seed_code = nullptr;   // concatenated canonical 2-bit k-mer codes (k <= 16)
    uint32_t *d_seed_loc = nullptr;    // concatenated: chunk << 16 | strand << 15 | offset in chunk
    uint64_t *d_seed_start = nullptr;  // [n+1] offsets into seed arrays (capacity layout)
    uint32_t *d_seed_count = nullptr;  // [n]
    uint32_t *d_seed_thr = nullptr;    // [n] selection threshold (2^32 - 1) / c_g of every genome
    std::vector<uint32_t> seed_thr;    // the same on the host; c_g = ghip_ani_density(glen[g], c)
    uint32_t *d_seg_count = nullptr;   // [n][GHIP_ANI_SEGMENTS] while the index is being built (seed_common.h), then freed
    uint32_t *d_bin_start = nullptr;   // [n][GHIP_ANI_BIN_COUNT+1] CSR offsets of the binned seed list
    uint32_t *d_chunk_total = nullptr; // concatenated per-chunk seed totals T_c
    uint64_t *d_chunk_start = nullptr; // [n+1]
    uint64_t *d_glen = nullptr;        // [n] stream lengths
    std::vector<uint64_t> glen, chunk_start, seed_start;
    std::vector<uint32_t> seed_count;
    uint32_t max_chunks = 0;
    bool owned = true;  // false: the four flat arrays are borrowed (ghip_ani_index_wrap_device)
    // Deferred finish (ghip_sketch_and_index): ani_bin runs on the context's side stream, next to whatever the caller does
    // next on the main stream (the pair stage reads only the sketch matrix); every use of the device arrays waits for it
    // first (ghip_index_wait).  bin_done = the event behind ani_bin, bin_scratch = its inputs, freed once it is done.
    mutable hipEvent_t bin_done = nullptr;
    mutable std::vector<void *> bin_scratch;
};
// comm.cpp, for the lazy ANI rounds of cluster.cpp: variable-length host all-gather whose sizes every rank already knows
// (one collective; out = the blocks in rank order); whether options.fault_stage names `stage` on this rank
int ghip_comm_gatherv_known(ghip_comm *c, const void *send, size_t bytes, std::vector<uint64_t> &sizes, std::vector<uint8_t> &out);
bool ghip_comm_fault(const ghip_comm *c, uint32_t stage);
int ghip_comm_note_error(ghip_comm *c, int rc);   // rc != 0: the context's last error text becomes the communicator's too; returns rc
int ghip_index_wait(ghip_ctx *ctx, const ghip_ani_index *idx);   // ctx->mu held; returns an error if the deferred kernel failed

#define GHIP_ANI_BIN_BITS 14
#define GHIP_ANI_BIN_COUNT (1u << GHIP_ANI_BIN_BITS)
#define GHIP_ANI_SEGMENTS 8u   // the seeding pass files a genome's seeds under the top 3 bits of their bin (seed_common.h)
#define GHIP_ANI_LDS_PAIR_CHUNKS 2900u  // chunks of BOTH genomes of a pair whose band votes + repeat counter (36 B) + aligned-chunk list (8 B) fit the
                                       // 160 KiB LDS next to the 28 KiB seed stage; a pair above it takes the kernel form with its votes in global memory
#define GHIP_ANI_MAX_CHUNKS 65535u     // per genome: a seed's chunk is a 16-bit field (1.3 Gb at the default 20 kb chunk)
#define GHIP_ANI_SEEDS_WANTED 8192u    // ghip_ani_density: a genome shorter than this many seeds' worth of bases is seeded four times denser
#define GHIP_ANI_POOL_BELOW 9u          // a pair with fewer aligned chunks than this is estimated from their pooled counts, not their median
#define GHIP_ANI_MAX_CHUNK_LEN 32768u  // a seed's offset in its chunk is a 15-bit field

int ghip_set_error(ghip_ctx *ctx, int code, const std::string &msg);
// raises the dynamic-LDS allowance of `kernel` on the context's device to at least `bytes` (no-op once it is that large)
void ghip_ensure_dyn_lds(ghip_ctx *ctx, const void *kernel, size_t bytes);

#define GHIP_HIP_CHECK(ctx, expr)                                                          \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ghip_set_error((ctx), GHIP_EHIP,                                        \
                                  std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

// RAII-less helpers for per-kernel timing; no-ops unless ctx->profile.
void ghip_prof_begin(ghip_ctx *ctx, const char *name);
void ghip_prof_end(ghip_ctx *ctx);

// ---- launchers (defined in the .hip files) ----
// gz_inflate.hip: the device path of gzip input, one batch (struct ghip_gz_job: gz_common.h)
struct ghip_gz_job;
size_t ghip_gz_chunks_of(uint64_t text_cap);   // chunk summaries the FASTA pass needs for a text of that many bytes
size_t ghip_gz_chunk_bytes();
void ghip_launch_gz_batch(hipStream_t stream, const uint8_t *d_in, uint8_t *d_text, ghip_gz_job *d_jobs, uint32_t n_jobs, uint64_t max_text_cap,
                          const uint32_t *d_chunk_start, void *d_chunks, uint32_t *d_rec_next, uint32_t *d_rec_pool, uint32_t rec_room, uint32_t *d_packed,
                          uint32_t *d_valid);
// ingest_gz.cpp: the .gz files among paths[cand[..]] inflated, parsed and packed on the device into g (whose layout and
// resident arrays exist); done[i] = 1 for every file it ingested -- the others are the host path's.  Returns GHIP_OK unless
// the device failed (not: a file it declined).
int ghip_ingest_gz_device(ghip_ctx *ctx, ghip_genomes *g, const char *const *paths, const std::vector<uint64_t> &cap, const std::vector<size_t> &cand,
                          int io_threads, std::vector<uint8_t> &done);
void ghip_launch_synth(ghip_ctx *ctx, uint32_t *d_packed, uint32_t *d_valid, const uint64_t *d_starts, uint64_t length,
                       uint32_t first, uint32_t count, uint32_t members, uint64_t seed, uint32_t sub_thr);

struct ghip_seed_args {  // fused seeding: where sketch_kmers<K, true> puts the ANI seeds
    uint32_t k, chunk;
    const uint32_t *d_seed_thr;   // [n] per-genome selection threshold
    uint32_t *d_seed_code;
    uint32_t *d_seed_loc;
    const uint64_t *d_seed_start;
    uint32_t *d_seg_count;   // [n][GHIP_ANI_SEGMENTS]
    uint32_t *d_chunk_total;
    const uint64_t *d_chunk_start;
};

// resident base format <-> stream bytes, and the validity bits of a genome that arrived as 2-bit codes + runs (sketch.hip)
void ghip_launch_pack_bases(hipStream_t stream, const uint8_t *d_src, uint64_t n, uint64_t gbase_plus_pos0, uint32_t *d_packed, uint32_t *d_valid);
void ghip_launch_unpack_bases(hipStream_t stream, const uint32_t *d_packed, const uint32_t *d_valid, uint64_t gbase, uint8_t *d_out, uint64_t len);
void ghip_launch_valid_from_runs(hipStream_t stream, const uint32_t *d_gtab /* [2 * n_members]: bitmap word offset, length */, uint32_t n_members,
                                 uint64_t max_len, const uint32_t *d_runs /* [3 * n_runs] */, uint32_t n_runs, uint32_t *d_valid_group, bool single);
// packs stream[0, len) into dst: ceil(len / 4) bytes rounded up to 16, then the run table (3 x u32 each) -- false if
// dst_bytes is too small or len >= 2^32.  *used = bytes written, *runs_off = offset of the table, *n_runs its entries
bool ghip_pack_stream(const uint8_t *stream, size_t len, uint8_t *dst, size_t dst_bytes, size_t *used, size_t *runs_off, uint32_t *n_runs);
// ghip_parse_fasta and ghip_pack_stream in one pass over the file image (the table behind room for cap_hint bases); *fit =
// false: the stream does not fit that layout, nothing usable was written
int ghip_parse_fasta_packed(const uint8_t *buf, size_t n, const char *path, uint8_t *dst, size_t dst_bytes, size_t cap_hint,
                            size_t *out_len, ghip_genome_stats &st, std::string &err, size_t *used, size_t *runs_off,
                            uint32_t *n_runs, bool *fit);
int ghip_launch_hash_floor(ghip_ctx *ctx, uint64_t wave_positions, double *ms);
void ghip_launch_sketch_kmers(ghip_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_valid, const uint64_t *d_starts,
                              const uint64_t *d_lens, const uint32_t *d_slot_genome,
                              const uint64_t *d_slot_thr, const uint64_t *d_slot_cand_start,
                              const uint32_t *d_slot_cand_cap, const ghip_sketch_work *d_work,
                              size_t n_work, uint32_t k, uint32_t seed, uint64_t *d_cand,
                              uint32_t *d_cand_count, const ghip_seed_args *seeds /* nullable */);
void ghip_launch_sketch_select(ghip_ctx *ctx, const uint32_t *d_slot_genome, size_t n_slots,
                               uint64_t *d_cand, const uint32_t *d_cand_count,
                               const uint64_t *d_slot_cand_start, const uint32_t *d_slot_cand_cap, uint32_t max_cap,
                               uint32_t s, uint64_t *d_hashes, uint32_t *d_lens, uint32_t *d_status);
void ghip_pair_geometry(uint32_t s, uint32_t *s_pad, uint32_t *sp, uint32_t *pt);

void ghip_launch_pairs(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n,
                       uint32_t s, const uint16_t *d_cmin, uint32_t rank, uint32_t world, uint32_t row_lo,
                       ghip_pair *d_out, unsigned long long *d_count, uint64_t cap,
                       uint64_t *pairs_compared);

void ghip_launch_pairs_global(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s,
                              const uint16_t *d_cmin, uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *d_out,
                              unsigned long long *d_count, uint64_t cap, uint64_t *pairs_compared);
constexpr uint32_t GHIP_MAX_SKETCH_SIZE = 65535;   // common / total travel as u16 in the device filter table (the reference: no bound, src/finch.rs:55-61; its CLI hard-wires 1000)
constexpr size_t GHIP_JOIN_MIN_N = 1200;  // the inverted-index form of the pair stage takes over from here (s = 1000: a tie with the
                                          // dense probe kernel at 1 000 genomes, 0.93 vs 1.40 ms at 1 500)
uint32_t ghip_cmin_floor(const std::vector<uint16_t> &cmin);
int ghip_pairs_join(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s,
                    const uint16_t *d_cmin, uint32_t cmin_floor, uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *d_out,
                    unsigned long long *d_count, uint64_t cap, uint64_t *pairs_compared, bool *used, bool *late_decline,
                    std::vector<uint32_t> *empties = nullptr /* out: the empty sketches, whose pairs the caller adds (null: decline on any) */,
                    uint8_t *d_big = nullptr /* n zeroed bytes: genomes of element buckets too large for the join are marked, their mutual pairs left to
                                                the caller's dense pass over the marked rows (null: such a bucket makes the join decline) */,
                    bool *has_big = nullptr);
void ghip_launch_gather_rows(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, const uint32_t *d_idx, size_t m, uint32_t s,
                             uint64_t *d_out, uint32_t *d_out_lens);

// hash-sharded multi-rank form of the join (pairs_join.hip; driven by comm.cpp)
constexpr size_t GHIP_JOIN_ENTRY_BYTES = 16;   // {u64 pair key (i << 32 | j), u32 partial common, u32 pad}
int ghip_pairs_join_partials(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s, uint32_t hrank,
                             uint32_t hworld, void **d_entries_out /* pool block, caller frees */, uint32_t *n_entries_out, uint32_t *status,
                             unsigned long long *rec_total);
int ghip_pairs_join_finish(ghip_ctx *ctx, const void *d_all, uint32_t n_all, uint32_t n_mine_bound, const uint64_t *d_hashes, const uint32_t *d_lens,
                           uint32_t s, const uint16_t *d_cmin, uint32_t cmin_floor, uint32_t rank, uint32_t world, ghip_pair *d_out,
                           unsigned long long *d_count, uint64_t cap, bool *ok);
// pieces of the pair stage shared with it (api_pairs.cpp; ctx->mu held by the caller of the first, not needed for the second)
int ghip_pair_filter_prepare(ghip_ctx *ctx, uint32_t s, uint32_t k, float min_ani);
int ghip_pairs_finalize(ghip_ctx *ctx, std::vector<ghip_pair> &host, uint32_t k, float min_ani, size_t n, bool filter_share,
                        uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n);
int ghip_precluster_dense_share(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n);
size_t ghip_probe_table_slots(uint32_t s);
size_t ghip_probe_arranged_slots(uint32_t s);
void ghip_launch_pair_tables(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s,
                             uint64_t *d_tables, uint32_t *d_tags, uint32_t *d_flags, uint64_t *d_arranged, uint32_t cbits);
uint32_t ghip_probe_constrained_bits(uint32_t option, uint32_t s);
uint64_t ghip_probe_work_rows(size_t n, int num_cus, uint32_t *cb_out, std::vector<uint64_t> &row_start);
uint64_t ghip_probe_pairs_of_rank(size_t n, uint32_t cb, const std::vector<uint64_t> &row_start, uint32_t rank, uint32_t world);
void ghip_launch_pairs_probe(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, const uint64_t *d_tables, const uint32_t *d_tags,
                             size_t n, uint32_t s, uint32_t cb, const uint64_t *d_row_start, uint32_t nta, uint64_t n_work,
                             const uint16_t *d_cmin, uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *d_out,
                             unsigned long long *d_count, uint64_t cap, const uint64_t *d_arranged, uint32_t cbits);

void ghip_launch_ani_seeds(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, const uint32_t *d_seed_thr, uint32_t chunk,
                           uint32_t *d_seed_code, uint32_t *d_seed_loc, const uint64_t *d_seed_start,
                           uint32_t *d_seg_count, uint32_t *d_chunk_total, const uint64_t *d_chunk_start,
                           const ghip_sketch_work *d_work, size_t n_work);
void ghip_launch_copy_runs(hipStream_t stream, const uint32_t *d_src, uint32_t *d_dst, const uint64_t *d_runs /* [n][3]: source word, destination word, words */,
                           size_t n_runs);
void ghip_launch_ani_bin(ghip_ctx *ctx, size_t n, const uint32_t *in_code, const uint32_t *in_loc, uint32_t *out_code,
                         uint32_t *out_loc, const uint64_t *d_seed_start, const uint32_t *d_seg_count,
                         uint32_t *d_bin_start, uint32_t *d_pos_tmp);
// the density a genome of `len` stream bytes is seeded at, given the base density c (oracle: go_ani_density)
uint32_t ghip_ani_density(uint64_t len, uint32_t c);
// pairs = the host copy of d_pairs (the launcher sorts the pairs into the LDS form and the general form by their genomes)
int ghip_launch_ani_pairs(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *pairs, const uint32_t *d_pairs, size_t n_pairs,
                          uint64_t *d_out /* [n_pairs][6] = M, T, aligned bases of q, aligned chunks, -, aligned bases of r */);

constexpr uint32_t GHIP_SKETCH_POS_PER_THREAD = 64;
constexpr uint32_t GHIP_SKETCH_THREADS = 256;
constexpr uint32_t GHIP_SKETCH_CHUNK = GHIP_SKETCH_POS_PER_THREAD * GHIP_SKETCH_THREADS;  // positions per block
constexpr size_t GHIP_MAX_GRID = 1u << 21;  // workgroups per dispatch (x <= 1024 threads stays below 2^32 work-items)
constexpr uint32_t GHIP_TAIL_PAD = 128;  // invalid positions after every genome (vector loads may over-read)
constexpr uint32_t GHIP_BASE_ALIGN = 64; // a genome's first base sits at a multiple of this many bases
