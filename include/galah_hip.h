/*
 * galah_hip.h -- C ABI of the MI355X-native galah precluster/ANI hot path (libgalah_hip.so).
 *
 * This is the drop-in boundary a galah maintainer binds from Rust (see INTEGRATION.md for the
 * `extern "C"` block and the `impl PreclusterDistanceFinder` / `impl ClusterDistanceFinder`
 * shims).  Plain pointers and sizes only; no unwinding, no callbacks except the optional ANI
 * callback of ghip_cluster; every function returns 0 on success or a GHIP_E* code, and
 * ghip_last_error() gives the text the Rust shim turns into the reference's panic!().
 *
 * Reference interfaces replaced (paths into wwood/galah @ v0.5.1):
 *   trait PreclusterDistanceFinder::distances      src/lib.rs:29-30, impl src/finch.rs:13-24
 *   finch::sketch_files call                        src/finch.rs:55-69
 *   finch::distance::distance pair loop             src/finch.rs:74-96
 *   trait ClusterDistanceFinder::calculate_ani      src/lib.rs:47-55, impl src/skani.rs:695-716
 *   SortedPairGenomeDistanceCache                   src/sorted_pair_genome_distance_cache.rs:5-59
 *   clusterer::cluster                              src/clusterer.rs:14-152
 *
 * Index space of every result = positions in the input path/genome list (src/finch.rs:75-92).
 */
#ifndef GALAH_HIP_H
#define GALAH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GHIP_OK 0
#define GHIP_EINVAL 1   /* bad argument */
#define GHIP_EIO 2      /* file could not be read / is not FASTA */
#define GHIP_EHIP 3     /* HIP runtime error (no device, OOM, launch failure) */
#define GHIP_ENOMEM 4   /* host allocation failure */
#define GHIP_EUNSUPPORTED 5 /* mode the finch back-end refuses (src/finch.rs:14-15,26-41) */
#define GHIP_ECALLBACK 6    /* the ANI callback of ghip_cluster returned < 0 (its calculate_ani failed) */
#define GHIP_EPEER 7        /* multi-rank call: ANOTHER rank failed; every rank returns together (the failing one with its own code) */

#define GHIP_ABI_VERSION 2   /* 2: ghip_options, ghip_comm_agree, ghip_cluster_index_comm, ghip_cluster_ranks */

typedef struct ghip_ctx ghip_ctx;           /* one per (process, GPU) */
typedef struct ghip_genomes ghip_genomes;   /* normalised base streams resident in HBM */
typedef struct ghip_sketches ghip_sketches; /* packed MinHash sketch matrix resident in HBM */
typedef struct ghip_ani_index ghip_ani_index; /* FracMinHash seed index resident in HBM */

/* One surviving precluster pair: what src/finch.rs:91-93 inserts into the cache.
 * i < j; common/total are raw_distance's integers; ani = (1.0 - mash_distance) as f32. */
typedef struct {
    uint32_t i, j;
    uint32_t common, total;
    float ani;
} ghip_pair;

/* ---------------------------------------------------------------- context */
int ghip_abi_version(void);
int ghip_device_count(void);                       /* 0 when no GPU is visible */
int ghip_init(int device, ghip_ctx **out);         /* device ordinal (LOCAL_RANK) */
void ghip_destroy(ghip_ctx *ctx);
const char *ghip_last_error(const ghip_ctx *ctx);  /* ctx may be NULL: last ghip_init error */
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = ctx's own stream. */
int ghip_set_stream(ghip_ctx *ctx, void *hip_stream);
int ghip_synchronize(ghip_ctx *ctx);
/* Async device-to-device copy on the ctx stream (moves index arrays into exchange buffers). */
int ghip_memcpy_d2d(ghip_ctx *ctx, void *d_dst, const void *d_src, size_t nbytes);

/* ---------------------------------------------------------------- options
 * Every switch the library has, as a struct set PER CONTEXT (round 4; until then 28 getenv sites inside the library, which a
 * Rust host cannot set per call and which are not thread-safe against setenv).  The environment variables named below are
 * read ONCE, when the process-wide defaults are first needed, and only seed those defaults; a context copies the defaults
 * at ghip_init; ghip_set_options(ctx, ..) changes one context, ghip_set_options(NULL, ..) the process-wide defaults (what
 * the context-less entry points -- ghip_cluster*, ghip_fasta_stream -- and later ghip_init calls use).  Set struct_size =
 * sizeof(ghip_options); a shorter struct from an older host sets the fields it has.  Call with no other call of the
 * context in flight. */
enum { GHIP_PAIR_AUTO = 0, GHIP_PAIR_JOIN = 1, GHIP_PAIR_PROBE = 2, GHIP_PAIR_MERGE = 3 };       /* GHIP_PAIR_KERNEL=join|probe|merge */
enum { GHIP_JOIN_HASH = 0, GHIP_JOIN_RECORDS = 1, GHIP_JOIN_REPLICATE = 2 };                     /* GHIP_JOIN_RANKS=records|replicate */
enum { GHIP_INGEST_PACKED = 0, GHIP_INGEST_ASCII = 1, GHIP_INGEST_PAGEABLE = 2, GHIP_INGEST_TWO_PHASE = 3 };  /* GHIP_INGEST=ascii|pageable|two-phase */
enum { GHIP_DEBUG_INGEST = 1, GHIP_DEBUG_PRECLUSTER = 2, GHIP_DEBUG_COMM = 4, GHIP_DEBUG_CLUSTER = 8, GHIP_DEBUG_ANI = 16,  /* GHIP_*_DEBUG: host laps on stderr */
       GHIP_DEBUG_POOL_EXACT = 32 };  /* the device-memory pool recycles a block only for a request of its own size (GHIP_POOL_EXACT): every
                                         buffer then ends where its request did -- for memory checkers (an address sanitizer, a guard page) */
/* where ghip_options.fault_stage makes rank fault_rank fail (tests of the multi-rank error paths; 0 = never) */
enum { GHIP_FAULT_NONE = 0, GHIP_FAULT_SKETCH = 1, GHIP_FAULT_PAIRS_STAGE1 = 2, GHIP_FAULT_PAIRS_STAGE2 = 3, GHIP_FAULT_INDEX_PACK = 4,
       GHIP_FAULT_ANI_ROUND = 5,
       GHIP_FAULT_GZ_SMALL_BATCHES = 6 /* not a failure: the device-side gzip path cuts its batches at 3 files (several in flight with a handful of
                                          files) and sizes their record pools for one record per 4 KiB of text; with fault_rank = 1 every run of two or
                                          more files also finds "no room" once and is halved */ };
typedef struct ghip_options {
    uint32_t struct_size;       /* sizeof(ghip_options) of the caller */
    uint32_t pair_form;         /* GHIP_PAIR_*: form of the pair stage (AUTO: by n and s; JOIN still declines what it cannot do) */
    uint32_t join_ranks;        /* GHIP_JOIN_*: how several ranks share the join (HASH: hash-sharded, the default) */
    uint32_t ingest_form;       /* GHIP_INGEST_*: how file bytes reach the device (PACKED: 2-bit codes + runs, the default) */
    uint32_t ingest_groups;     /* 1 (default): small files are ingested in groups; 0: one by one   (GHIP_INGEST_NO_GROUPS) */
    uint32_t io_threads_plain;  /* reader threads for plain files, 0 = from the CPU quota   (GHIP_INGEST_THREADS_PLAIN) */
    uint32_t io_threads_gz;     /* inflating threads, 0 = from the CPU quota                (GHIP_INGEST_THREADS_GZ) */
    uint32_t copy_streams;      /* 1..4 host-to-device copy streams of the ingest, default 2 (GHIP_COPY_STREAMS) */
    uint32_t use_libdeflate;    /* 1 (default): libdeflate when the host has it; 0: zlib     (GHIP_NO_LIBDEFLATE) */
    uint32_t pipeline_pieces;   /* 1 (default): plain inputs above 1 GiB are ingested in pieces next to the kernels (GHIP_PIPELINE=0) */
    uint32_t overlap_binning;   /* 1 (default): ani_bin runs on the side stream next to the pair stage (GHIP_NO_OVERLAP) */
    uint32_t lazy_flush_below;  /* a lazy ANI round of fewer requests asks for everything still open, default 512 (GHIP_LAZY_FLUSH_BELOW) */
    uint32_t cluster_threads;   /* worker threads of the host clusterer, 0 = by size         (GHIP_CLUSTER_THREADS) */
    uint32_t ani_force_general; /* 1: every pair through the general ani_pairs form (measurement aid) (GHIP_ANI_FORCE_GENERAL) */
    uint32_t ani_tall_below;    /* pair lists shorter than this run 16 waves per pair, default 200 (GHIP_ANI_TALL_BELOW) */
    uint32_t debug;             /* GHIP_DEBUG_* bits */
    uint32_t pair_debug;        /* merge-path kernel's debug selector (GHIP_PAIR_DEBUG) */
    uint32_t fault_stage;       /* GHIP_FAULT_*: tests only */
    uint32_t fault_rank;
    uint32_t join_fused;        /* 1: the join's partitions in their fused form -- first level into fixed-capacity buckets (no histogram
                                   pass, no scan, no host round trip), single-launch scans: 12 launches instead of 20; an outgrown
                                   capacity repeats the call in the exact form.  0: the exact form   (GHIP_JOIN_FUSED) */
    uint32_t probe_arranged;    /* != 0: the dense probe kernel's B rows dealt to the lanes by bucket residue and the second cuckoo
                                   choice sharing low bucket bits with the first (1: 3 bits for tables of >= 1 024 buckets, 2 below;
                                   2..4: that many) -- fewer LDS bank conflicts.  Takes effect when a matrix's tables are built.
                                   (GHIP_PROBE_ARRANGED) */
    uint32_t comm_timeout_ms;   /* RCCL transport: how long a rank waits inside one collective for its peers before it aborts the
                                   communicator and returns GHIP_EPEER.  0 (default): for ever -- RCCL's own error report
                                   (ncclCommGetAsyncError) and the stream's are then the only ways out.  The time counts from this
                                   rank's own entry, so it also bounds how late a healthy peer may be: the status-word exchange at
                                   a phase boundary, where uneven shards show, is given 10 times the value, the data collectives
                                   behind it (every rank was just seen) the value itself.  Choose it above the worst imbalance
                                   the job may show between two boundaries / 10, e.g. 60 000   (GHIP_COMM_TIMEOUT_MS) */
    uint32_t gz_device;         /* N > 0: gzip input is inflated, checked (CRC-32, ISIZE), parsed and packed ON THE DEVICE, one wavefront per file
                                   (gz_inflate.hip) -- for every file named *.gz of which the call holds N files' worth (its gzip files together
                                   have at least N times its text: N files of one size, or more of mixed sizes; a launch takes as long as its
                                   largest file).  A file that path does not take (too large a share, further members, FHCRC, a damaged
                                   stream, ...) goes through the host's inflate as before, which alone words the errors.  0 (default until a
                                   GPU run has timed it): host inflate   (GHIP_GZ_DEVICE) */
} ghip_options;
int ghip_get_options(const ghip_ctx *ctx /* NULL: the process-wide defaults */, ghip_options *out);
int ghip_set_options(ghip_ctx *ctx /* NULL: the process-wide defaults */, const ghip_options *opt);

/* What the file ingest of this context has done so far: out[0] = gzip files inflated on the device, out[1] = gzip files
 * offered to the device path that went through the host's inflate instead, out[2] = microseconds of device time of the
 * device path's launches (HIP events), out[3] = calls in which a stream outgrew its capacity hint (a multi-member gzip that is
 * not BGZF): those files were read a second time and the resident genomes moved into a layout of exact lengths. */
int ghip_ingest_counters(ghip_ctx *ctx, uint64_t out[4]);

/* Per-kernel HIP-event timing (recorded on the launch stream).  Kernel names:
 * "sketch_kmers", "sketch_select", "pair_table_build", "pair_intersect_tile", "ani_seeds", "ani_bin",
 * "ani_pairs",
 * "synth_genomes".  enable=1 starts collecting, ghip_kernel_stats drains finished events. */
int ghip_profile_enable(ghip_ctx *ctx, int enable);
int ghip_profile_reset(ghip_ctx *ctx);
int ghip_kernel_stats(ghip_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms);

/* Measures the sketch pass's own roof on this device: the MurmurHash3_x64_128 filter instructions alone (no bases, no
 * tables, no selection), issued for `wave_positions` wave-level evaluations on all SIMDs; *out_ms = kernel duration.
 * bench.py reports sketch_kmers against it (bases per launch / 64 wave-positions). */
int ghip_selftest_hash_floor(ghip_ctx *ctx, uint64_t wave_positions, double *out_ms);

/* ---------------------------------------------------------------- genome ingest
 * Device format of one genome: stream G = for each FASTA record, needletail-normalised bytes
 * followed by one 'N' (k-mers never span records, src/finch.rs:69 -> finch sketch_stream).
 * Any byte other than A,C,G,T breaks k-mers.  (From files the stream crosses PCIe as 2-bit codes plus the runs of its
 * other bytes and is expanded on the device -- the resident form is the one described here; GHIP_INGEST=ascii ships it
 * as it is.) */
int ghip_genomes_from_files(ghip_ctx *ctx, const char *const *paths, size_t n, int io_threads,
                            ghip_genomes **out);
/* Host-only (no GPU, no context): the device-format stream of one FASTA file and its assembly statistics
 * {contigs, ambiguous bases, N50} (src/genome_stats.rs:11-51) -- the parser ghip_genomes_from_files runs per file.
 * *out_stream is malloc'd (free with ghip_free).  Returns GHIP_EIO for unreadable / non-FASTA input. */
int ghip_fasta_stream(const char *path, uint8_t **out_stream, size_t *out_len, uint64_t out_stats[3]);
/* Host streams already in device format: genome g = bytes[offsets[g] .. offsets[g+1]).  Every byte other
 * than A,C,G,T is stored as 'N' (so ghip_genomes_to_host returns the sanitised stream). */
int ghip_genomes_from_host(ghip_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, size_t n,
                           ghip_genomes **out);
/* Counter-based synthetic genomes generated directly in HBM (bench/test input):
 * genome index g = species * members + member. */
int ghip_genomes_synthetic(ghip_ctx *ctx, uint64_t seed, uint32_t n_species, uint32_t members,
                           uint64_t length, double sub_rate, ghip_genomes **out);
/* Same generator, only genomes [first, first+count) of the series (one shard of a multi-GPU job). */
int ghip_genomes_synthetic_range(ghip_ctx *ctx, uint64_t seed, uint32_t members, uint64_t first,
                                 uint64_t count, uint64_t length, double sub_rate, ghip_genomes **out);
size_t ghip_genomes_count(const ghip_genomes *g);
uint64_t ghip_genomes_total_bases(const ghip_genomes *g);
uint64_t ghip_genomes_length(const ghip_genomes *g, size_t idx);
int ghip_genomes_to_host(ghip_ctx *ctx, const ghip_genomes *g, size_t idx, uint8_t *out);
/* Assembly statistics gathered in the same parse (replaces the second file read of
 * calculate_genome_stats, src/genome_stats.rs:11-51); GHIP_EUNSUPPORTED unless built from files. */
int ghip_genomes_stats(const ghip_genomes *g, size_t idx, uint64_t *num_contigs,
                       uint64_t *num_ambiguous_bases, uint64_t *n50);
void ghip_genomes_free(ghip_genomes *g);

/* ---------------------------------------------------------------- MinHash sketching
 * Replaces finch::sketch_files (src/finch.rs:55-69): the s smallest distinct
 * murmurhash3_x64_128(canonical k-mer ASCII, seed).0, ascending; fewer than s allowed
 * (no_strict).  Layout: u64 hashes[n][s] row-major, rows padded with UINT64_MAX; u32 len[n].
 * k in 1..=32, s in 1..=65535 (the reference puts no bound on num_kmers, src/finch.rs:55-61; its CLI hard-wires 1000). */
int ghip_sketch_genomes(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s,
                        uint64_t seed, ghip_sketches **out);
int ghip_sketch_files(ghip_ctx *ctx, const char *const *paths, size_t n, uint32_t k, uint32_t s,
                      uint64_t seed, int io_threads, ghip_sketches **out);
int ghip_sketches_from_host(ghip_ctx *ctx, const uint64_t *hashes, const uint32_t *lens, size_t n,
                            uint32_t s, uint32_t k, ghip_sketches **out);
/* Borrow device memory owned by the caller (e.g. the output of an RCCL all-gather). */
int ghip_sketches_wrap_device(ghip_ctx *ctx, void *d_hashes, void *d_lens, size_t n, uint32_t s,
                              uint32_t k, ghip_sketches **out);
int ghip_sketches_to_host(ghip_ctx *ctx, const ghip_sketches *sk, uint64_t *hashes, uint32_t *lens);
/* Device-to-device copy into caller-owned HBM (e.g. this rank's slice of an all-gather buffer). */
int ghip_sketches_copy_into(ghip_ctx *ctx, const ghip_sketches *sk, void *d_hashes_dst, void *d_lens_dst);
/* Persisted sketch matrix: incremental dereplication without re-sketching (the reference's closest relatives are
 * skani's db directories, src/skani.rs:266-304 and :502-565, and the workflow of docs/preludes/cluster_prelude.md:13-15).
 * File "GHIPSK02": k, s, hash seed, n, row lengths, hashes, genome NAMES (names[i] nullable = empty), FNV-1a checksum; a
 * damaged file is refused with GHIP_EIO.  *out_names (nullable; free with ghip_free) = n NUL-terminated names back to
 * back.  An incremental run = load, sketch the new files, ghip_sketches_concat, ghip_precluster_from(row_lo = saved n):
 * the pairs that touch a new genome; with the saved run's own pairs that is the cache a full run would produce. */
int ghip_sketches_save(ghip_ctx *ctx, const ghip_sketches *sk, const char *path);
int ghip_sketches_save_named(ghip_ctx *ctx, const ghip_sketches *sk, const char *const *names, uint64_t seed, const char *path);
int ghip_sketches_load(ghip_ctx *ctx, const char *path, ghip_sketches **out);
int ghip_sketches_load_named(ghip_ctx *ctx, const char *path, ghip_sketches **out, char **out_names, size_t *out_names_bytes,
                             uint64_t *out_seed);
int ghip_sketches_concat(ghip_ctx *ctx, const ghip_sketches *a, const ghip_sketches *b, ghip_sketches **out);
size_t ghip_sketches_count(const ghip_sketches *sk);
uint32_t ghip_sketches_size(const ghip_sketches *sk);     /* s */
uint32_t ghip_sketches_kmer(const ghip_sketches *sk);     /* k */
void *ghip_sketches_device_hashes(const ghip_sketches *sk); /* u64[n][s] in HBM */
void *ghip_sketches_device_lens(const ghip_sketches *sk);   /* u32[n] in HBM */
void ghip_sketches_free(ghip_sketches *sk);

/* ---------------------------------------------------------------- precluster pairs
 * Replaces the i<j loop of finch::distances (src/finch.rs:74-96).  Output sorted by (i, j).
 * min_ani_fraction is FinchPreclusterer::min_ani (a fraction, src/finch.rs:5-6).
 * The caller frees *out_pairs with ghip_free(). */
int ghip_precluster(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani_fraction,
                    ghip_pair **out_pairs, size_t *out_n);
/* The (new x all) rectangle of an incremental run: only the pairs (i, j), i < j, with j >= row_lo -- rows [0, row_lo) are the
 * genomes of a saved matrix whose mutual pairs the caller already holds.  row_lo = 0 is ghip_precluster. */
int ghip_precluster_from(ghip_ctx *ctx, const ghip_sketches *sk, size_t row_lo, float min_ani_fraction,
                         ghip_pair **out_pairs, size_t *out_n);
/* Multi-GPU share of the pair work: the dense forms deal the upper-triangle tiles block-cyclically (tile t -> rank
 * t % world); the inverted-index form (N >= 1200) keeps the pairs with (i + j) % world == rank -- its element stage runs
 * on every rank, its records are only materialised for the rank's own pairs.  Either way the shares of all ranks
 * partition the triangle, and every rank takes the same form (the choice depends on the sketches alone). */
int ghip_precluster_shard(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani_fraction,
                          uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n);
/* What a multi-rank job calls on every rank with the SAME full sketch matrix: this rank's share as above and
 * *out_replicated = 0 -- or, with GHIP_JOIN_RANKS=replicate in the environment and the inverted-index form running, the
 * WHOLE list on every rank and *out_replicated = 1 (no exchange of candidate lists needed; DESIGN.md section 6 has both
 * timings). */
int ghip_precluster_ranks(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani_fraction,
                          uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n,
                          int *out_replicated);
/* Number of genome pairs the last ghip_precluster[_shard|_ranks] call on this ctx compared. */
uint64_t ghip_last_pairs_compared(const ghip_ctx *ctx);

/* ---------------------------------------------------------------- ANI on candidate pairs
 * Replaces SkaniClusterer::calculate_ani (src/skani.rs:708-716 -> calculate_skani :718-788),
 * batched.  Returns PERCENT, 0.0 when below the aligned-fraction gate (skani prints no row).
 * Build-defined estimator (FracMinHash seeds, matches colinear within a chunk), skani parity unpinned: see DESIGN.md
 * "ANI" and oracle/galah_oracle_ani.c.  Symmetric in the pair.  Thread-safe. */
/* Seed density is PER GENOME: the base density c (125; 30 = skani's --small-genomes, src/skani.rs:152-153) for a genome long
 * enough to hold ~8192 seeds at it, else four times denser, and so on (125 -> 31 -> 7 -> 1): c_g = c; while (c_g > 1 &&
 * len < 8192 c_g) c_g = max(1, c_g / 4).  A pair is evaluated at the sparser of its two densities (FracMinHash samples nest).
 * Limits: k <= 16; chunk <= 32768; at most 65535 chunks per genome (1.3 Gb at the default 20 kb chunk). */
int ghip_ani_index_build(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t c,
                         uint32_t chunk, ghip_ani_index **out);
/* MinHash sketches and ANI index from ONE pass over the bases (fused kernel for k = 21; otherwise the
 * two passes run back to back).  Results are identical to ghip_sketch_genomes + ghip_ani_index_build. */
int ghip_sketch_and_index(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s, uint64_t seed,
                          uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, ghip_sketches **out_sk,
                          ghip_ani_index **out_idx);
/* Files in -> sketches + ANI index (+ statistics) in one call: ONE read of every FASTA file serves both back-ends
 * (the reference reads each genome for finch and again, twice per pair, for skani: src/finch.rs:69,
 * src/skani.rs:725-726), and at most batch_bytes of bases are resident at a time (0 = default 96 GiB), so inputs
 * larger than HBM work.  out_idx NULL: sketches only (then ani_* are ignored).  out_stats NULL or u64[n][3] =
 * contigs, ambiguous bases, N50 (src/genome_stats.rs:11-51). */
int ghip_sketch_and_index_files(ghip_ctx *ctx, const char *const *paths, size_t n, uint32_t k, uint32_t s,
                                uint64_t seed, uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, int io_threads,
                                uint64_t batch_bytes, ghip_sketches **out_sk, ghip_ani_index **out_idx,
                                uint64_t *out_stats);
/* The ANI estimator is build-defined (skani's own numbers cannot be reproduced here: INTEGRATION.md section 6), so its
 * definition carries a version: it changes whenever ghip_ani_pairs would return another value for some input (a run's ANI
 * values are comparable with another run's only at the same version; a persisted table of them should record it). */
#define GHIP_ANI_DEFINITION_VERSION 5u
uint32_t ghip_ani_definition_version(void);
int ghip_ani_pairs(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *pairs /* [n][2] */,
                   size_t n, float min_aligned_fraction, float *out_ani_percent,
                   float *out_af /* nullable, [n][2] */);
/* The integers behind ghip_ani_pairs' values (what the device hands back before the host's pow):
 * out[6 p ..] = M, T of the lower-median chunk, chunks the median was taken over, aligned bases of q, of r, c_pair.
 * tests/golden/ani_golden.json freezes them for the reference's fixture genomes. */
int ghip_ani_pairs_detail(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *pairs /* [n][2] */, size_t n,
                          uint64_t *out /* [n][6] */);
void ghip_ani_index_free(ghip_ani_index *idx);
/* Exchange of an ANI index between GPUs (RCCL all-gather of the flat arrays): genome g owns
 * seed slots [sum seed_cap[<g], +seed_cap[g]), bin slots [g*16385, +16385) (offsets relative to its
 * first seed slot) and chunk slots [sum n_chunks[<g], +ceil(genome_len[g]/chunk)); concatenating the
 * arrays of consecutive shards in genome order gives the index of the union. */
typedef struct {
    size_t n;
    uint64_t n_seed_slots, n_bin_slots, n_chunk_slots;
    void *d_seed_code;   /* u32[n_seed_slots]: canonical 2-bit k-mer codes (k <= 16), binned by hash */
    void *d_seed_loc;    /* u32[n_seed_slots]: chunk << 16 | strand << 15 | offset in chunk */
    void *d_bin_start;   /* u32[n_bin_slots]: CSR bin offsets, 2^14 + 1 per genome */
    void *d_chunk_total; /* u32[n_chunk_slots] */
} ghip_ani_layout;
int ghip_ani_index_layout(const ghip_ani_index *idx, ghip_ani_layout *out);
int ghip_ani_index_meta(const ghip_ani_index *idx, uint64_t *genome_len, uint64_t *seed_cap,
                        uint32_t *seed_count); /* host arrays, each [n] */
int ghip_ani_index_wrap_device(ghip_ctx *ctx, size_t n, uint32_t k, uint32_t c, uint32_t chunk,
                               const uint64_t *genome_len, const uint64_t *seed_cap,
                               const uint32_t *seed_count, void *d_seed_code, void *d_seed_loc,
                               void *d_bin_start, void *d_chunk_total, ghip_ani_index **out);

/* ---------------------------------------------------------------- multi-GPU exchange (RCCL over xGMI)
 * The finch path of the reference is one serial loop in one process (src/finch.rs:74-96); this is its sharded form
 * (SURVEY.md 8e): genomes in contiguous blocks (rank r owns [r*B, min((r+1)*B, N)), B = ceil(N / world)), every rank
 * sketches its block, the packed sketch matrix is all-gathered once, the pair work is dealt over the ranks, a pair's ANI
 * is computed where its first genome lives.  One communicator per (rank, context); three transports, same logic:
 *   ghip_comm_init_rank      one process per GPU: RCCL (ncclAllGather over xGMI).  Rank 0 makes the id with
 *                            ghip_comm_unique_id and ships its 128 bytes to the other ranks by any means.
 *   ghip_comm_init_local     ONE process driving `world` contexts, one thread per context (galah's `cluster` CLI is a
 *                            single process): peer copies over xGMI (hipMemcpyPeer), no RCCL needed.
 *   ghip_comm_init_callback  the host supplies an all-gather of host bytes (MPI, gloo, ...); device payloads are staged
 *                            through host memory.  ctx may be NULL for a communicator used for host payloads only.
 * Collectives must be called by every rank of the group, in the same order. */
typedef struct ghip_comm ghip_comm;
#define GHIP_UNIQUE_ID_BYTES 128
/* all-gather of host bytes: recv[r*bytes .. (r+1)*bytes) = rank r's send; return 0 on success */
typedef int (*ghip_allgather_fn)(void *user, const void *send, size_t bytes, void *recv);
int ghip_comm_unique_id(uint8_t id[GHIP_UNIQUE_ID_BYTES]);
int ghip_comm_init_rank(ghip_ctx *ctx, uint32_t rank, uint32_t world, const uint8_t id[GHIP_UNIQUE_ID_BYTES], ghip_comm **out);
int ghip_comm_init_local(ghip_ctx *const *ctxs, uint32_t world, ghip_comm **out_comms /* [world] */);
int ghip_comm_init_callback(ghip_ctx *ctx, uint32_t rank, uint32_t world, ghip_allgather_fn fn, void *user, ghip_comm **out);
void ghip_comm_destroy(ghip_comm *comm);
uint32_t ghip_comm_rank(const ghip_comm *comm);
uint32_t ghip_comm_world(const ghip_comm *comm);
const char *ghip_comm_transport(const ghip_comm *comm);   /* "rccl", "local-peer-copy", "host-callback", "self" */
const char *ghip_comm_last_error(const ghip_comm *comm);
int ghip_comm_allgather_device(ghip_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank);
int ghip_comm_allgather_host(ghip_comm *comm, const void *send, size_t bytes_per_rank, void *recv);
/* All-to-all-v of device bytes: this rank's bytes [send_off[d], send_off[d+1]) of d_send arrive at rank d's
 * d_recv + its recv_off[this rank]; recv_off[r+1] - recv_off[r] is what this rank expects from rank r.  Both arrays
 * hold world + 1 offsets; sender and receiver must agree on every size (GHIP_EINVAL otherwise).  Every rank calls it.
 * rccl: one ncclSend / ncclRecv per peer in one group; local: peer copies; host-callback: all-gather, keep one's parts. */
int ghip_comm_exchange_device(ghip_comm *comm, const void *d_send, const uint64_t *send_off, void *d_recv, const uint64_t *recv_off);
/* the block of rank `rank`: first genome, number of genomes, block size B */
void ghip_shard_range(size_t n_total, uint32_t rank, uint32_t world, size_t *first, size_t *count, size_t *block);
/* local = the sketches of this rank's block -> the matrix of all n_total genomes on every rank (N*s*8 bytes, once) */
int ghip_allgather_sketches(ghip_comm *comm, const ghip_sketches *local, size_t n_total, ghip_sketches **out_full);
/* This rank's share of the precluster pairs of the gathered matrix (every rank passes the same matrix; collective): the
 * multi-rank form of ghip_precluster.  From 1 200 genomes the inverted-index form runs hash-sharded -- every rank
 * partitions 1/world of the hashes, the per-pair partial counts are exchanged (16 bytes per sharing pair and rank),
 * every rank finishes the pairs with (i + j) % world == rank; otherwise, or when any rank declines, as
 * ghip_precluster_ranks (GHIP_JOIN_RANKS=records forces that, =replicate the whole list on every rank). */
int ghip_precluster_comm(ghip_comm *comm, const ghip_sketches *sk, float min_ani_fraction, ghip_pair **out_pairs, size_t *out_n,
                         int *out_replicated);
/* every rank's share of the candidate list (sorted by (i, j)) -> the whole list in (i, j) order on every rank */
int ghip_allgather_pairs(ghip_comm *comm, const ghip_pair *local, size_t n_local, ghip_pair **out_all, size_t *out_n);
/* The ANI index slices a rank needs but does not own (second genomes of pairs whose first genome it owns; `pairs` is
 * the whole list).  *out_index = `local` itself when nothing had to move, else a new handle (free it) = local genomes
 * followed by the foreign genomes THIS rank's pairs reference; out_local_ids[g] = position of global genome g in it,
 * UINT32_MAX if absent.  An owner sends a genome's slices only to the ranks whose pairs reference it
 * (ghip_comm_exchange_device; the host-callback transport all-gathers the packed slices and a rank keeps its parts). */
int ghip_exchange_ani_index(ghip_comm *comm, const ghip_ani_index *local, size_t n_total, const ghip_pair *pairs,
                            size_t n_pairs, ghip_ani_index **out_index, uint32_t *out_local_ids /* [n_total] */);
typedef struct {   /* wall milliseconds of one ghip_distances_and_ani_ranks call on this rank */
    double sketch_ms, allgather_sketches_ms, pairs_ms, allgather_pairs_ms, exchange_ani_index_ms, ani_pairs_ms, gather_ani_ms;
    uint64_t pairs_compared;   /* genome pairs this rank's pair stage was responsible for */
} ghip_rank_times;
/* One whole pass on this rank's block `local` of the n_total genomes: FinchPreclusterer::distances (src/finch.rs:48-97)
 * plus the batched ClusterDistanceFinder::calculate_ani (src/skani.rs:708-716) of every surviving pair.  Every rank
 * returns the same pairs (sorted by (i, j), global indices) and ANI values (percent); free both with ghip_free().
 * out_sketches (nullable): the gathered sketch matrix (free with ghip_sketches_free). */
int ghip_distances_and_ani_ranks(ghip_comm *comm, const ghip_genomes *local, size_t n_total, uint32_t k, uint32_t s,
                                 uint64_t seed, float min_ani_fraction, uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk,
                                 float min_aligned_fraction, ghip_pair **out_pairs, float **out_ani_percent, size_t *out_n,
                                 ghip_sketches **out_sketches, ghip_rank_times *times /* nullable */);
/* The same front, then clusterer::cluster (src/clusterer.rs:14-152) with the LAZY batched ANI rounds of ghip_cluster_index dealt
 * over the ranks -- what one rank runs, for every world size: only the pairs the greedy rules look at are computed
 * (src/clusterer.rs:194-204, 276-296, 377-405), each by the rank that owns its first genome, one variable-length gather per
 * round.  Every rank runs the (deterministic) host clusterer and returns the same clusters as ghip_cluster does; order
 * (nullable) = galah's quality order, as in ghip_cluster_index.  out_pairs / out_sketches nullable (free with ghip_free /
 * ghip_sketches_free).  A failure on one rank takes every rank out of the call together (that rank with its own code, the
 * others with GHIP_EPEER): a status word is agreed at every phase boundary. */
typedef struct {   /* wall milliseconds of one ghip_cluster_ranks call on this rank */
    double sketch_ms, allgather_sketches_ms, pairs_ms, allgather_pairs_ms, exchange_ani_index_ms, ani_rounds_ms, cluster_host_ms;
    uint64_t pairs_compared;    /* genome pairs this rank's pair stage was responsible for */
    uint64_t ani_pairs_asked;   /* precluster pairs whose ANI the clusterer asked for (all ranks together) */
    uint64_t ani_pairs_here;    /* ... of which this rank computed */
    uint64_t lazy_rounds;
} ghip_cluster_times;
int ghip_cluster_ranks(ghip_comm *comm, const ghip_genomes *local, size_t n_total, uint32_t k, uint32_t s, uint64_t seed,
                       float min_ani_fraction, uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, float min_aligned_fraction,
                       const uint32_t *order, float ani_threshold_percent, uint32_t **out_members, uint64_t **out_offsets,
                       size_t *out_n_clusters, ghip_pair **out_pairs, size_t *out_n_pairs, ghip_sketches **out_sketches,
                       ghip_cluster_times *times);
/* The lazy rounds alone, for a host that strings the phases itself: EVERY rank calls it with the same pair list; idx /
 * local_ids = what ghip_exchange_ani_index returned on this rank (local_ids NULL: positions are genome ids).
 * out_stats[5] = pairs asked, rounds, ns in the rounds, ns in all, pairs this rank computed. */
int ghip_cluster_index_comm(ghip_comm *comm, const ghip_ani_index *idx, const uint32_t *local_ids, size_t n_genomes,
                            const ghip_pair *pairs, size_t n_pairs, const uint32_t *order, float ani_threshold_percent,
                            float min_aligned_fraction, uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters,
                            uint64_t *out_stats);
/* Phase boundary of a host that strings the phases itself: every rank passes GHIP_OK or the code of what failed on it; all
 * return together -- GHIP_OK, the rank's own code, or GHIP_EPEER.  (The entry points above do this between their phases.)
 * The word also carries the settings that decide a call's collectives (ghip_options.pair_form, join_ranks,
 * lazy_flush_below): when the ranks hold different ones, every rank returns GHIP_EINVAL -- here, at the head of
 * ghip_precluster_comm and of the lazy rounds, and at every phase boundary of the whole-pass entry points. */
int ghip_comm_agree(ghip_comm *comm, int status);
ghip_ctx *ghip_comm_context(const ghip_comm *comm);

/* Files in -> clusters out on `world` GPUs driven by ONE process (what galah's `cluster` does, src/clusterer.rs:14-152
 * with the finch preclusterer and the batched ANI clusterer): one thread per context ingests and sketches its block of
 * the file list, the exchanges above run over peer copies, the greedy clusterer runs on the caller's thread.
 * Output as ghip_cluster. */
int ghip_cluster_files_multi(ghip_ctx *const *ctxs, uint32_t world, const char *const *paths, size_t n, uint32_t k, uint32_t s,
                             float min_ani_fraction, float ani_threshold_percent, float min_aligned_fraction, uint32_t ani_c,
                             int io_threads, uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters);

/* ---------------------------------------------------------------- host clusterer
 * clusterer::cluster from the precluster cache onwards (src/clusterer.rs:56-152):
 * partition_sketches, find_precluster_cluster_representatives, ..._memberships.
 * pair_ani (nullable) = clusterer ANI per precluster pair, same order as `pairs`, in the
 * clusterer's unit (percent for skani); NaN = None.  When pair_ani is NULL and skip_clusterer
 * is 0, `ani_cb` is called like ClusterDistanceFinder::calculate_ani(genome_a, genome_b).
 * Output: clusters as members[offsets[c] .. offsets[c+1]), representative first
 * (src/cluster_argument_parsing.rs:730); free both with ghip_free(). */
/* Callback return: 1 = Some(*ani), 0 = None, < 0 = calculate_ani failed (the reference would panic): ghip_cluster
 * stops asking and returns GHIP_ECALLBACK.  With n_pairs == 0 neither pair_ani nor ani_cb is needed. */
typedef int (*ghip_ani_callback)(void *user, uint32_t genome_a, uint32_t genome_b, float *ani);
int ghip_cluster(size_t n_genomes, const ghip_pair *pairs, size_t n_pairs, const float *pair_ani,
                 int skip_clusterer, float ani_threshold, ghip_ani_callback ani_cb, void *user,
                 uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters);

/* The same with the clusterer's ANI asked for lazily, in batches: only the precluster pairs that touch a representative
 * are ever needed (src/clusterer.rs:194-204, 377-405 -- what the reference computes one `skani dist` at a time), and which
 * genomes are representatives unfolds in genome order, so all preclusters advance in lock step and one callback per round
 * answers every request: edge_index[x] = index into `pairs`, out_ani[x] = ANI in the clusterer's unit (NaN = None);
 * return 0, or non-zero to abort (GHIP_ECALLBACK).  Rounds = (most representatives in one precluster) + 1.  Same
 * clusters as ghip_cluster with pair_ani for every pair.  *out_pairs_requested (nullable) = pairs asked in total.
 * A round costs a GPU callee one launch's latency however few pairs it holds, so a round of fewer than 512 requests
 * (environment GHIP_LAZY_FLUSH_BELOW) also asks for everything the still-open preclusters lack: a small input is asked for
 * whole in its first round, the tail of a large one is one round instead of several tiny ones. */
typedef int (*ghip_ani_batch_callback)(void *user, const uint32_t *edge_index, size_t n, float *out_ani);
int ghip_cluster_lazy(size_t n_genomes, const ghip_pair *pairs, size_t n_pairs, float ani_threshold,
                      ghip_ani_batch_callback batch_cb, void *user, uint32_t **out_members, uint64_t **out_offsets,
                      size_t *out_n_clusters, uint64_t *out_pairs_requested);
/* The same rounds answered by the HOST's ClusterDistanceFinder (the batched callback of ghip_cluster_lazy) on every rank: each
 * rank's callback gets the edges that rank answers -- first genome in its block of ceil(n / world) genomes -- as indices into
 * `pairs`.  Needs no device (a communicator of ghip_comm_init_callback(NULL, ..) will do).  out_stats[5] = pairs asked (all ranks),
 * rounds, ns in the rounds, ns in all, pairs this rank answered.  A callback failing on one rank takes every rank out together. */
int ghip_cluster_lazy_comm(ghip_comm *comm, size_t n_genomes, const ghip_pair *pairs, size_t n_pairs, const uint32_t *order,
                           float ani_threshold_percent, ghip_ani_batch_callback batch_cb, void *user, uint32_t **out_members,
                           uint64_t **out_offsets, size_t *out_n_clusters, uint64_t *out_stats);


/* clusterer::cluster with the resident ANI index as the ClusterDistanceFinder (src/clusterer.rs:56-152 driven by
 * SkaniClusterer::calculate_ani, src/skani.rs:718-788), whole in native code: ghip_cluster_lazy's rounds answered by
 * ghip_ani_pairs(idx, ..., min_aligned_fraction) -- no host-language callback per round.  `pairs` = the precluster pairs of
 * the index's genomes (ghip_precluster's list).  `order` (nullable) = galah's quality order (the reference sorts its genomes
 * before clustering, src/cluster_argument_parsing.rs:863-1157): order[x] = the genome that comes x-th; the clusters then hold
 * POSITIONS x (the genome is order[x]), representative first, exactly what the reference returns for the sorted list; with
 * order == NULL positions are genome indices.  out_stats (nullable, u64[4]) = pairs asked, rounds, nanoseconds spent in
 * ghip_ani_pairs, nanoseconds in total.  Same clusters as ghip_cluster given the ANI of every pair. */
int ghip_cluster_index(ghip_ctx *ctx, const ghip_ani_index *idx, size_t n_genomes, const ghip_pair *pairs, size_t n_pairs,
                       const uint32_t *order, float ani_threshold, float min_aligned_fraction,
                       uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters, uint64_t *out_stats);

void ghip_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
