/*
 * galah_oracle.h -- CPU restatement of the wwood/galah finch precluster hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (galah_amd/, the C-ABI
 * library libgalah_hip.so) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * The arithmetic of this path lives in third-party crates that are NOT vendored in
 * /root/reference (Cargo.toml:30-32, no Cargo.lock):
 *   finch 0.6.*        (sketch_files, MashSketcher, distance::raw_distance)
 *   needletail 0.5.*   (FASTA parse, Sequence::normalize, canonical_kmers)
 *   murmurhash3 0.0.5  (murmurhash3_x64_128)
 *   skani >= 0.2.2     (external binary; ANI)  -> see ani section, PARITY UNPINNED
 * Their published algorithms are restated here; parity is anchored on galah's own
 * call sites (src/finch.rs:48-97) and its only golden for this path
 * (src/finch.rs:111-119: set1/1mbp.fna vs set1/500kb.fna -> Some(0.9808188)).
 */
#ifndef GALAH_ORACLE_H
#define GALAH_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashing / normalisation / sketching (finch + needletail + murmurhash3) ---- */
uint64_t go_murmur3_x64_128_h1(const uint8_t *key, size_t len, uint32_t seed);
void go_murmur3_x64_128(const uint8_t *key, size_t len, uint32_t seed, uint64_t out[2]);
size_t go_normalize(const uint8_t *in, size_t n, uint8_t *out);

typedef struct go_sketcher go_sketcher;
go_sketcher *go_sketcher_new(uint32_t s, uint32_t k, uint32_t seed);
void go_sketcher_push_record(go_sketcher *sk, const uint8_t *norm, size_t n);
uint32_t go_sketcher_finish(go_sketcher *sk, uint64_t *out /* capacity s */);
uint64_t go_sketcher_total_kmers(const go_sketcher *sk);
void go_sketcher_free(go_sketcher *sk);

/* One sketch per FASTA file (plain or gzip).  0 = ok. */
int go_sketch_file(const char *path, uint32_t k, uint32_t s, uint32_t seed,
                   uint64_t *out, uint32_t *out_len);
int go_sketch_files(const char *const *paths, size_t n, uint32_t k, uint32_t s,
                    uint32_t seed, uint64_t *out /* n*s */, uint32_t *lens, int threads);
/* Sketch an in-memory normalised byte stream in which any non-ACGT byte breaks k-mers. */
uint32_t go_sketch_bytes(const uint8_t *norm, size_t n, uint32_t k, uint32_t s,
                         uint32_t seed, uint64_t *out);

/* ---- distance (finch::distance::raw_distance, old_mode = false) ---- */
void go_raw_distance(const uint64_t *a, uint32_t na, const uint64_t *b, uint32_t nb,
                     uint64_t *common, uint64_t *total);
void go_raw_distance_closed_form(const uint64_t *a, uint32_t na, const uint64_t *b,
                                 uint32_t nb, uint64_t *common, uint64_t *total);
double go_mash_ani(uint64_t common, uint64_t total, uint32_t k);

typedef struct {
    uint32_t i, j;
    uint32_t common, total;
    float ani; /* the exact f32 src/finch.rs:92 stores */
} go_pair;

/* finch::distances pair loop (src/finch.rs:74-96).  threads==1: the reference's serial
 * loop; threads>1: parallel over i (the "fair" baseline B2), same output order. */
size_t go_distances_rows(const uint64_t *sketches, const uint32_t *lens, size_t n, uint32_t s, uint32_t k, float min_ani, size_t row_lo,
                         size_t row_hi, go_pair *out, size_t cap, uint64_t *compared);   /* rows [row_lo, row_hi) of the same loop, serial */
size_t go_distances(const uint64_t *sk, const uint32_t *lens, size_t n, uint32_t s,
                    uint32_t k, float min_ani, go_pair *out, size_t cap, int threads);

/* ---- SortedPairGenomeDistanceCache + clusterer (src/clusterer.rs) ---- */
typedef struct go_cache go_cache;
go_cache *go_cache_new(void);
void go_cache_free(go_cache *c);
void go_cache_insert(go_cache *c, size_t a, size_t b, int has_value, float v);
/* returns 0 = key absent; 1 = present with None; 2 = present with Some(*v) */
int go_cache_get(const go_cache *c, size_t a, size_t b, float *v);
int go_cache_contains(const go_cache *c, size_t a, size_t b);
size_t go_cache_len(const go_cache *c);
void go_cache_entry(const go_cache *c, size_t idx, size_t *a, size_t *b, int *has, float *v);
go_cache *go_cache_transform_ids(const go_cache *c, const size_t *ids, size_t n);

/* calculate_ani callback: returns has_value (0/1) and writes *ani (the clusterer's units). */
typedef int (*go_ani_fn)(void *ctx, size_t genome_a, size_t genome_b, float *ani);

/* clusterer::cluster (src/clusterer.rs:14-152) given the precluster cache.
 * Writes clusters as a flat list: out_members (n entries, rep first in each cluster) and
 * out_offsets (n_clusters+1).  Precluster order: disjoint sets in first-element order,
 * stable-sorted by size descending; threads=1 semantics.  Returns n_clusters. */
size_t go_cluster(size_t n, const go_cache *precluster_cache, int skip_clusterer,
                  float ani_threshold, go_ani_fn ani, void *ani_ctx,
                  size_t *out_members, size_t *out_offsets);

/* ---- genome assembly statistics (src/genome_stats.rs:11-51) ---- */
int go_genome_stats(const char *path, uint64_t *num_contigs, uint64_t *num_ambiguous_bases, uint64_t *n50);

/* ---- synthetic genomes (bench/test input generator; counter-based) ---- */
uint64_t go_splitmix64(uint64_t x);
void go_synth_genome(uint64_t seed, uint32_t species, uint32_t member, uint64_t length,
                     double sub_rate, uint8_t *out /* length ASCII bases */);

/* ---- ANI on candidate pairs (skani-equivalent; PARITY UNPINNED, see galah_oracle_ani.c) ---- */
/* The estimator is build-defined (no skani float exists to pin it), so the definition is VERSIONED: the number changes with
 * every change of what go_ani_pair returns for some input, and tests/golden/ani_golden.json names the version it freezes
 * (tests/test_oracle_golden.py::test_ani_definition_is_versioned).  5 = round 5's rule (pooled counts below 9 aligned chunks);
 * the device's copy is GHIP_ANI_DEFINITION_VERSION (include/galah_hip.h). */
#define GO_ANI_DEFINITION_VERSION 5
uint32_t go_ani_definition_version(void);
typedef struct go_ani_sketch go_ani_sketch;
go_ani_sketch *go_ani_sketch_bytes(const uint8_t *norm, size_t n, uint32_t k, uint32_t c,
                                   uint32_t chunk);
int go_ani_sketch_file(const char *path, uint32_t k, uint32_t c, uint32_t chunk,
                       go_ani_sketch **out);
void go_ani_sketch_free(go_ani_sketch *s);
size_t go_ani_sketch_nseeds(const go_ani_sketch *s);
const uint64_t *go_ani_sketch_seeds(const go_ani_sketch *s);
const uint32_t *go_ani_sketch_chunks(const go_ani_sketch *s);
const uint32_t *go_ani_sketch_positions(const go_ani_sketch *s);   /* start position of every seed */
const uint8_t *go_ani_sketch_strands(const go_ani_sketch *s);      /* 1: canonical code = reverse complement */
uint64_t go_ani_sketch_length(const go_ani_sketch *s);
/* per-genome seed density: c_g = c; while (c_g > 1 && len < 8192 * c_g) c_g = max(1, c_g / 4) */
uint32_t go_ani_density(uint64_t len, uint32_t c);
uint32_t go_ani_sketch_density(const go_ani_sketch *s);
/* returns ANI in PERCENT (skani's unit, src/skani.rs:203-209), 0.0 when AF < min_af */
float go_ani_pair_pool_below(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, uint32_t pool_below, float *af_q,
                             float *af_r, uint64_t detail[6]);   /* measurement only */
float go_ani_pair(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction,
                  float *af_q, float *af_r);
/* the same, plus the integers behind the value: detail = {M, T of the median chunk, aligned chunks, aligned bases of q,
 * aligned bases of r, c_pair} (what the device hands back before the host's pow; tests/golden/ani_golden.json) */
float go_ani_pair_detail(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, float *af_q, float *af_r,
                         uint64_t detail[6]);

/* EXPERIMENTAL (measurement only, scripts/ani_chain_vs_band.py): ordered colinear chain / span denominators / other
 * aggregates in place of the frozen band vote; flags: 1 chain with span denominators (0: chain, whole-chunk denominators),
 * 2 T-weighted median, 4 pooled, 8 band vote with span denominators.  See galah_oracle_ani.c. */
float go_ani_pair_mode(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, int flags, float *af_q, float *af_r,
                       uint64_t detail[6]);

#ifdef __cplusplus
}
#endif
#endif
