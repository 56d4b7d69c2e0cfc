/*
 * galah_oracle.c -- CPU restatement of galah's finch precluster path and host clusterer.
 *
 * TEST INFRASTRUCTURE ONLY (see galah_oracle.h).  Plain C, no dependency on the product.
 * Every function cites the reference file:line it follows (paths into /root/reference),
 * or the third-party crate whose published algorithm it restates.
 *
 * Pinned by: src/finch.rs:111-119 (0.9808188) -- tests/test_oracle_golden.py.
 */
#define _GNU_SOURCE
#include "galah_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * murmurhash3 crate 0.0.5, murmurhash3_x64_128(bytes, seed) -> (h1, h2); finch uses `.0`
 * (call site: finch::sketch_schemes::hashing::hash_f, reached from src/finch.rs:69).
 * This is Austin Appleby's public-domain MurmurHash3_x64_128.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

void go_murmur3_x64_128(const uint8_t *key, size_t len, uint32_t seed, uint64_t out[2]) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    size_t nblocks = len / 16;
    for (size_t b = 0; b < nblocks; b++) {
        uint64_t k1, k2;
        memcpy(&k1, key + 16 * b, 8);
        memcpy(&k2, key + 16 * b + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = key + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
    case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
    case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
    case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
    case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
    case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
    case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
    case 9:  k2 ^= (uint64_t)tail[8];
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
    case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
    case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
    case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
    case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
    case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
    case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
    case 1:  k1 ^= (uint64_t)tail[0];
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

uint64_t go_murmur3_x64_128_h1(const uint8_t *key, size_t len, uint32_t seed) {
    uint64_t o[2];
    go_murmur3_x64_128(key, len, seed, o);
    return o[0];
}

/* ------------------------------------------------------------------------------------------
 * needletail 0.5 sequence::normalize(seq, iupac=false), as called by finch's sketch_stream
 * (`seqrec.normalize(false)`): ACGT kept; acg -> upper; t,u,U -> T; '-' kept; '.','~' -> '-';
 * space, tab, CR, LF dropped; everything else (incl. N, n, IUPAC codes) -> 'N'.
 * ------------------------------------------------------------------------------------------ */
size_t go_normalize(const uint8_t *in, size_t n, uint8_t *out) {
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = in[i], o;
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': o = c; break;
        case 'a': o = 'A'; break;
        case 'c': o = 'C'; break;
        case 'g': o = 'G'; break;
        case 't': case 'u': case 'U': o = 'T'; break;
        case '-': o = '-'; break;
        case '.': case '~': o = '-'; break;
        case ' ': case '\t': case '\r': case '\n': continue;
        default: o = 'N';
        }
        out[m++] = o;
    }
    return m;
}

/* ------------------------------------------------------------------------------------------
 * finch 0.6 MashSketcher (sketch_schemes/mash.rs): keep the `size` smallest DISTINCT hashes
 * (BinaryHeap + count map; duplicates only bump counts), to_vec() sorted ascending.
 * Restated as threshold buffer + periodic sort/unique/truncate -- the same set, see
 * tests/test_oracle_props.py::test_bottom_s_matches_naive.
 * ------------------------------------------------------------------------------------------ */
struct go_sketcher {
    uint32_t s, k, seed;
    uint64_t *buf;
    size_t nbuf, cap;
    uint64_t thr;   /* current s-th smallest distinct hash once s are known, else UINT64_MAX */
    uint64_t total_kmers;
    uint8_t *rc;    /* scratch */
    size_t rc_cap;
};

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

go_sketcher *go_sketcher_new(uint32_t s, uint32_t k, uint32_t seed) {
    go_sketcher *sk = (go_sketcher *)calloc(1, sizeof(*sk));
    sk->s = s; sk->k = k; sk->seed = seed;
    sk->cap = (size_t)s * 4 + 64;
    sk->buf = (uint64_t *)malloc(sk->cap * sizeof(uint64_t));
    sk->thr = UINT64_MAX;
    return sk;
}

static void sketcher_compact(go_sketcher *sk) {
    qsort(sk->buf, sk->nbuf, sizeof(uint64_t), cmp_u64);
    size_t m = 0;
    for (size_t i = 0; i < sk->nbuf; i++)
        if (m == 0 || sk->buf[i] != sk->buf[m - 1]) sk->buf[m++] = sk->buf[i];
    if (m >= sk->s) { m = sk->s; sk->thr = sk->s ? sk->buf[m - 1] : 0; }
    sk->nbuf = m;
}

static inline void sketcher_push_hash(go_sketcher *sk, uint64_t h) {
    sk->total_kmers++;
    if (h > sk->thr) return; /* MashSketcher::push: new_hash <= old_max || len < size */
    if (sk->nbuf == sk->cap) sketcher_compact(sk);
    if (h > sk->thr) return;
    sk->buf[sk->nbuf++] = h;
}

static inline uint8_t comp_base(uint8_t c) {
    switch (c) { /* needletail complement() on a normalised sequence */
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return c;
    }
}

/* needletail canonical_kmers(k, &rc): windows containing a non-ACGT byte are skipped;
 * canonical = rc window iff rc < fwd (byte-lexicographic), else fwd.  k-mers never span
 * records because each record is pushed separately (finch sketch_stream loop). */
void go_sketcher_push_record(go_sketcher *sk, const uint8_t *norm, size_t n) {
    uint32_t k = sk->k;
    if (n < k || sk->s == 0) return;
    if (sk->rc_cap < n) {
        free(sk->rc);
        sk->rc = (uint8_t *)malloc(n);
        sk->rc_cap = n;
    }
    for (size_t i = 0; i < n; i++) sk->rc[n - 1 - i] = comp_base(norm[i]);
    size_t good = 0; /* consecutive ACGT bases ending at position p */
    for (size_t p = 0; p < n; p++) {
        uint8_t c = norm[p];
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T') good++; else good = 0;
        if (good < k) continue;
        size_t start = p + 1 - k;
        const uint8_t *fwd = norm + start;
        const uint8_t *rcw = sk->rc + (n - start - k);
        const uint8_t *canon = (memcmp(rcw, fwd, k) < 0) ? rcw : fwd;
        sketcher_push_hash(sk, go_murmur3_x64_128_h1(canon, k, sk->seed));
    }
}

uint32_t go_sketcher_finish(go_sketcher *sk, uint64_t *out) {
    sketcher_compact(sk);
    memcpy(out, sk->buf, sk->nbuf * sizeof(uint64_t));
    return (uint32_t)sk->nbuf;
}

uint64_t go_sketcher_total_kmers(const go_sketcher *sk) { return sk->total_kmers; }

void go_sketcher_free(go_sketcher *sk) {
    if (!sk) return;
    free(sk->buf); free(sk->rc); free(sk);
}

/* Reads a whole (possibly gzip-compressed) file; needletail auto-detects gzip. */
static uint8_t *slurp(const char *path, size_t *n_out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return NULL;
    size_t cap = 1 << 22, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (;;) {
        if (n == cap) { cap *= 2; buf = (uint8_t *)realloc(buf, cap); }
        int r = gzread(f, buf + n, (unsigned)((cap - n) > (1u << 30) ? (1u << 30) : (cap - n)));
        if (r < 0) { free(buf); gzclose(f); return NULL; }
        if (r == 0) break;
        n += (size_t)r;
    }
    gzclose(f);
    *n_out = n;
    return buf;
}

/* Calls cb(record_sequence_bytes, len) for each FASTA record (needletail parse_fastx_file,
 * FASTA branch): '>' at line start opens a header line; sequence = all following lines. */
typedef void (*record_cb)(void *ctx, const uint8_t *seq, size_t n);
static int for_each_fasta_record(const uint8_t *buf, size_t n, record_cb cb, void *ctx) {
    size_t p = 0;
    while (p < n && (buf[p] == '\n' || buf[p] == '\r')) p++;
    if (p == n) return 0;           /* empty file: no records */
    if (buf[p] != '>') return -2;   /* not FASTA */
    while (p < n) {
        /* header line */
        while (p < n && buf[p] != '\n') p++;
        if (p < n) p++;
        size_t start = p;
        while (p < n) {
            if (buf[p] == '>' && (p == start || buf[p - 1] == '\n')) break;
            p++;
        }
        cb(ctx, buf + start, p - start);
    }
    return 0;
}

struct sketch_file_ctx { go_sketcher *sk; uint8_t *norm; size_t norm_cap; };
static void sketch_record_cb(void *vctx, const uint8_t *seq, size_t n) {
    struct sketch_file_ctx *c = (struct sketch_file_ctx *)vctx;
    if (c->norm_cap < n) { free(c->norm); c->norm = (uint8_t *)malloc(n + 1); c->norm_cap = n; }
    size_t m = go_normalize(seq, n, c->norm);
    go_sketcher_push_record(c->sk, c->norm, m);
}

/* finch::sketch_files -> sketch_stream: ONE sketch per file, all records feed one sketcher
 * (call site src/finch.rs:55-69: Mash{kmers_to_sketch=final_size=s, no_strict, k, seed 0}). */
int go_sketch_file(const char *path, uint32_t k, uint32_t s, uint32_t seed,
                   uint64_t *out, uint32_t *out_len) {
    size_t n;
    uint8_t *buf = slurp(path, &n);
    if (!buf) return -1;
    struct sketch_file_ctx c = { go_sketcher_new(s, k, seed), NULL, 0 };
    int rc = for_each_fasta_record(buf, n, sketch_record_cb, &c);
    if (rc == 0) *out_len = go_sketcher_finish(c.sk, out);
    go_sketcher_free(c.sk);
    free(c.norm); free(buf);
    return rc;
}

int go_sketch_files(const char *const *paths, size_t n, uint32_t k, uint32_t s,
                    uint32_t seed, uint64_t *out, uint32_t *lens, int threads) {
    int err = 0;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (long i = 0; i < (long)n; i++) {
        uint32_t len = 0;
        int rc = go_sketch_file(paths[i], k, s, seed, out + (size_t)i * s, &len);
        lens[i] = len;
        for (uint32_t t = len; t < s; t++) out[(size_t)i * s + t] = UINT64_MAX;
        if (rc != 0) {
#pragma omp critical
            err = rc;
        }
    }
    return err;
}

uint32_t go_sketch_bytes(const uint8_t *norm, size_t n, uint32_t k, uint32_t s,
                         uint32_t seed, uint64_t *out) {
    /* One "record" whose invalid bytes break k-mers; equivalent to pushing the ACGT-only
     * runs separately, because windows containing a non-ACGT byte are skipped anyway. */
    go_sketcher *sk = go_sketcher_new(s, k, seed);
    go_sketcher_push_record(sk, norm, n);
    uint32_t len = go_sketcher_finish(sk, out);
    go_sketcher_free(sk);
    return len;
}

/* ------------------------------------------------------------------------------------------
 * finch 0.6 distance::raw_distance(query, ref, scale = 0) as reached through
 * distance(s1, s2, old_mode=false) at src/finch.rs:79: merge until EITHER list is exhausted.
 * ------------------------------------------------------------------------------------------ */
void go_raw_distance(const uint64_t *a, uint32_t na, const uint64_t *b, uint32_t nb,
                     uint64_t *common, uint64_t *total) {
    uint32_t i = 0, j = 0;
    uint64_t c = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) i++;
        else if (a[i] > b[j]) j++;
        else { c++; i++; j++; }
    }
    *common = c;
    *total = (uint64_t)i + j - c;
}

/* SURVEY.md 0.5 closed form: m = min(max A, max B); i = #{a<=m}; j = #{b<=m}. */
void go_raw_distance_closed_form(const uint64_t *a, uint32_t na, const uint64_t *b,
                                 uint32_t nb, uint64_t *common, uint64_t *total) {
    if (na == 0 || nb == 0) { *common = 0; *total = 0; return; }
    uint64_t m = a[na - 1] < b[nb - 1] ? a[na - 1] : b[nb - 1];
    uint64_t i = 0, j = 0, c = 0;
    for (uint32_t x = 0; x < na; x++) i += a[x] <= m;
    for (uint32_t y = 0; y < nb; y++) j += b[y] <= m;
    for (uint32_t x = 0; x < na; x++) { /* |A n B| by bsearch */
        uint32_t lo = 0, hi = nb;
        while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (b[mid] < a[x]) lo = mid + 1; else hi = mid; }
        c += (lo < nb && b[lo] == a[x]);
    }
    *common = c;
    *total = i + j - c;
}

/* Rust f64::max / f64::min ignore a NaN operand. */
static inline double rust_fmax(double a, double b) { return isnan(a) ? b : (isnan(b) ? a : (a > b ? a : b)); }
static inline double rust_fmin(double a, double b) { return isnan(a) ? b : (isnan(b) ? a : (a < b ? a : b)); }

/* 1.0 - mash_distance, with finch's  jaccard = common/total,
 * mash = -1.0 * ln(2j / (1 + j)) / k, clamped min(1, max(0, .)).   (src/finch.rs:78-86) */
double go_mash_ani(uint64_t common, uint64_t total, uint32_t k) {
    double jaccard = (double)common / (double)total;
    double mash = -1.0 * log((2.0 * jaccard) / (1.0 + jaccard)) / (double)k;
    mash = rust_fmin(1.0, rust_fmax(0.0, mash));
    return 1.0 - mash;
}

/* The same loop (src/finch.rs:74-96) over the rows row_lo <= i < row_hi of its outer index only, serial: a bounded SAMPLE of the
 * faithful (serial) pair loop at a size where the whole of it would take minutes (bench.py's cpu_baseline at 10 000 genomes).
 * Returns the hits (written up to cap); *compared = the pairs it looked at. */
size_t go_distances_rows(const uint64_t *sk, const uint32_t *lens, size_t n, uint32_t s, uint32_t k, float min_ani, size_t row_lo,
                         size_t row_hi, go_pair *out, size_t cap, uint64_t *compared) {
    double thr = (double)min_ani;
    size_t m = 0;
    uint64_t looked = 0;
    for (size_t i = row_lo; i < row_hi && i < n; i++)
        for (size_t j = i + 1; j < n; j++) {
            uint64_t c, t;
            go_raw_distance(sk + i * s, lens[i], sk + j * s, lens[j], &c, &t);
            double ani = go_mash_ani(c, t, k);
            looked++;
            if (ani >= thr) {
                if (m < cap) {
                    out[m].i = (uint32_t)i; out[m].j = (uint32_t)j;
                    out[m].common = (uint32_t)c; out[m].total = (uint32_t)t;
                    out[m].ani = (float)ani;
                }
                m++;
            }
        }
    if (compared) *compared = looked;
    return m;
}

/* src/finch.rs:74-96: for i<j (row-major), keep if ANI(f64) >= (min_ani as f64), store f32. */
size_t go_distances(const uint64_t *sk, const uint32_t *lens, size_t n, uint32_t s,
                    uint32_t k, float min_ani, go_pair *out, size_t cap, int threads) {
    double thr = (double)min_ani;
    if (threads <= 1) {
        size_t m = 0;
        for (size_t i = 0; i < n; i++)
            for (size_t j = i + 1; j < n; j++) {
                uint64_t c, t;
                go_raw_distance(sk + i * s, lens[i], sk + j * s, lens[j], &c, &t);
                double ani = go_mash_ani(c, t, k);
                if (ani >= thr) {
                    if (m < cap) {
                        out[m].i = (uint32_t)i; out[m].j = (uint32_t)j;
                        out[m].common = (uint32_t)c; out[m].total = (uint32_t)t;
                        out[m].ani = (float)ani;
                    }
                    m++;
                }
            }
        return m;
    }
    /* B2 "fair" baseline: rows in parallel, per-row hit lists concatenated in row order. */
    go_pair **rows = (go_pair **)calloc(n, sizeof(*rows));
    size_t *cnt = (size_t *)calloc(n, sizeof(size_t));
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (long i = 0; i < (long)n; i++) {
        size_t rcap = 16, m = 0;
        go_pair *row = (go_pair *)malloc(rcap * sizeof(go_pair));
        for (size_t j = (size_t)i + 1; j < n; j++) {
            uint64_t c, t;
            go_raw_distance(sk + (size_t)i * s, lens[i], sk + j * s, lens[j], &c, &t);
            double ani = go_mash_ani(c, t, k);
            if (ani >= thr) {
                if (m == rcap) { rcap *= 2; row = (go_pair *)realloc(row, rcap * sizeof(go_pair)); }
                row[m].i = (uint32_t)i; row[m].j = (uint32_t)j;
                row[m].common = (uint32_t)c; row[m].total = (uint32_t)t;
                row[m].ani = (float)ani;
                m++;
            }
        }
        rows[i] = row; cnt[i] = m;
    }
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        for (size_t x = 0; x < cnt[i]; x++, m++)
            if (m < cap) out[m] = rows[i][x];
        free(rows[i]);
    }
    free(rows); free(cnt);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * SortedPairGenomeDistanceCache (src/sorted_pair_genome_distance_cache.rs:5-59):
 * BTreeMap<(usize,usize), Option<f32>> with the key sorted on insert/get.
 * Restated as a sorted array (insert = bsearch + memmove), iteration order == BTreeMap order.
 * ------------------------------------------------------------------------------------------ */
typedef struct { size_t a, b; int has; float v; } cache_entry;
struct go_cache { cache_entry *e; size_t n, cap; };

go_cache *go_cache_new(void) { return (go_cache *)calloc(1, sizeof(go_cache)); }
void go_cache_free(go_cache *c) { if (c) { free(c->e); free(c); } }
size_t go_cache_len(const go_cache *c) { return c->n; }

static size_t cache_lower_bound(const go_cache *c, size_t a, size_t b) {
    size_t lo = 0, hi = c->n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        const cache_entry *e = &c->e[mid];
        if (e->a < a || (e->a == a && e->b < b)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

void go_cache_insert(go_cache *c, size_t a, size_t b, int has_value, float v) {
    if (!(a < b)) { size_t t = a; a = b; b = t; } /* :22-28 (a==b keeps (b,a) order too) */
    size_t p = cache_lower_bound(c, a, b);
    if (p < c->n && c->e[p].a == a && c->e[p].b == b) { c->e[p].has = has_value; c->e[p].v = v; return; }
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 64; c->e = (cache_entry *)realloc(c->e, c->cap * sizeof(cache_entry)); }
    memmove(c->e + p + 1, c->e + p, (c->n - p) * sizeof(cache_entry));
    c->e[p].a = a; c->e[p].b = b; c->e[p].has = has_value; c->e[p].v = v;
    c->n++;
}

int go_cache_get(const go_cache *c, size_t a, size_t b, float *v) {
    if (!(a < b)) { size_t t = a; a = b; b = t; } /* :30-36 */
    size_t p = cache_lower_bound(c, a, b);
    if (p < c->n && c->e[p].a == a && c->e[p].b == b) {
        if (c->e[p].has) { if (v) *v = c->e[p].v; return 2; }
        return 1;
    }
    return 0;
}

int go_cache_contains(const go_cache *c, size_t a, size_t b) { return go_cache_get(c, a, b, NULL) != 0; }

void go_cache_entry(const go_cache *c, size_t idx, size_t *a, size_t *b, int *has, float *v) {
    *a = c->e[idx].a; *b = c->e[idx].b; *has = c->e[idx].has; *v = c->e[idx].v;
}

/* :47-58 */
go_cache *go_cache_transform_ids(const go_cache *c, const size_t *ids, size_t n) {
    go_cache *r = go_cache_new();
    for (size_t i = 0; i < n; i++)
        for (size_t j = i + 1; j < n; j++) {
            float v = 0;
            int st = go_cache_get(c, ids[i], ids[j], &v);
            if (st) go_cache_insert(r, i, j, st == 2, v);
        }
    return r;
}

/* ------------------------------------------------------------------------------------------
 * clusterer.rs
 * ------------------------------------------------------------------------------------------ */

/* disjoint 0.8 DisjointSetVec: only the partition matters; sets() enumerated by first
 * (smallest) element [recollection; canonical-form parity only, SURVEY.md H2]. */
static size_t dsu_find(size_t *parent, size_t x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
}

typedef struct { size_t idx; int has; float v; } ref_cand;
static int cmp_ref_cand(const void *pa, const void *pb) {
    /* clusterer.rs:200  a.1.partial_cmp(b.1): Option<f32> ordering, None < Some, ascending */
    const ref_cand *a = (const ref_cand *)pa, *b = (const ref_cand *)pb;
    if (a->has != b->has) return a->has - b->has;
    if (a->has && a->v != b->v) return a->v < b->v ? -1 : 1;
    return (a->idx > b->idx) - (a->idx < b->idx); /* stable tie-break (sort_unstable: unspecified) */
}

/* find_precluster_cluster_representatives (clusterer.rs:182-259), threads=1 semantics of
 * calculate_clusterer_many_to_one_pairwise_stop_early (:276-296): in order, stop at first hit. */
static void find_reps(size_t m, const go_cache *pre, int skip, float thr, go_ani_fn ani,
                      void *actx, const size_t *orig, uint8_t *is_rep, go_cache **calc_out) {
    go_cache *calc = go_cache_new();
    size_t *reps = (size_t *)malloc(m * sizeof(size_t));
    size_t nreps = 0;
    ref_cand *cand = (ref_cand *)malloc(m * sizeof(ref_cand));
    for (size_t i = 0; i < m; i++) {
        size_t nc = 0;
        for (size_t r = 0; r < nreps; r++) {
            float v = 0;
            int st = go_cache_get(pre, i, reps[r], &v);
            if (st) { cand[nc].idx = reps[r]; cand[nc].has = (st == 2); cand[nc].v = v; nc++; }
        }
        qsort(cand, nc, sizeof(ref_cand), cmp_ref_cand);
        int rep = 1;
        if (skip) {
            /* compute_ani_from_preclusterer (:298-313) */
            for (size_t x = 0; x < nc; x++)
                if (cand[x].has && cand[x].v >= thr) rep = 0;
        } else {
            for (size_t x = 0; x < nc; x++) {
                float a = 0;
                int has = ani(actx, orig[cand[x].idx], orig[i], &a); /* calculate_ani(query=rep, ref=i) */
                if (has) {
                    go_cache_insert(calc, cand[x].idx, i, 1, a);   /* :238-240 */
                    if (a >= thr) { rep = 0; break; }              /* find_any stops on a hit */
                }
            }
        }
        is_rep[i] = (uint8_t)rep;
        if (rep) reps[nreps++] = i;
    }
    free(cand); free(reps);
    if (skip) {
        go_cache_free(calc);
        calc = go_cache_new();
        for (size_t x = 0; x < pre->n; x++)
            go_cache_insert(calc, pre->e[x].a, pre->e[x].b, pre->e[x].has, pre->e[x].v); /* :254-255 */
    }
    *calc_out = calc;
}

/* find_precluster_cluster_memberships (clusterer.rs:350-449). */
static void find_memberships(size_t m, const go_cache *pre, go_cache *calc, const uint8_t *is_rep,
                             go_ani_fn ani, void *actx, const size_t *orig, size_t *assign) {
    for (size_t i = 0; i < m; i++) {
        if (is_rep[i]) { assign[i] = i; continue; }
        for (size_t r = 0; r < m; r++) {
            if (!is_rep[r]) continue;
            if (go_cache_contains(calc, i, r)) continue;      /* :381-386 */
            if (!go_cache_contains(pre, i, r)) continue;      /* :388 */
            float a = 0;
            int has = ani(actx, orig[r], orig[i], &a);        /* :392-399 */
            go_cache_insert(calc, i, r, has, a);              /* :400-405 */
        }
        int have_best = 0; float best = 0; size_t best_rep = (size_t)-1;
        for (size_t r = 0; r < m; r++) {                      /* BTreeSet order = ascending */
            if (!is_rep[r]) continue;
            float a = 0;
            int st = go_cache_get(calc, i, r, &a);
            if (st == 2 && (!have_best || a > best)) { have_best = 1; best = a; best_rep = r; }
        }
        if (!have_best) { fprintf(stderr, "go_cluster: best_rep.unwrap() on None (clusterer.rs:444)\n"); abort(); }
        assign[i] = best_rep;
    }
}

size_t go_cluster(size_t n, const go_cache *pcache, int skip_clusterer, float ani_threshold,
                  go_ani_fn ani, void *ani_ctx, size_t *out_members, size_t *out_offsets) {
    /* partition_sketches (:452-487): O(n^2) contains_key probes, joins in (i asc, j asc) order */
    size_t *parent = (size_t *)malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; i++) parent[i] = i;
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < i; j++)
            if (go_cache_contains(pcache, i, j)) {
                size_t ri = dsu_find(parent, i), rj = dsu_find(parent, j);
                if (ri != rj) parent[ri > rj ? ri : rj] = ri > rj ? rj : ri;
            }
    /* sets(): enumerate by first element; members ascending (:67-76) */
    size_t *set_of = (size_t *)malloc(n * sizeof(size_t));
    size_t *root_set = (size_t *)malloc(n * sizeof(size_t));
    size_t nsets = 0;
    for (size_t i = 0; i < n; i++) root_set[i] = (size_t)-1;
    size_t *set_size = (size_t *)calloc(n + 1, sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        size_t r = dsu_find(parent, i);
        if (root_set[r] == (size_t)-1) root_set[r] = nsets++;
        set_of[i] = root_set[r];
        set_size[set_of[i]]++;
    }
    /* sort preclusters by size descending (:79), stable */
    size_t *order = (size_t *)malloc(nsets * sizeof(size_t));
    {
        size_t maxsz = 0;
        for (size_t s = 0; s < nsets; s++) if (set_size[s] > maxsz) maxsz = set_size[s];
        size_t *bucket = (size_t *)calloc(maxsz + 2, sizeof(size_t));
        for (size_t s = 0; s < nsets; s++) bucket[maxsz - set_size[s] + 1]++;
        for (size_t b = 1; b <= maxsz + 1; b++) bucket[b] += bucket[b - 1];
        for (size_t s = 0; s < nsets; s++) order[bucket[maxsz - set_size[s]]++] = s;
        free(bucket);
    }
    size_t *set_start = (size_t *)malloc((nsets + 1) * sizeof(size_t));
    size_t *set_members = (size_t *)malloc(n * sizeof(size_t));
    set_start[0] = 0;
    for (size_t s = 0; s < nsets; s++) set_start[s + 1] = set_start[s] + set_size[s];
    size_t *fill = (size_t *)calloc(nsets, sizeof(size_t));
    for (size_t i = 0; i < n; i++) { size_t s = set_of[i]; set_members[set_start[s] + fill[s]++] = i; }

    size_t nclusters = 0, nout = 0;
    out_offsets[0] = 0;
    for (size_t oi = 0; oi < nsets; oi++) {
        size_t s = order[oi], m = set_size[s];
        const size_t *orig = set_members + set_start[s];
        go_cache *pre = go_cache_transform_ids(pcache, orig, m);  /* :92-93 */
        uint8_t *is_rep = (uint8_t *)calloc(m, 1);
        size_t *assign = (size_t *)malloc(m * sizeof(size_t));
        go_cache *calc = NULL;
        find_reps(m, pre, skip_clusterer, ani_threshold, ani, ani_ctx, orig, is_rep, &calc);
        find_memberships(m, pre, calc, is_rep, ani, ani_ctx, orig, assign);
        for (size_t r = 0; r < m; r++) {                          /* rep first (:371-373) */
            if (!is_rep[r]) continue;
            out_members[nout++] = orig[r];
            for (size_t i = 0; i < m; i++)
                if (!is_rep[i] && assign[i] == r) out_members[nout++] = orig[i];
            out_offsets[++nclusters] = nout;
        }
        go_cache_free(pre); go_cache_free(calc); free(is_rep); free(assign);
    }
    free(parent); free(set_of); free(root_set); free(set_size); free(order);
    free(set_start); free(set_members); free(fill);
    return nclusters;
}

/* ------------------------------------------------------------------------------------------
 * calculate_genome_stats (src/genome_stats.rs:11-51): records, 'N'/'n' count, N50.
 * ------------------------------------------------------------------------------------------ */
struct stats_ctx { uint64_t contigs, ambiguous; uint64_t *lens; size_t n, cap; };
static void stats_record_cb(void *vctx, const uint8_t *seq, size_t n) {
    struct stats_ctx *c = (struct stats_ctx *)vctx;
    uint64_t bases = 0;
    for (size_t i = 0; i < n; i++) {
        if (seq[i] == '\n' || seq[i] == '\r') continue;
        bases++;
        if (seq[i] == 'N' || seq[i] == 'n') c->ambiguous++;   /* :27-31 */
    }
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 64; c->lens = (uint64_t *)realloc(c->lens, c->cap * sizeof(uint64_t)); }
    c->lens[c->n++] = bases;
    c->contigs++;
}

int go_genome_stats(const char *path, uint64_t *num_contigs, uint64_t *num_ambiguous_bases, uint64_t *n50) {
    size_t n;
    uint8_t *buf = slurp(path, &n);
    if (!buf) return -1;
    struct stats_ctx c = { 0, 0, NULL, 0, 0 };
    int rc = for_each_fasta_record(buf, n, stats_record_cb, &c);
    free(buf);
    if (rc) { free(c.lens); return rc; }
    qsort(c.lens, c.n, sizeof(uint64_t), cmp_u64);            /* :34 */
    uint64_t total = 0, run = 0, v = 0;
    for (size_t i = 0; i < c.n; i++) total += c.lens[i];
    for (size_t i = 0; i < c.n; i++) { run += c.lens[i]; if (run >= total / 2) { v = c.lens[i]; break; } }   /* :35-44 */
    *num_contigs = c.contigs; *num_ambiguous_bases = c.ambiguous; *n50 = v;
    free(c.lens);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic genomes: counter-based, so CPU and GPU produce identical bytes
 * with no storage.  Build-defined (no reference counterpart).
 *   ancestor(species)[p] = 2 bits of splitmix64(key(seed, species, 0) + p/32)
 *   member m: position p substituted iff top-32(u) < rate*2^32, u = splitmix64(key(seed,
 *   species, m+1) + p); replacement = (base + 1 + ((u >> 8) % 3)) & 3.
 * ------------------------------------------------------------------------------------------ */
uint64_t go_splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static inline uint64_t synth_key(uint64_t seed, uint32_t species, uint32_t stream) {
    return go_splitmix64(go_splitmix64(seed ^ ((uint64_t)species << 20)) ^ ((uint64_t)stream * 0xd1b54a32d192ed03ULL));
}

void go_synth_genome(uint64_t seed, uint32_t species, uint32_t member, uint64_t length,
                     double sub_rate, uint8_t *out) {
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    uint64_t ka = synth_key(seed, species, 0), km = synth_key(seed, species, member + 1);
    uint32_t thr = (uint32_t)(sub_rate * 4294967296.0);
    for (uint64_t p = 0; p < length; p++) {
        uint64_t w = go_splitmix64(ka + (p >> 5));
        uint32_t base = (uint32_t)(w >> (2 * (p & 31))) & 3;
        uint64_t u = go_splitmix64(km + p);
        if ((uint32_t)(u >> 32) < thr) base = (base + 1 + (uint32_t)((u >> 8) % 3)) & 3;
        out[p] = (uint8_t)ACGT[base];
    }
}
