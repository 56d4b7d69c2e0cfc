/*
 * galah_oracle_ani.c -- CPU statement of the "skani-equivalent" ANI used on candidate pairs.
 *
 * TEST INFRASTRUCTURE ONLY (see galah_oracle.h).
 *
 * PARITY UNPINNED.  The reference obtains ANI by fork/exec of the external `skani` binary
 * (src/skani.rs:718-788, `skani dist --min-af X -q A -r B`, TSV column 2 parsed as f32
 * percent, 0.0 when skani prints no row).  skani (>=0.2.2, pixi.lock:124) is not in
 * /root/reference, is not installed here, and no reference test asserts a skani float
 * (SURVEY.md 8c).  What follows is therefore a BUILD-DEFINED estimator in skani's style
 * (Shaw & Yu 2023: FracMinHash seeds k=15 c=125, ~20 kb query chunks, ANI from seed
 * containment^(1/k), aligned fraction gate, two-decimal TSV output) WITHOUT skani's
 * colinear chaining and learned regression, and with a 32-bit invertible seed hash.  It defines what the HIP ani_pairs kernel
 * must reproduce; it makes no claim to reproduce skani's floats.
 *
 * Definition (all integer until the final pow):
 *   stream G  = for each FASTA record: normalised bytes, then one 'N'          (length L)
 *   seed at p = window G[p..p+k) all ACGT (k <= 16); code = min(2-bit fwd, 2-bit revcomp) (A0 C1 G2
 *               T3, first base most significant); kept iff fmix32(code) < (2^32-1) / c
 *   chunk(p)  = p / chunk_len
 *   q->r      : for every chunk of q: T_c = #seeds, M_c = #seeds whose h is a seed of r;
 *               chunk aligned iff T_c >= 1 and M_c * 10000 >= 510 * T_c   (0.82^15 ~ 0.0510)
 *   M,T       = the (M_c, T_c) of the LOWER MEDIAN containment M_c/T_c over the aligned chunks of BOTH
 *               directions (exact order by cross-multiplication; skani likewise reports a robust
 *               per-chunk statistic rather than a pooled count, and with this choice the reference's
 *               own membership tests src/clusterer.rs:631-690 are reproduced, see tests)
 *   AF_x      = (bases in aligned chunks of x) / L_x
 *   p0        = (L_q + L_r) / 4^k      -- chance that a k-mer occurs somewhere in a genome of the mean length
 *   c         = max(0, (M/T - p0) / (1 - p0))   -- observed containment = c + (1 - c) p0 (matches are by value, unchained)
 *   ANI%      = 100 * c^(1/k); 0 if no chunk aligned or (AF_q < min_af and AF_r < min_af)
 *   returned  = strtof(sprintf("%.2f", ANI%))   -- skani prints two decimals, galah parses f32
 */
#define _GNU_SOURCE
#include "galah_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

struct go_ani_sketch {
    uint32_t k, c, chunk;
    uint64_t length;     /* L */
    size_t n;            /* seeds in position order */
    uint64_t *h;         /* hash per seed (position order) */
    uint32_t *chunk_id;  /* chunk per seed */
    size_t nd;           /* distinct */
    uint64_t *sorted;    /* sorted distinct hashes */
    uint32_t n_chunks;
};

/* Seed-selection hash: MurmurHash3's 32-bit finaliser (a bijection on 32 bits) over the canonical
 * 2-bit k-mer code (k <= 16).  skani uses minimap2's invertible 64-bit mix for the same purpose; any
 * invertible mix gives a FracMinHash, and this one is 8 integer instructions per position on the GPU
 * instead of ~30 (the seeding pass hashes every position of every genome). */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

go_ani_sketch *go_ani_sketch_bytes(const uint8_t *g, size_t n, uint32_t k, uint32_t c, uint32_t chunk) {
    go_ani_sketch *s = (go_ani_sketch *)calloc(1, sizeof(*s));
    s->k = k; s->c = c; s->chunk = chunk; s->length = n;
    s->n_chunks = (uint32_t)((n + chunk - 1) / chunk);
    size_t cap = n / (c ? c : 1) + 1024;
    s->h = (uint64_t *)malloc(cap * sizeof(uint64_t));
    s->chunk_id = (uint32_t *)malloc(cap * sizeof(uint32_t));
    const uint32_t thr = UINT32_MAX / c;
    const uint64_t mask = (k < 32) ? ((1ULL << (2 * k)) - 1) : UINT64_MAX;
    uint64_t fwd = 0, rev = 0;
    size_t good = 0;
    for (size_t p = 0; p < n; p++) {
        int b;
        switch (g[p]) { case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break; case 'T': b = 3; break; default: b = -1; }
        if (b < 0) { good = 0; fwd = rev = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)b) & mask;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k - 1)));
        if (++good < k) continue;
        uint64_t code = fwd < rev ? fwd : rev;
        if (fmix32((uint32_t)code) >= thr) continue;
        uint64_t h = code; /* the mix is a bijection: seeds are identified by their code */
        if (s->n == cap) {
            cap *= 2;
            s->h = (uint64_t *)realloc(s->h, cap * sizeof(uint64_t));
            s->chunk_id = (uint32_t *)realloc(s->chunk_id, cap * sizeof(uint32_t));
        }
        s->h[s->n] = h;
        s->chunk_id[s->n] = (uint32_t)((p + 1 - k) / chunk);
        s->n++;
    }
    s->sorted = (uint64_t *)malloc((s->n + 1) * sizeof(uint64_t));
    memcpy(s->sorted, s->h, s->n * sizeof(uint64_t));
    qsort(s->sorted, s->n, sizeof(uint64_t), cmp_u64);
    size_t m = 0;
    for (size_t i = 0; i < s->n; i++)
        if (m == 0 || s->sorted[i] != s->sorted[m - 1]) s->sorted[m++] = s->sorted[i];
    s->nd = m;
    return s;
}

/* Build stream G from a FASTA file: per record, normalised bytes then 'N'. */
int go_ani_sketch_file(const char *path, uint32_t k, uint32_t c, uint32_t chunk, go_ani_sketch **out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return -1;
    size_t cap = 1 << 22, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (;;) {
        if (n == cap) { cap *= 2; buf = (uint8_t *)realloc(buf, cap); }
        int r = gzread(f, buf + n, (unsigned)((cap - n) > (1u << 30) ? (1u << 30) : (cap - n)));
        if (r < 0) { free(buf); gzclose(f); return -1; }
        if (r == 0) break;
        n += (size_t)r;
    }
    gzclose(f);
    uint8_t *g = (uint8_t *)malloc(n + 16);
    size_t m = 0, p = 0;
    while (p < n && (buf[p] == '\n' || buf[p] == '\r')) p++;
    if (p < n && buf[p] != '>') { free(buf); free(g); return -2; }
    while (p < n) {
        while (p < n && buf[p] != '\n') p++;
        if (p < n) p++;
        size_t start = p;
        while (p < n) {
            if (buf[p] == '>' && (p == start || buf[p - 1] == '\n')) break;
            p++;
        }
        m += go_normalize(buf + start, p - start, g + m);
        g[m++] = 'N';
    }
    *out = go_ani_sketch_bytes(g, m, k, c, chunk);
    free(buf); free(g);
    return 0;
}

void go_ani_sketch_free(go_ani_sketch *s) {
    if (!s) return;
    free(s->h); free(s->chunk_id); free(s->sorted); free(s);
}
size_t go_ani_sketch_nseeds(const go_ani_sketch *s) { return s->n; }
const uint64_t *go_ani_sketch_seeds(const go_ani_sketch *s) { return s->h; }
const uint32_t *go_ani_sketch_chunks(const go_ani_sketch *s) { return s->chunk_id; }
uint64_t go_ani_sketch_length(const go_ani_sketch *s) { return s->length; }

static int contains_sorted(const uint64_t *a, size_t n, uint64_t x) {
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && a[lo] == x;
}

typedef struct { uint64_t m, t; } chunk_frac;

/* exact order of the fractions m/t (t >= 1): cross-multiplication, no floating point */
static int cmp_frac(const void *pa, const void *pb) {
    const chunk_frac *a = (const chunk_frac *)pa, *b = (const chunk_frac *)pb;
    uint64_t l = a->m * b->t, r = b->m * a->t;
    return (l > r) - (l < r);
}

/* one direction: appends (M_c, T_c) of every aligned chunk of q and adds its aligned bases */
static void ani_direction(const go_ani_sketch *q, const go_ani_sketch *r, chunk_frac *out, size_t *n_out, uint64_t *aligned_bases) {
    size_t i = 0;
    while (i < q->n) {
        uint32_t c = q->chunk_id[i];
        uint64_t tc = 0, mc = 0;
        while (i < q->n && q->chunk_id[i] == c) {
            tc++;
            mc += (uint64_t)contains_sorted(r->sorted, r->nd, q->h[i]);
            i++;
        }
        if (tc >= 1 && mc * 10000 >= 510 * tc) {
            out[*n_out].m = mc; out[*n_out].t = tc; (*n_out)++;
            uint64_t lo = (uint64_t)c * q->chunk, hi = lo + q->chunk;
            if (hi > q->length) hi = q->length;
            *aligned_bases += hi - lo;
        }
    }
}

float go_ani_pair(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, float *af_q, float *af_r) {
    uint64_t bq = 0, br = 0;
    size_t n = 0;
    chunk_frac *fr = (chunk_frac *)malloc(((size_t)q->n_chunks + r->n_chunks + 1) * sizeof(chunk_frac));
    ani_direction(q, r, fr, &n, &bq);
    ani_direction(r, q, fr, &n, &br);
    double afq = q->length ? (double)bq / (double)q->length : 0.0;
    double afr = r->length ? (double)br / (double)r->length : 0.0;
    if (af_q) *af_q = (float)afq;
    if (af_r) *af_r = (float)afr;
    if (n == 0) { free(fr); return 0.0f; }
    /* lower median of the per-chunk containments over the aligned chunks of both directions */
    qsort(fr, n, sizeof(chunk_frac), cmp_frac);
    const chunk_frac med = fr[(n - 1) / 2];
    free(fr);
    if (afq < (double)min_af_fraction && afr < (double)min_af_fraction) return 0.0f;
    /* chance matches: seeds are matched by value against the whole other genome (no chaining), so a query seed whose
     * k-mer mutated still "matches" when the k-mer happens to occur anywhere in the other genome -- probability
     * p0 = L / (4^k / 2) with L the mean of the two lengths (there are 4^k / 2 canonical k-mers, k odd).  The observed
     * containment is c + (1 - c) p0; solve for c.  (+0.03 ANI points at 95 % for 2 Mb genomes, +0.2 at 88 % for 5 Mb.) */
    const double c_obs = (double)med.m / (double)med.t;
    const double p0 = q->k <= 31 ? (double)(q->length + r->length) / (double)(1ull << (2 * q->k)) : 0.0;
    double c = p0 < 1.0 ? (c_obs - p0) / (1.0 - p0) : 0.0;
    if (c < 0.0) c = 0.0;
    double ani = 100.0 * pow(c, 1.0 / (double)q->k);
    char txt[64];
    snprintf(txt, sizeof txt, "%.2f", ani);
    return strtof(txt, NULL);
}
