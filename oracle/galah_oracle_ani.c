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
 * (Shaw & Yu 2023: FracMinHash seeds k=15 c=125, ~20 kb query chunks, seed matches must be
 * COLINEAR within a chunk, ANI from seed containment^(1/k), aligned fraction gate, two-decimal TSV
 * output) without skani's gap-cost chaining DP and learned regression.  It defines what the HIP
 * ani_pairs kernel must reproduce; it makes no claim to reproduce skani's floats.
 *
 * Definition (all integer until the final pow):
 *   stream G  = for each FASTA record: normalised bytes, then one 'N'          (length L)
 *   seed at p = window G[p..p+k) all ACGT (k <= 16); fwd / rev = 2-bit codes of the k-mer and of its reverse
 *               complement (A0 C1 G2 T3, first base most significant); code = min(fwd, rev), strand = (rev < fwd);
 *               kept iff (u32)((0 - 2 - fwd - rev) * (0x85EBCA6B << (32 - 2k))) < (2^32-1) / c  -- the selection key
 *               fwd + rev mod 4^k is the same on both strands, like the canonical code, and costs the device one add
 *               where min() of two masked fields cost four instructions on every base; the shifted odd multiplier is a
 *               bijective mix of the key's 2k bits (and ignores whatever a register holds above them)
 *   density   : the `c` above is PER GENOME.  A genome gets the base density c (125; 30 with --small-genomes) when it is
 *               long enough to hold ~8192 seeds at it, else a four times denser one, and so on down to every k-mer:
 *                   c_g = c;  while (c_g > 1 && L < 8192 * c_g) c_g = max(1, c_g / 4)        (125 -> 31 -> 7 -> 1)
 *               (skani's `--small-genomes` = `-c 30` is the reference's own remedy for the same noise,
 *               src/skani.rs:152-153; here it is chosen per genome -- a 200 kb plasmid or a 5 kb contig carries as many
 *               seeds as a genome does).  FracMinHash samples are nested (hash < 2^32/c), so a PAIR is evaluated at the
 *               sparser of its two densities, c_pair = max(c_q, c_r): seeds of the denser genome whose selection hash is
 *               not below (2^32-1)/c_pair are ignored, and its T_c counts only the seeds that remain
 *   chunk(p)  = p / chunk_len  (chunk_len <= 32768; at most 65535 chunks per genome)
 *   anchor    = (seed a of q, seed b of r) with equal codes; orientation o = strand_a ^ strand_b;
 *               band = o ? ((pos_b + pos_a) >> 12 & 7) | 8 : ((2 (pos_b - pos_a) + 1 + 4096) >> 13) & 7   (mod 2^32)
 *               -- anchors of one alignment share the orientation and, up to indels, the diagonal: 4 kb bands,
 *               placed so that exchanging q and r only mirrors them (the estimate is symmetric)
 *   votes     : a seed whose anchors all fall into ONE band votes for it: V_q[chunk(a)][band] = number of such seeds; a
 *               seed with anchors in several bands (a repeat: insertion sequences, rRNA operons) is counted in
 *               A_q[chunk(a)] instead; V_r, A_r likewise
 *   per chunk : T_c = #seeds; S_c = sum of V[band] over the bands with V[band] >= 3 -- single-copy seed matches count
 *               only when at least three seeds of the chunk agree on orientation and diagonal (a chunk of a
 *               fragmented assembly may hold several contigs, each with its own diagonal; a chunk across a
 *               rearrangement two); M_c = min(T_c, S_c + (S_c > 0 ? A_c : 0)) -- a repeat seed is a match, once, where
 *               the chunk aligns anyway, and cannot align a chunk on its own;
 *               chunk aligned iff T_c >= 1 and M_c * 10000 >= 510 * T_c   (0.82^15 ~ 0.0510)
 *   M,T       = the (M_c, T_c) of the LOWER MEDIAN containment M_c/T_c over the aligned chunks of the SHORTER genome
 *               (smaller L; both genomes' chunks when the lengths are equal -- so the value is symmetric in the pair);
 *               with FEWER THAN NINE such chunks, their pooled counts (sum M_c, sum T_c) instead: a lower median of two
 *               is the minimum, of four the second smallest, of eight the fourth -- measured (round 5,
 *               profiles/r05_ani_few_chunks.txt) at -0.24 points of bias and errors up to 1.1 at 30 kb, -0.12 and 0.6-0.8
 *               at 70 kb, against 0.03 / 0.33 and 0.06 / 0.46 pooled; from ten chunks on the two rules differ by less
 *               than the spread of either, and the median keeps what it was chosen for (a chunk across a rearrangement or
 *               a foreign island cannot move it).
 *               A chunk of the longer genome is diluted wherever the shorter one ends inside it: a 4 kb contig inside a
 *               50 kb one fills a fifth of the one chunk it touches, and the median over both directions came out at
 *               86 % for a 96 % pair; the shorter genome's chunks are covered whole wherever it aligns at all (skani
 *               likewise computes ANI over the chunks of the query).  Exact order by cross-multiplication; a robust
 *               per-chunk statistic rather than a pooled count, and with this choice the reference's own membership
 *               tests src/clusterer.rs:631-690 are reproduced, see tests
 *   AF_x      = (bases in aligned chunks of x) / L_x
 *   ANI%      = 100 * (M/T)^(1/k); 0 if no chunk aligned or (AF_q < min_af and AF_r < min_af)
 *               (a chance match of a mutated seed elsewhere in the other genome only counts if it falls into a band
 *               that already holds two other seeds of the chunk, so the chance term the unchained form needed is gone)
 *   returned  = strtof(sprintf("%.2f", ANI%))   -- skani prints two decimals, galah parses f32
 */
#define _GNU_SOURCE
#include "galah_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define GO_ANI_BANDS 16
#define GO_ANI_SLOTS 17   /* per chunk: 16 band counters + the seeds whose anchors fall into more than one band */
#define GO_ANI_BAND_SHIFT 12
#define GO_ANI_MIN_COLINEAR 3
#define GO_ANI_POOL_BELOW 9     /* fewer aligned chunks than this are pooled instead of taking their median (scripts/ani_few_chunks.py, profiles/r05_ani_few_chunks.txt) */
uint32_t go_ani_definition_version(void) { return GO_ANI_DEFINITION_VERSION; }
#define GO_ANI_SEEDS_WANTED 8192   /* a genome shorter than this many seeds' worth of bases is seeded four times denser */

typedef struct { uint32_t code, idx; } code_ref;

struct go_ani_sketch {
    uint32_t k, c, chunk;   /* c: the density THIS genome was seeded at (go_ani_density of its length and the base c) */
    uint32_t *sel;       /* selection hash per seed (< (2^32-1)/c): what a sparser partner's threshold is applied to */
    uint64_t length;     /* L */
    size_t n;            /* seeds in position order */
    uint64_t *h;         /* code per seed (position order; u64 for the accessor's sake) */
    uint32_t *chunk_id;  /* chunk per seed */
    uint32_t *pos;       /* start position per seed */
    uint8_t *strand;     /* 1: the canonical code is the reverse complement */
    code_ref *sorted;    /* (code, seed index) sorted by code */
    uint32_t n_chunks;
    uint32_t *chunk_total; /* T_c */
};

static int cmp_code_ref(const void *a, const void *b) {
    const code_ref *x = (const code_ref *)a, *y = (const code_ref *)b;
    if (x->code != y->code) return (x->code > y->code) - (x->code < y->code);
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* the density a genome of `len` stream bytes is seeded at, given the base density c */
uint32_t go_ani_density(uint64_t len, uint32_t c) {
    uint32_t t = c ? c : 1;
    while (t > 1 && len < (uint64_t)GO_ANI_SEEDS_WANTED * t) t = t / 4 ? t / 4 : 1;
    return t;
}

go_ani_sketch *go_ani_sketch_bytes(const uint8_t *g, size_t n, uint32_t k, uint32_t c, uint32_t chunk) {
    go_ani_sketch *s = (go_ani_sketch *)calloc(1, sizeof(*s));
    c = go_ani_density(n, c);
    s->k = k; s->c = c; s->chunk = chunk; s->length = n;
    s->n_chunks = (uint32_t)((n + chunk - 1) / chunk);
    s->chunk_total = (uint32_t *)calloc((size_t)s->n_chunks + 1, sizeof(uint32_t));
    size_t cap = n / (c ? c : 1) + 1024;
    s->h = (uint64_t *)malloc(cap * sizeof(uint64_t));
    s->chunk_id = (uint32_t *)malloc(cap * sizeof(uint32_t));
    s->pos = (uint32_t *)malloc(cap * sizeof(uint32_t));
    s->strand = (uint8_t *)malloc(cap);
    s->sel = (uint32_t *)malloc(cap * sizeof(uint32_t));
    const uint32_t thr = UINT32_MAX / c, mix = 0x85EBCA6Bu << (32 - 2 * k);   /* k <= 16 */
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
        const uint32_t code = (uint32_t)(fwd < rev ? fwd : rev);
        const uint32_t sel = (uint32_t)((0u - 2u - (uint32_t)fwd - (uint32_t)rev) * mix);
        if (sel >= thr) continue;
        if (s->n == cap) {
            cap *= 2;
            s->sel = (uint32_t *)realloc(s->sel, cap * sizeof(uint32_t));
            s->h = (uint64_t *)realloc(s->h, cap * sizeof(uint64_t));
            s->chunk_id = (uint32_t *)realloc(s->chunk_id, cap * sizeof(uint32_t));
            s->pos = (uint32_t *)realloc(s->pos, cap * sizeof(uint32_t));
            s->strand = (uint8_t *)realloc(s->strand, cap);
        }
        const size_t start = p + 1 - k;
        s->h[s->n] = code;
        s->sel[s->n] = sel;
        s->pos[s->n] = (uint32_t)start;
        s->strand[s->n] = rev < fwd;
        s->chunk_id[s->n] = (uint32_t)(start / chunk);
        s->chunk_total[start / chunk]++;
        s->n++;
    }
    s->sorted = (code_ref *)malloc((s->n + 1) * sizeof(code_ref));
    for (size_t i = 0; i < s->n; i++) { s->sorted[i].code = (uint32_t)s->h[i]; s->sorted[i].idx = (uint32_t)i; }
    qsort(s->sorted, s->n, sizeof(code_ref), cmp_code_ref);
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
    free(s->h); free(s->chunk_id); free(s->pos); free(s->strand); free(s->sel); free(s->sorted); free(s->chunk_total); free(s);
}
size_t go_ani_sketch_nseeds(const go_ani_sketch *s) { return s->n; }
const uint64_t *go_ani_sketch_seeds(const go_ani_sketch *s) { return s->h; }
const uint32_t *go_ani_sketch_chunks(const go_ani_sketch *s) { return s->chunk_id; }
const uint32_t *go_ani_sketch_positions(const go_ani_sketch *s) { return s->pos; }
const uint8_t *go_ani_sketch_strands(const go_ani_sketch *s) { return s->strand; }
uint64_t go_ani_sketch_length(const go_ani_sketch *s) { return s->length; }
uint32_t go_ani_sketch_density(const go_ani_sketch *s) { return s->c; }

typedef struct { uint64_t m, t; } chunk_frac;

/* exact order of the fractions m/t (t >= 1): cross-multiplication, no floating point */
static int cmp_frac(const void *pa, const void *pb) {
    const chunk_frac *a = (const chunk_frac *)pa, *b = (const chunk_frac *)pb;
    uint64_t l = a->m * b->t, r = b->m * a->t;
    return (l > r) - (l < r);
}

static size_t lower_bound_code(const code_ref *a, size_t n, uint32_t code) {
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (a[mid].code < code) lo = mid + 1; else hi = mid; }
    return lo;
}

/* band votes of both genomes: vq[chunk][band], vr[chunk][band] */
static void ani_votes(const go_ani_sketch *q, const go_ani_sketch *r, uint32_t thr, uint32_t *vq, uint32_t *vr) {
    uint16_t *rmask = (uint16_t *)calloc(r->n + 1, sizeof(uint16_t));
    for (size_t a = 0; a < q->n; a++) {
        const uint32_t code = (uint32_t)q->h[a];
        uint32_t qmask = 0;
        if (q->sel[a] >= thr) continue;   /* not a seed at the pair's density (equal codes have equal hashes: the r side agrees) */
        for (size_t x = lower_bound_code(r->sorted, r->n, code); x < r->n && r->sorted[x].code == code; x++) {
            const uint32_t b = r->sorted[x].idx;
            const uint32_t o = (uint32_t)(q->strand[a] ^ r->strand[b]);
            /* same orientation: bands of the odd number 2 (pos_b - pos_a) + 1, centred on 0 -- the boundaries are even, so
             * swapping q and r negates the number and mirrors the bands: ANI(q, r) == ANI(r, q) */
            const uint32_t band = o ? (((r->pos[b] + q->pos[a]) >> GO_ANI_BAND_SHIFT) & 7u) | 8u
                                    : ((2u * (r->pos[b] - q->pos[a]) + 1u + (1u << GO_ANI_BAND_SHIFT)) >> (GO_ANI_BAND_SHIFT + 1)) & 7u;
            qmask |= 1u << band;
            rmask[b] |= (uint16_t)(1u << band);
        }
        if (qmask) vq[(size_t)q->chunk_id[a] * GO_ANI_SLOTS + ((qmask & (qmask - 1)) ? GO_ANI_BANDS : (uint32_t)__builtin_ctz(qmask))]++;
    }
    for (size_t b = 0; b < r->n; b++) {
        const uint32_t m = rmask[b];
        if (m) vr[(size_t)r->chunk_id[b] * GO_ANI_SLOTS + ((m & (m - 1)) ? GO_ANI_BANDS : (uint32_t)__builtin_ctz(m))]++;
    }
    free(rmask);
}

/* appends (M_c, T_c) of every aligned chunk of x and adds its aligned bases */
static void ani_collect(const go_ani_sketch *x, const uint32_t *totals, const uint32_t *votes, int listed, chunk_frac *out, size_t *n_out, uint64_t *aligned_bases) {
    for (uint32_t c = 0; c < x->n_chunks; c++) {
        const uint64_t tc = totals[c];
        if (tc < 1) continue;
        const uint32_t *v = votes + (size_t)c * GO_ANI_SLOTS;
        uint64_t mc = 0;
        for (uint32_t band = 0; band < GO_ANI_BANDS; band++) if (v[band] >= GO_ANI_MIN_COLINEAR) mc += v[band];
        if (mc) mc += v[GO_ANI_BANDS];   /* repeats: matched once, and only in a chunk that aligns by its single-copy seeds */
        if (mc > tc) mc = tc;
        if (mc * 10000 >= 510 * tc) {
            if (listed) { out[*n_out].m = mc; out[*n_out].t = tc; (*n_out)++; }
            uint64_t lo = (uint64_t)c * x->chunk, hi = lo + x->chunk;
            if (hi > x->length) hi = x->length;
            *aligned_bases += hi - lo;
        }
    }
}

/* T_c of x at the pair's density: its own totals when that is its own density, else a recount of the seeds that remain */
static uint32_t *ani_totals(const go_ani_sketch *x, uint32_t thr) {
    uint32_t *t = (uint32_t *)calloc((size_t)x->n_chunks + 1, sizeof(uint32_t));
    if (UINT32_MAX / x->c == thr) memcpy(t, x->chunk_total, (size_t)x->n_chunks * sizeof(uint32_t));
    else for (size_t a = 0; a < x->n; a++) if (x->sel[a] < thr) t[x->chunk_id[a]]++;
    return t;
}

/* detail (nullable): M, T of the median chunk, number of aligned chunks, aligned bases of q, aligned bases of r, c_pair */
static float ani_pair_rule(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, float *af_q, float *af_r,
                           uint64_t detail[6], size_t pool_below) {
    uint64_t bq = 0, br = 0;
    size_t n = 0;
    const uint32_t c_pair = q->c > r->c ? q->c : r->c, thr = UINT32_MAX / c_pair;
    chunk_frac *fr = (chunk_frac *)malloc(((size_t)q->n_chunks + r->n_chunks + 1) * sizeof(chunk_frac));
    uint32_t *vq = (uint32_t *)calloc(((size_t)q->n_chunks + 1) * GO_ANI_SLOTS, sizeof(uint32_t));
    uint32_t *vr = (uint32_t *)calloc(((size_t)r->n_chunks + 1) * GO_ANI_SLOTS, sizeof(uint32_t));
    uint32_t *tq = ani_totals(q, thr), *tr = ani_totals(r, thr);
    ani_votes(q, r, thr, vq, vr);
    ani_collect(q, tq, vq, q->length <= r->length, fr, &n, &bq);   /* the shorter genome's chunks (both at equal length) */
    ani_collect(r, tr, vr, r->length <= q->length, fr, &n, &br);
    free(vq); free(vr); free(tq); free(tr);
    double afq = q->length ? (double)bq / (double)q->length : 0.0;
    double afr = r->length ? (double)br / (double)r->length : 0.0;
    if (af_q) *af_q = (float)afq;
    if (af_r) *af_r = (float)afr;
    if (detail) { detail[0] = detail[1] = 0; detail[2] = n; detail[3] = bq; detail[4] = br; detail[5] = c_pair; }
    if (n == 0) { free(fr); return 0.0f; }
    chunk_frac med;
    if (n < pool_below) {
        /* too few chunks for a median to be robust (of two it is the minimum): the pooled count over them */
        med.m = med.t = 0;
        for (size_t e = 0; e < n; e++) { med.m += fr[e].m; med.t += fr[e].t; }
    } else {
        /* lower median of the per-chunk containments over the listed chunks */
        qsort(fr, n, sizeof(chunk_frac), cmp_frac);
        med = fr[(n - 1) / 2];
        /* every member of the median's tie group holds the same fraction: the smallest (m, t) of the group is reported */
        for (size_t e = 0; e < n; e++)
            if (fr[e].m * med.t == med.m * fr[e].t && (fr[e].m < med.m || (fr[e].m == med.m && fr[e].t < med.t))) med = fr[e];
    }
    free(fr);
    if (detail) { detail[0] = med.m; detail[1] = med.t; }
    if (afq < (double)min_af_fraction && afr < (double)min_af_fraction) return 0.0f;
    const double c = (double)med.m / (double)med.t;
    double ani = 100.0 * pow(c, 1.0 / (double)q->k);
    char txt[64];
    snprintf(txt, sizeof txt, "%.2f", ani);
    return strtof(txt, NULL);
}

float go_ani_pair_detail(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, float *af_q, float *af_r,
                         uint64_t detail[6]) {
    return ani_pair_rule(q, r, min_af_fraction, af_q, af_r, detail, GO_ANI_POOL_BELOW);
}

/* MEASUREMENT ONLY (scripts/ani_few_chunks.py): the same with another pooling limit (0: always the median) */
float go_ani_pair_pool_below(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, uint32_t pool_below, float *af_q,
                             float *af_r, uint64_t detail[6]) {
    return ani_pair_rule(q, r, min_af_fraction, af_q, af_r, detail, pool_below);
}

float go_ani_pair(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, float *af_q, float *af_r) {
    return go_ani_pair_detail(q, r, min_af_fraction, af_q, af_r, NULL);
}

/* ------------------------------------------------------------------------------------------------------------------
 * EXPERIMENTAL second mode (round 4; measurement only -- scripts/ani_chain_vs_band.py; nothing in the product or in the
 * frozen golden uses it): the ORDERED COLINEAR CHAIN of Shaw & Yu 2023 in place of the band vote, and/or denominators
 * limited to the SPAN a chunk's matches cover.  Restated from the paper's description (skani's source is not here):
 *   anchors of a chunk  = (seed a of x in the chunk, seed b of y) with equal codes, at the pair's density; at most
 *                         GO_CHAIN_MAX_OCC anchors per x seed (a more repetitive seed is ignored, as skani does)
 *   orientation groups  : o = strand_a ^ strand_b; u = o ? -pos_b : pos_b, so that a colinear run ascends in (pos_a, u)
 *   chain DP            : anchors sorted by (o, pos_a, u); f(i) = 100 + max(0, max_j f(j) - |(u_i-u_j) - (x_i-x_j)|) over the
 *                         GO_CHAIN_LOOKBACK predecessors j of the same orientation with x_j < x_i and u_j < u_i  (gap cost
 *                         = the diagonal shift in bases, one anchor = 100: a 100-base indel costs one anchor)
 *   chains of a chunk   : best end first, back-tracked until a used anchor; kept when it holds >= 3 distinct x seeds and
 *                         its x span does not overlap a kept chain's (several contigs / a rearrangement inside a chunk)
 *   M_c                 = distinct x seeds of the kept chains
 *   T_c                 = GO_CHAIN_SPAN: x seeds (pair density) with first <= pos <= last of each kept chain, and the
 *                         aligned bases are the spans'; otherwise the chunk's total and the chunk's bases (as the band vote)
 *   GO_BAND_SPAN        : the BAND VOTE's matches with span denominators: M_c as the frozen definition, T_c = the chunk's
 *                         seeds between the first and the last seed that voted for a qualifying band
 *   aggregate           : GO_AGG_MEDIAN lower median over the shorter genome's aligned chunks (the frozen rule),
 *                         GO_AGG_WMEDIAN the same weighted by T_c, GO_AGG_POOLED sum M / sum T
 */
#define GO_CHAIN_MAX_OCC 32
#define GO_CHAIN_LOOKBACK 50
enum { GO_CHAIN_SPAN = 1, GO_AGG_WMEDIAN = 2, GO_AGG_POOLED = 4, GO_BAND_SPAN = 8, GO_BAND_SUB = 16 /* closing distance D in bits 8..11 */ };

typedef struct { int64_t x, u; uint32_t o, a; } chain_anchor;
static int cmp_anchor(const void *pa, const void *pb) {
    const chain_anchor *a = (const chain_anchor *)pa, *b = (const chain_anchor *)pb;
    if (a->o != b->o) return (a->o > b->o) - (a->o < b->o);
    if (a->x != b->x) return (a->x > b->x) - (a->x < b->x);
    return (a->u > b->u) - (a->u < b->u);
}
typedef struct { int64_t f; uint32_t i; } chain_end;
static int cmp_end(const void *pa, const void *pb) {
    const chain_end *a = (const chain_end *)pa, *b = (const chain_end *)pb;
    if (a->f != b->f) return (a->f < b->f) - (a->f > b->f);   /* descending score */
    return (a->i > b->i) - (a->i < b->i);
}

/* chunks of x against y: appends the aligned chunks' (M, T) when listed, adds the aligned bases */
static void chain_direction(const go_ani_sketch *x, const go_ani_sketch *y, uint32_t thr, int flags, int listed,
                            chunk_frac *out, size_t *n_out, uint64_t *aligned_bases) {
    size_t a0 = 0;
    chain_anchor *an = NULL; size_t cap = 0;
    for (uint32_t c = 0; c < x->n_chunks; c++) {
        size_t a1 = a0;
        while (a1 < x->n && x->chunk_id[a1] == c) a1++;
        size_t n = 0, tc_full = 0;
        for (size_t a = a0; a < a1; a++) {
            if (x->sel[a] >= thr) continue;
            tc_full++;
            const uint32_t code = (uint32_t)x->h[a];
            size_t lo = lower_bound_code(y->sorted, y->n, code), hi = lo, occ = 0;
            while (hi < y->n && y->sorted[hi].code == code) { if (y->sel[y->sorted[hi].idx] < thr) occ++; hi++; }
            if (occ == 0 || occ > GO_CHAIN_MAX_OCC) continue;
            for (size_t e = lo; e < hi; e++) {
                const uint32_t b = y->sorted[e].idx;
                if (y->sel[b] >= thr) continue;
                if (n == cap) { cap = cap ? 2 * cap : 1024; an = (chain_anchor *)realloc(an, cap * sizeof(*an)); }
                const uint32_t o = (uint32_t)(x->strand[a] ^ y->strand[b]);
                an[n].x = x->pos[a]; an[n].u = o ? -(int64_t)y->pos[b] : (int64_t)y->pos[b]; an[n].o = o; an[n].a = (uint32_t)a; n++;
            }
        }
        uint64_t mc = 0, tc = 0, bases = 0;
        if (n >= GO_ANI_MIN_COLINEAR) {
            qsort(an, n, sizeof(*an), cmp_anchor);
            int64_t *f = (int64_t *)malloc(n * sizeof(int64_t));
            int32_t *prev = (int32_t *)malloc(n * sizeof(int32_t));
            uint8_t *used = (uint8_t *)calloc(n, 1);
            chain_end *ends = (chain_end *)malloc(n * sizeof(chain_end));
            for (size_t i = 0; i < n; i++) {
                int64_t best = 0; int32_t bp = -1;
                for (size_t back = 1; back <= GO_CHAIN_LOOKBACK && back <= i; back++) {
                    const size_t j = i - back;
                    if (an[j].o != an[i].o) break;
                    if (an[j].x >= an[i].x || an[j].u >= an[i].u) continue;
                    int64_t dd = (an[i].u - an[j].u) - (an[i].x - an[j].x);
                    if (dd < 0) dd = -dd;
                    const int64_t cand = f[j] - dd;
                    if (cand > best) { best = cand; bp = (int32_t)j; }
                }
                f[i] = 100 + best; prev[i] = bp;
                ends[i].f = f[i]; ends[i].i = (uint32_t)i;
            }
            qsort(ends, n, sizeof(chain_end), cmp_end);
            int64_t (*kept)[2] = (int64_t (*)[2])malloc(n * sizeof(int64_t[2]));
            size_t nk = 0;
            for (size_t e = 0; e < n; e++) {
                int32_t i = (int32_t)ends[e].i;
                if (used[i]) continue;
                int64_t last = an[i].x, first = last;
                uint64_t seeds = 0; uint32_t prev_a = UINT32_MAX;
                for (; i >= 0 && !used[i]; i = prev[i]) { used[i] = 1; first = an[i].x; if (an[i].a != prev_a) { seeds++; prev_a = an[i].a; } }
                if (seeds < GO_ANI_MIN_COLINEAR) continue;
                int overlap = 0;
                for (size_t z = 0; z < nk; z++) if (first <= kept[z][1] && kept[z][0] <= last) { overlap = 1; break; }
                if (overlap) continue;
                kept[nk][0] = first; kept[nk][1] = last; nk++;
                mc += seeds;
                if (flags & GO_CHAIN_SPAN) {
                    for (size_t a = a0; a < a1; a++) if (x->sel[a] < thr && (int64_t)x->pos[a] >= first && (int64_t)x->pos[a] <= last) tc++;
                    bases += (uint64_t)(last - first) + x->k;
                }
            }
            free(f); free(prev); free(used); free(ends); free(kept);
        }
        if (!(flags & GO_CHAIN_SPAN)) {
            tc = tc_full;
            uint64_t lo = (uint64_t)c * x->chunk, hi = lo + x->chunk;
            if (hi > x->length) hi = x->length;
            bases = hi - lo;
        }
        if (mc > tc) mc = tc;
        if (tc >= 1 && mc >= GO_ANI_MIN_COLINEAR && mc * 10000 >= 510 * tc) {
            if (listed) { out[*n_out].m = mc; out[*n_out].t = tc; (*n_out)++; }
            *aligned_bases += bases;
        }
        a0 = a1;
    }
    free(an);
}

/* the band vote's matches with span denominators (GO_BAND_SPAN) */
static void band_span_direction(const go_ani_sketch *x, const go_ani_sketch *y, uint32_t thr, int swap, int listed,
                                chunk_frac *out, size_t *n_out, uint64_t *aligned_bases) {
    uint32_t *mask = (uint32_t *)calloc(x->n + 1, sizeof(uint32_t));
    for (size_t a = 0; a < x->n; a++) {
        if (x->sel[a] >= thr) continue;
        const uint32_t code = (uint32_t)x->h[a];
        for (size_t e = lower_bound_code(y->sorted, y->n, code); e < y->n && y->sorted[e].code == code; e++) {
            const uint32_t b = y->sorted[e].idx;
            if (y->sel[b] >= thr) continue;
            const uint32_t o = (uint32_t)(x->strand[a] ^ y->strand[b]);
            /* the band of the frozen definition is stated for (q, r); x may be r */
            const uint32_t pq = swap ? y->pos[b] : x->pos[a], pr = swap ? x->pos[a] : y->pos[b];
            const uint32_t band = o ? (((pr + pq) >> GO_ANI_BAND_SHIFT) & 7u) | 8u
                                    : ((2u * (pr - pq) + 1u + (1u << GO_ANI_BAND_SHIFT)) >> (GO_ANI_BAND_SHIFT + 1)) & 7u;
            mask[a] |= 1u << band;
        }
    }
    size_t a0 = 0;
    for (uint32_t c = 0; c < x->n_chunks; c++) {
        size_t a1 = a0;
        while (a1 < x->n && x->chunk_id[a1] == c) a1++;
        uint32_t v[GO_ANI_SLOTS] = {0};
        for (size_t a = a0; a < a1; a++) if (mask[a]) v[(mask[a] & (mask[a] - 1)) ? GO_ANI_BANDS : (uint32_t)__builtin_ctz(mask[a])]++;
        uint64_t mc = 0; uint32_t good = 0;
        for (uint32_t b = 0; b < GO_ANI_BANDS; b++) if (v[b] >= GO_ANI_MIN_COLINEAR) { mc += v[b]; good |= 1u << b; }
        if (mc) {
            mc += v[GO_ANI_BANDS];
            int64_t first = -1, last = -1;
            for (size_t a = a0; a < a1; a++) if (mask[a] && !(mask[a] & (mask[a] - 1)) && (mask[a] & good)) { if (first < 0) first = x->pos[a]; last = x->pos[a]; }
            uint64_t tc = 0, m2 = 0;
            for (size_t a = a0; a < a1; a++) if (x->sel[a] < thr && (int64_t)x->pos[a] >= first && (int64_t)x->pos[a] <= last) {
                tc++;
                if (mask[a] && ((mask[a] & (mask[a] - 1)) || (mask[a] & good))) m2++;   /* matches inside the span only */
            }
            mc = m2;
            if (tc >= 1 && mc * 10000 >= 510 * tc) {
                if (listed) { out[*n_out].m = mc; out[*n_out].t = tc; (*n_out)++; }
                *aligned_bases += (uint64_t)(last - first) + x->k;
            }
        }
        a0 = a1;
    }
    free(mask);
}

/* GO_BAND_SUB: the band vote's matches; a chunk is cut into sub-blocks of 2^t positions (t the smallest with (chunk-1) >> t <= 15:
 * 2048 at chunk 20000); a sub-block is SUPPORTED when one of its seeds has an anchor; runs of at most D unsupported sub-blocks
 * between supported ones, or between a supported one and the chunk's edge, are filled in; T_c and the aligned bases count the
 * supported and filled sub-blocks only -- a fully covered chunk keeps its whole denominator (no span-end bias), an uncovered
 * stretch of more than D sub-blocks leaves it. */
static void band_sub_direction(const go_ani_sketch *x, const go_ani_sketch *y, uint32_t thr, int swap, uint32_t D, int listed,
                               chunk_frac *out, size_t *n_out, uint64_t *aligned_bases) {
    uint32_t *mask = (uint32_t *)calloc(x->n + 1, sizeof(uint32_t));
    uint32_t t = 0;
    while (((x->chunk - 1) >> t) > 15) t++;
    for (size_t a = 0; a < x->n; a++) {
        if (x->sel[a] >= thr) continue;
        const uint32_t code = (uint32_t)x->h[a];
        for (size_t e = lower_bound_code(y->sorted, y->n, code); e < y->n && y->sorted[e].code == code; e++) {
            const uint32_t b = y->sorted[e].idx;
            if (y->sel[b] >= thr) continue;
            const uint32_t o = (uint32_t)(x->strand[a] ^ y->strand[b]);
            const uint32_t pq = swap ? y->pos[b] : x->pos[a], pr = swap ? x->pos[a] : y->pos[b];
            const uint32_t band = o ? (((pr + pq) >> GO_ANI_BAND_SHIFT) & 7u) | 8u
                                    : ((2u * (pr - pq) + 1u + (1u << GO_ANI_BAND_SHIFT)) >> (GO_ANI_BAND_SHIFT + 1)) & 7u;
            mask[a] |= 1u << band;
        }
    }
    size_t a0 = 0;
    for (uint32_t c = 0; c < x->n_chunks; c++) {
        size_t a1 = a0;
        while (a1 < x->n && x->chunk_id[a1] == c) a1++;
        uint32_t v[GO_ANI_SLOTS] = {0}, tsb[16] = {0}, sup = 0;
        for (size_t a = a0; a < a1; a++) {
            if (x->sel[a] >= thr) continue;
            const uint32_t sb = (x->pos[a] - c * x->chunk) >> t;
            tsb[sb]++;
            if (mask[a]) { v[(mask[a] & (mask[a] - 1)) ? GO_ANI_BANDS : (uint32_t)__builtin_ctz(mask[a])]++; sup |= 1u << sb; }
        }
        uint64_t lo = (uint64_t)c * x->chunk, hi = lo + x->chunk;
        if (hi > x->length) hi = x->length;
        const uint32_t nsb = (uint32_t)(((hi - lo) - 1) >> t) + 1;
        uint64_t mc = 0;
        for (uint32_t b = 0; b < GO_ANI_BANDS; b++) if (v[b] >= GO_ANI_MIN_COLINEAR) mc += v[b];
        if (mc) {
            mc += v[GO_ANI_BANDS];
            /* closing: fill runs of <= D unsupported sub-blocks bounded by supported ones or the chunk's edges */
            uint32_t closed = sup;
            for (uint32_t s0 = 0; s0 < nsb;) {
                if (sup >> s0 & 1u) { s0++; continue; }
                uint32_t s1 = s0;
                while (s1 < nsb && !(sup >> s1 & 1u)) s1++;
                if (s1 - s0 <= D) for (uint32_t z = s0; z < s1; z++) closed |= 1u << z;
                s0 = s1;
            }
            uint64_t tc = 0, bases = 0;
            for (uint32_t z = 0; z < nsb; z++) if (closed >> z & 1u) {
                tc += tsb[z];
                uint64_t b0 = lo + ((uint64_t)z << t), b1 = b0 + (1ull << t);
                if (b1 > hi) b1 = hi;
                bases += b1 - b0;
            }
            if (mc > tc) mc = tc;
            if (tc >= 1 && mc * 10000 >= 510 * tc) {
                if (listed) { out[*n_out].m = mc; out[*n_out].t = tc; (*n_out)++; }
                *aligned_bases += bases;
            }
        }
        a0 = a1;
    }
    free(mask);
}

float go_ani_pair_mode(const go_ani_sketch *q, const go_ani_sketch *r, float min_af_fraction, int flags, float *af_q, float *af_r,
                       uint64_t detail[6]) {
    uint64_t bq = 0, br = 0;
    size_t n = 0;
    const uint32_t c_pair = q->c > r->c ? q->c : r->c, thr = UINT32_MAX / c_pair;
    chunk_frac *fr = (chunk_frac *)malloc(((size_t)q->n_chunks + r->n_chunks + 1) * sizeof(chunk_frac));
    if (flags & GO_BAND_SUB) {
        band_sub_direction(q, r, thr, 0, (uint32_t)(flags >> 8) & 15u, q->length <= r->length, fr, &n, &bq);
        band_sub_direction(r, q, thr, 1, (uint32_t)(flags >> 8) & 15u, r->length <= q->length, fr, &n, &br);
    } else if (flags & GO_BAND_SPAN) {
        band_span_direction(q, r, thr, 0, q->length <= r->length, fr, &n, &bq);
        band_span_direction(r, q, thr, 1, r->length <= q->length, fr, &n, &br);
    } else {
        chain_direction(q, r, thr, flags, q->length <= r->length, fr, &n, &bq);
        chain_direction(r, q, thr, flags, r->length <= q->length, fr, &n, &br);
    }
    double afq = q->length ? (double)bq / (double)q->length : 0.0;
    double afr = r->length ? (double)br / (double)r->length : 0.0;
    if (af_q) *af_q = (float)afq;
    if (af_r) *af_r = (float)afr;
    if (detail) { detail[0] = detail[1] = 0; detail[2] = n; detail[3] = bq; detail[4] = br; detail[5] = c_pair; }
    if (n == 0) { free(fr); return 0.0f; }
    chunk_frac med;
    if (flags & GO_AGG_POOLED) {
        med.m = med.t = 0;
        for (size_t e = 0; e < n; e++) { med.m += fr[e].m; med.t += fr[e].t; }
    } else {
        qsort(fr, n, sizeof(chunk_frac), cmp_frac);
        if (flags & GO_AGG_WMEDIAN) {
            uint64_t tot = 0, run = 0; size_t e = 0;
            for (size_t z = 0; z < n; z++) tot += fr[z].t;
            for (; e < n; e++) { run += fr[e].t; if (2 * run >= tot) break; }
            med = fr[e < n ? e : n - 1];
        } else med = fr[(n - 1) / 2];
    }
    free(fr);
    if (detail) { detail[0] = med.m; detail[1] = med.t; }
    if (afq < (double)min_af_fraction && afr < (double)min_af_fraction) return 0.0f;
    double ani = 100.0 * pow((double)med.m / (double)med.t, 1.0 / (double)q->k);
    char txt[64];
    snprintf(txt, sizeof txt, "%.2f", ani);
    return strtof(txt, NULL);
}
