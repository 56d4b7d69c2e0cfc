// C-ABI of libgalah_hip.so, pair stage: ghip_precluster* -- which form runs, the device filter table, the exact host recheck.
#include "api_internal.h"

using namespace ghip_api;

// ------------------------------------------------------------------------------------ precluster
extern "C" uint64_t ghip_last_pairs_compared(const ghip_ctx *ctx) { return ctx ? ctx->last_pairs : 0; }

// Builds the cuckoo sets and the work rows of the probe-form pair kernel once per sketch matrix.
static int prepare_probe(ghip_ctx *ctx, ghip_sketches *sk) {
    if (sk->probe_ready) return GHIP_OK;
    const size_t slots = ghip_probe_table_slots(sk->s);
    int rc;
    uint32_t *d_flags = nullptr;
    if ((rc = dmalloc(ctx, &sk->d_tables, sk->n * slots))) return rc;
    if ((rc = dmalloc(ctx, &sk->d_tags, sk->n * slots))) return rc;
    if ((rc = dmalloc(ctx, &d_flags, 1))) return rc;
    DeviceFree tmp(ctx); tmp.add(d_flags);
    // the arranged form (ghip_options.probe_arranged, fixed for the life of the matrix's tables): constrained second bucket,
    // B rows dealt to the lanes by bucket residue
    if (ctx->opt.probe_arranged && (rc = dmalloc(ctx, &sk->d_arranged, sk->n * ghip_probe_arranged_slots(sk->s)))) return rc;
    sk->probe_cbits = ghip_probe_constrained_bits(ctx->opt.probe_arranged, sk->s);
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_flags, 0, sizeof(uint32_t), ctx->stream));
    ghip_launch_pair_tables(ctx, sk->d_hashes, sk->d_lens, sk->n, sk->s, sk->d_tables, sk->d_tags, d_flags, sk->d_arranged, sk->probe_cbits);
    sk->n_work = ghip_probe_work_rows(sk->n, ctx->num_cus, &sk->probe_cb, sk->row_start);
    if ((rc = dmalloc(ctx, &sk->d_row_start, sk->row_start.size()))) return rc;
    if ((rc = h2d(ctx, sk->d_row_start, sk->row_start.data(), sk->row_start.size()))) return rc;
    if ((rc = d2h(ctx, &sk->probe_flags, d_flags, 1))) return rc;
    sk->probe_ready = true;
    return GHIP_OK;
}

// replicate_join: a multi-rank caller that prefers the whole list on every rank whenever the join form runs (its
// cost, one pass over all N*s hashes, does not shard) to a share it then has to exchange.
// cmin[total] = smallest common whose reference ANI clears the threshold (src/finch.rs:91:
// `distance >= min_ani as f64`).  The device filter only has to be a superset; the exact
// test is repeated on the host for every emitted pair.  Cached in the context per (min_ani, s, k); ctx->mu held.
int ghip_pair_filter_prepare(ghip_ctx *ctx, uint32_t s, uint32_t k, float min_ani) {
    const double thr = (double)min_ani;
    uint32_t ani_bits;
    memcpy(&ani_bits, &min_ani, 4);
    if (ctx->cmin.valid && ctx->cmin.ani_bits == ani_bits && ctx->cmin.s == s && ctx->cmin.k == k) return GHIP_OK;
    const uint32_t max_total = 2 * s;
    std::vector<uint16_t> cmin(max_total + 2, 0xffff);
    for (uint32_t total = 0; total <= max_total; total++) {
        // finch_ani is non-decreasing in common for a fixed total: the smallest passing common by bisection (s = 10 000
        // would cost 2e8 logarithms the linear way); total = 0 is the NaN corner (ANI 1.0 whatever common is)
        const uint32_t cmax = std::min(total, s);
        if (!(finch_ani(cmax, total, k) >= thr)) continue;   // nothing passes: 0xffff
        uint32_t lo = 0, hi = cmax;                            // invariant: hi passes
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (finch_ani(mid, total, k) >= thr) hi = mid; else lo = mid + 1; }
        cmin[total] = (uint16_t)hi;
    }
    if (ctx->cmin.d_cmin) ghip_pool_free(ctx, ctx->cmin.d_cmin);
    ctx->cmin.valid = false;
    int rc;
    if ((rc = dmalloc(ctx, &ctx->cmin.d_cmin, cmin.size()))) return rc;
    if ((rc = h2d(ctx, ctx->cmin.d_cmin, cmin.data(), cmin.size()))) return rc;
    ctx->cmin.ani_bits = ani_bits; ctx->cmin.s = s; ctx->cmin.k = k; ctx->cmin.valid = true;
    ctx->cmin.floor = ghip_cmin_floor(cmin);
    return GHIP_OK;
}

// The host end of the pair stage: exact reference arithmetic (f64 ANI, threshold, `as f32`) on every candidate the device
// filter let through, then (i, j) order.  filter_share: keep only the pairs with (i + j) mod world == rank.
// (serial on purpose: one f64 log per emitted pair is 1 ms per 45 000 pairs, and spawning threads on the 256-core
// host cost more than that -- measured 4.7 -> 6.6 ms for the stage at 10 000 genomes)
int ghip_pairs_finalize(ghip_ctx *ctx, std::vector<ghip_pair> &host, uint32_t k, float min_ani, size_t n, bool filter_share,
                        uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n) {
    const double thr = (double)min_ani;
    size_t m = 0;
    // finch_ani is a pure function of (common, total, k): its f64 results are kept per context in a table indexed by
    // (total, common) -- one logarithm per distinct value instead of one per candidate and call (45 000 candidates cost
    // 0.9 ms at 10 000 genomes, every step; the values a run meets cluster in a few hundred KB of the table).  ctx->mu held.
    const uint32_t s_max = ctx && ctx->cmin.valid ? ctx->cmin.s : 0;
    const bool memo = ctx && s_max >= 1 && s_max <= 2048;
    if (memo && (ctx->ani_memo_k != k || ctx->ani_memo_s != s_max)) {
        ctx->ani_memo.assign((size_t)(2 * s_max + 1) * (s_max + 1), std::nan(""));
        ctx->ani_memo_k = k; ctx->ani_memo_s = s_max;
    }
    if (host.size() >= 200000 && ctx && ctx->ingest_mu.try_lock()) {
        // long lists (a large collection, a very large family): the f64 values on the context's workers, each over its own
        // range and without the table (its slots are written on first use: not from several threads); -1.0f marks a pair
        // that fails (an ANI is never negative)
        const size_t total_n = host.size(), workers = std::min<size_t>(16, total_n / 50000), per = (total_n + workers - 1) / workers;
        ctx->io.run((int)workers, [&](int w) {
            for (size_t i = std::min(total_n, (size_t)w * per), e = std::min(total_n, ((size_t)w + 1) * per); i < e; i++) {
                host[i].ani = -1.0f;
                if (filter_share && (host[i].i + host[i].j) % world != rank) continue;
                const double ani = finch_ani(host[i].common, host[i].total, k);
                if (ani >= thr) host[i].ani = (float)ani;
            }
        });
        ctx->ingest_mu.unlock();
        for (size_t i = 0; i < total_n; i++) if (host[i].ani != -1.0f) host[m++] = host[i];
    } else
    for (size_t i = 0; i < host.size(); i++) {
        if (filter_share && (host[i].i + host[i].j) % world != rank) continue;
        double ani;
        if (memo && host[i].total <= 2 * s_max && host[i].common <= s_max) {
            double &slot = ctx->ani_memo[(size_t)host[i].total * (s_max + 1) + host[i].common];
            if (std::isnan(slot)) slot = finch_ani(host[i].common, host[i].total, k);
            ani = slot;
        } else ani = finch_ani(host[i].common, host[i].total, k);
        if (ani >= thr) { host[i].ani = (float)ani; host[m++] = host[i]; }
    }
    host.resize(m);
    // (i, j) order: counting sort by i straight into the result (O(m + n)), then the few entries of each i by j --
    // a comparison sort of the whole list costs 0.24 ms at 4 500 hits and 3.8 ms at 45 000
    ghip_pair *res = (ghip_pair *)malloc(std::max<size_t>(m, 1) * sizeof(ghip_pair));
    if (!res) return ghip_set_error(ctx, GHIP_ENOMEM, "out of host memory");
    {
        std::vector<size_t> at(n + 1, 0);
        for (size_t x = 0; x < m; x++) at[host[x].i + 1]++;
        for (size_t g = 0; g < n; g++) at[g + 1] += at[g];
        std::vector<size_t> fill(at.begin(), at.end() - 1);
        for (size_t x = 0; x < m; x++) res[fill[host[x].i]++] = host[x];
        auto sort_rows = [&](size_t g0, size_t g1) {
            for (size_t g = g0; g < g1; g++)
                if (at[g + 1] - at[g] > 1)
                    std::sort(res + at[g], res + at[g + 1], [](const ghip_pair &a, const ghip_pair &b) { return a.j < b.j; });
        };
        if (m >= 200000 && ctx && ctx->ingest_mu.try_lock()) {   // long lists: the rows on the context's workers, equal shares of the ENTRIES
            const size_t workers = std::min<size_t>(16, m / 50000);
            std::vector<size_t> cut(workers + 1, n);
            cut[0] = 0;
            for (size_t w = 1; w < workers; w++) cut[w] = std::lower_bound(at.begin(), at.end(), m * w / workers) - at.begin();
            for (size_t w = 1; w <= workers; w++) cut[w] = std::min(std::max(cut[w], cut[w - 1]), n);
            ctx->io.run((int)workers, [&](int w) { sort_rows(cut[w], cut[w + 1]); });
            ctx->ingest_mu.unlock();
        } else sort_rows(0, n);
    }
    *out_pairs = res; *out_n = m;
    return GHIP_OK;
}

// row_lo > 0: the (new x all) rectangle of an incremental run -- only the pairs (i, j), i < j, with j >= row_lo.
// dense_share: no join; a dense pass over EVERY pair, of which the (i + j) mod world == rank share is returned (what a
// rank of the hash-sharded join owes when its own second stage had to give up).
static int precluster_impl(ghip_ctx *ctx, const ghip_sketches *sk_in, float min_ani, uint32_t rank, uint32_t world,
                           bool replicate_join, size_t row_lo, ghip_pair **out_pairs, size_t *out_n, int *out_replicated,
                           bool dense_share = false) {
    ghip_sketches *sk = const_cast<ghip_sketches *>(sk_in);  // lazily caches the probe-form tables
    if (out_replicated) *out_replicated = 0;
    if (!ctx || !sk || !out_pairs || !out_n || world == 0 || rank >= world) return GHIP_EINVAL;
    if (sk->s > GHIP_MAX_SKETCH_SIZE) return ghip_set_error(ctx, GHIP_EINVAL, "sketch size above 65535 is not supported");
    if (row_lo > sk->n || (row_lo && world > 1)) return ghip_set_error(ctx, GHIP_EINVAL, "row_lo must not exceed the sketch count (single rank only)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    *out_pairs = nullptr; *out_n = 0;
    const size_t n = sk->n;
    const uint32_t s = sk->s, k = sk->k;
    ctx->last_pairs = 0;
    if (n < 2) return GHIP_OK;

    DeviceFree tmp(ctx);
    int rc = ghip_pair_filter_prepare(ctx, s, k, min_ani);
    if (rc) return rc;
    uint16_t *d_cmin = ctx->cmin.d_cmin;

    // Three forms of the pair stage, identical results (tests/test_gpu_parity.py runs all three against the oracle):
    //   join  (pairs_join.hip)   inverted index over all N*s hashes; N >= GHIP_JOIN_MIN_N, declines dense inputs
    //   probe (pairs_probe.hip)  dense, cuckoo sets in LDS; s <= 1024
    //   merge (pairs.hip)        dense, 64-way merge path; s > 1024, or a sketch holds 2^64-1 / a cuckoo insertion failed
    // GHIP_PAIR_KERNEL=join|probe|merge forces a form (join still declines what it cannot do).
    const uint32_t force = ctx->opt.pair_form;
    // (sketches too long for LDS tiles, s > 4096, go to the join whatever n is: the dense form left for them reads global memory)
    const bool want_join = !dense_share && (force != GHIP_PAIR_AUTO ? force == GHIP_PAIR_JOIN : (n >= GHIP_JOIN_MIN_N || s > 4096));
    bool use_probe = s <= 1024 && force != GHIP_PAIR_MERGE;
    bool probe_checked = false;

    const bool dbg_laps = ghip_dbg(ctx->opt, GHIP_DEBUG_PRECLUSTER);   // host laps of the stage on stderr
    auto lap_t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg_laps) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[precluster] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - lap_t0).count());
        lap_t0 = t;
    };
    const uint64_t P = (uint64_t)n * (n - 1) / 2;
    uint64_t cap = std::min<uint64_t>(P, std::max<uint64_t>(1u << 20, 64ull * n));
    unsigned long long *d_count = nullptr;
    if ((rc = dmalloc(ctx, &d_count, 1))) return rc;
    tmp.add(d_count);
    std::vector<ghip_pair> host;
    std::vector<uint32_t> empties;   // empty sketches of a joined run: their pairs are added on the host
    uint8_t *d_big = nullptr;        // genomes of element buckets too large for the join (their mutual pairs: dense, below)
    bool big_pending = false;
    bool filter_share = dense_share;  // dense pass over ALL pairs, this rank's (i + j) mod world share picked on the host
    bool listed = false;  // an attempt whose candidate list held every hit
    for (int attempt = 0; attempt < 4 && !listed; attempt++) {
        ghip_pair *d_out = nullptr;
        if ((rc = dmalloc(ctx, &d_out, cap))) return rc;
        DeviceFree t2(ctx); t2.add(d_out);
        GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
        uint64_t compared = 0;
        bool joined = false, late = false;
        empties.clear();
        bool has_big = false;
        if (want_join && !filter_share) {
            if (!d_big) { if ((rc = dmalloc(ctx, &d_big, n))) return rc; tmp.add(d_big); }
            GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_big, 0, n, ctx->stream));
            if ((rc = ghip_pairs_join(ctx, sk->d_hashes, sk->d_lens, n, s, d_cmin, ctx->cmin.floor,
                                      replicate_join ? 0 : rank, replicate_join ? 1 : world, (uint32_t)row_lo,
                                      d_out, d_count, cap, &compared, &joined, &late, &empties, d_big, &has_big))) return rc;
        }
        if (!joined) { empties.clear(); has_big = false; }   // a dense form enumerates every pair itself
        if (joined && replicate_join) {  // every rank holds every pair; book an equal share of the comparisons
            compared = P / world + (rank < P % world ? 1 : 0);
            if (out_replicated) *out_replicated = 1;
        }
        // The sharded join declined AFTER the point where all ranks decide alike (this rank's own records overflowed a
        // table): the other ranks deliver their (i + j) mod world shares, so this one must deliver exactly its own --
        // a dense pass over every pair, filtered on the host below.
        if (late && world > 1 && !replicate_join) filter_share = true;
        const uint32_t drank = filter_share ? 0 : rank, dworld = filter_share ? 1 : world;
        if (!joined && use_probe && !probe_checked) {
            if ((rc = prepare_probe(ctx, sk))) return rc;
            use_probe = sk->probe_flags == 0;
            probe_checked = true;
        }
        if (joined) {
        } else if (use_probe) {
            ghip_launch_pairs_probe(ctx, sk->d_hashes, sk->d_lens, sk->d_tables, sk->d_tags, n, s, sk->probe_cb, sk->d_row_start,
                                    (uint32_t)(sk->row_start.size() - 1), sk->n_work, d_cmin, drank, dworld, (uint32_t)row_lo, d_out, d_count, cap, sk->d_arranged, sk->probe_cbits);
            compared = ghip_probe_pairs_of_rank(n, sk->probe_cb, sk->row_start, drank, dworld);
        } else if (s <= 4096) {
            ghip_launch_pairs(ctx, sk->d_hashes, sk->d_lens, n, s, d_cmin, drank, dworld, (uint32_t)row_lo, d_out, d_count, cap, &compared);
        } else {
            ghip_launch_pairs_global(ctx, sk->d_hashes, sk->d_lens, n, s, d_cmin, drank, dworld, (uint32_t)row_lo, d_out, d_count, cap, &compared);
        }
        if (row_lo) compared = P - (uint64_t)row_lo * (row_lo - 1) / 2;   // the rectangle
        ctx->last_pairs = compared;
        unsigned long long cnt = 0;
        lap("kernels issued");
        if ((rc = d2h(ctx, &cnt, d_count, 1))) return rc;
        lap("count back (kernels done)");
        { hipError_t e = hipGetLastError(); if (e != hipSuccess) return ghip_set_error(ctx, GHIP_EHIP, std::string("pair_intersect_tile: ") + hipGetErrorString(e)); }
        if (cnt > cap) { cap = cnt; continue; }  // list overflowed: rerun with room for every hit
        host.resize(cnt);
        if ((rc = d2h(ctx, host.data(), d_out, cnt))) return rc;
        lap("candidates back");
        listed = true;
        big_pending = has_big;
        // the pairs of the empty sketches, which share no hash with anybody and pair with everybody (ANI 1.0 by the
        // reference's NaN arithmetic, common = total = 0): N - 1 each, in this rank's share of a sharded join
        if (!empties.empty()) {
            std::vector<uint8_t> is_empty(n, 0);
            for (uint32_t e : empties) is_empty[e] = 1;
            const uint32_t jrank = replicate_join ? 0 : rank, jworld = replicate_join ? 1 : world;
            for (uint32_t e : empties)
                for (size_t x = 0; x < n; x++) {
                    if (x == e || (is_empty[x] && x < e)) continue;   // two empty sketches: once
                    const uint32_t i = (uint32_t)std::min<size_t>(e, x), j = (uint32_t)std::max<size_t>(e, x);
                    if (j < row_lo) continue;
                    if (jworld > 1 && (i + j) % jworld != jrank) continue;
                    ghip_pair r; r.i = i; r.j = j; r.common = 0; r.total = 0; r.ani = 0.0f;
                    host.push_back(r);
                }
        }
    }
    if (!listed) return ghip_set_error(ctx, GHIP_EHIP, "precluster candidate list overflowed on every attempt");
    if (big_pending) {
        // The join left out the pairs of two genomes that both sit in an element bucket too large for it (a hash shared by
        // more than J_ELEM_CAP genomes: one very large family).  Those genomes' rows are gathered into a compact matrix, a
        // dense form runs over it, and the pairs come back under their own indices: the family costs |G|^2 / 2 probes, the
        // rest of the collection stays with the join (a dense pass over everything is 0.9 s at 50 000 genomes).
        std::vector<uint8_t> big(n);
        if ((rc = d2h(ctx, big.data(), d_big, n))) return rc;
        std::vector<uint32_t> G;
        for (size_t g = 0; g < n; g++) if (big[g]) G.push_back((uint32_t)g);
        if (G.size() >= 2) {
            ghip_sketches sub;
            sub.ctx = ctx; sub.n = G.size(); sub.s = s; sub.k = k; sub.owned = false;
            uint32_t *d_G = nullptr;
            if ((rc = dmalloc(ctx, &d_G, G.size())) || (tmp.add(d_G), false) || (rc = h2d(ctx, d_G, G.data(), G.size())) ||
                (rc = dmalloc(ctx, &sub.d_hashes, G.size() * (size_t)s)) || (tmp.add(sub.d_hashes), false) ||
                (rc = dmalloc(ctx, &sub.d_lens, G.size())) || (tmp.add(sub.d_lens), false)) return rc;
            ghip_launch_gather_rows(ctx, sk->d_hashes, sk->d_lens, d_G, G.size(), s, sub.d_hashes, sub.d_lens);
            bool sub_probe = s <= 1024 && force != GHIP_PAIR_MERGE;
            if (sub_probe) {
                if ((rc = prepare_probe(ctx, &sub))) return rc;
                tmp.add(sub.d_tables); tmp.add(sub.d_tags); tmp.add(sub.d_row_start); if (sub.d_arranged) tmp.add(sub.d_arranged);
                sub_probe = sub.probe_flags == 0;
            }
            const uint64_t Ps = (uint64_t)G.size() * (G.size() - 1) / 2;
            uint64_t scap = std::min<uint64_t>(Ps, std::max<uint64_t>(1u << 20, 64ull * G.size()));
            bool sub_listed = false;
            for (int attempt = 0; attempt < 4 && !sub_listed; attempt++) {
                ghip_pair *d_sub = nullptr;
                if ((rc = dmalloc(ctx, &d_sub, scap))) return rc;
                DeviceFree t3(ctx); t3.add(d_sub);
                GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
                uint64_t unused = 0;
                if (sub_probe)
                    ghip_launch_pairs_probe(ctx, sub.d_hashes, sub.d_lens, sub.d_tables, sub.d_tags, sub.n, s, sub.probe_cb, sub.d_row_start,
                                            (uint32_t)(sub.row_start.size() - 1), sub.n_work, d_cmin, 0, 1, 0, d_sub, d_count, scap, sub.d_arranged, sub.probe_cbits);
                else if (s <= 4096) ghip_launch_pairs(ctx, sub.d_hashes, sub.d_lens, sub.n, s, d_cmin, 0, 1, 0, d_sub, d_count, scap, &unused);
                else ghip_launch_pairs_global(ctx, sub.d_hashes, sub.d_lens, sub.n, s, d_cmin, 0, 1, 0, d_sub, d_count, scap, &unused);
                unsigned long long cnt = 0;
                if ((rc = d2h(ctx, &cnt, d_count, 1))) return rc;
                if (cnt > scap) { scap = cnt; continue; }
                std::vector<ghip_pair> part(cnt);
                if ((rc = d2h(ctx, part.data(), d_sub, cnt))) return rc;
                const uint32_t jrank = replicate_join ? 0 : rank, jworld = replicate_join ? 1 : world;
                for (ghip_pair &r : part) {
                    r.i = G[r.i]; r.j = G[r.j];   // (G ascends: i < j stays)
                    if (r.j < row_lo) continue;
                    if (jworld > 1 && (r.i + r.j) % jworld != jrank) continue;
                    host.push_back(r);
                }
                sub_listed = true;
            }
            if (!sub_listed) return ghip_set_error(ctx, GHIP_EHIP, "precluster candidate list overflowed on every attempt");
            lap("dense pass over the genomes of oversized buckets");
        }
    }
    rc = ghip_pairs_finalize(ctx, host, k, min_ani, n, filter_share, rank, world, out_pairs, out_n);
    lap("finalize (f64 recheck, (i, j) order)");
    return rc;
}

int ghip_precluster_dense_share(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, uint32_t rank, uint32_t world, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, rank, world, false, 0, out_pairs, out_n, nullptr, true);
}

extern "C" int ghip_precluster_shard(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, uint32_t rank,
                                     uint32_t world, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, rank, world, false, 0, out_pairs, out_n, nullptr);
}

extern "C" int ghip_precluster_ranks(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, uint32_t rank,
                                     uint32_t world, ghip_pair **out_pairs, size_t *out_n, int *out_replicated) {
    if (!out_replicated) return GHIP_EINVAL;
    // Default: the pair work is SHARDED -- dense forms by tile, the join form by (i + j) mod world at record emission
    // (its element stage, one pass over all N*s hashes, runs on every rank).  GHIP_JOIN_RANKS=replicate makes every
    // rank run the whole join and keep the whole list instead (no candidate exchange; DESIGN.md section 6 has both timings).
    const bool replicate = world > 1 && ctx->opt.join_ranks == GHIP_JOIN_REPLICATE;
    return precluster_impl(ctx, sk, min_ani, rank, world, replicate, 0, out_pairs, out_n, out_replicated);
}

extern "C" int ghip_precluster(ghip_ctx *ctx, const ghip_sketches *sk, float min_ani, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, 0, 1, false, 0, out_pairs, out_n, nullptr);
}

extern "C" int ghip_precluster_from(ghip_ctx *ctx, const ghip_sketches *sk, size_t row_lo, float min_ani, ghip_pair **out_pairs, size_t *out_n) {
    return precluster_impl(ctx, sk, min_ani, 0, 1, false, row_lo, out_pairs, out_n, nullptr);
}

