// Host clusterer: clusterer::cluster from the precluster cache onwards
// (reference src/clusterer.rs:56-152, 182-259, 350-487), on the sorted edge list the GPU
// pair stage returns instead of a BTreeMap probed O(N^2) times:
//   partition_sketches          -> union-find over the E edges
//   transform_ids               -> per-precluster adjacency lists
//   find_precluster_cluster_representatives / _memberships -> same decisions, same tie rules
// Result semantics follow the reference with --threads 1: preclusters in disjoint-set order
// (first element ascending) stable-sorted by size descending; within a precluster clusters in
// representative order; representative first, members ascending.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "ghip_internal.h"

namespace {

struct Adj {
    uint32_t nbr;   // local index of the neighbour
    uint32_t edge;  // index into the global pair list
};

struct Dsu {
    std::vector<uint32_t> p;
    explicit Dsu(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
    uint32_t find(uint32_t x) {
        while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
        return x;
    }
    void join(uint32_t a, uint32_t b) {
        a = find(a); b = find(b);
        if (a == b) return;
        if (a < b) p[b] = a; else p[a] = b;  // root = smallest member
    }
};

}  // namespace

static int cluster_impl(size_t n, const ghip_pair *pairs, size_t n_pairs, const float *pair_ani,
                        int skip_clusterer, float ani_threshold, ghip_ani_callback ani_cb, ghip_ani_batch_callback batch_cb,
                        void *user, uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters,
                        uint64_t *out_requested, ghip_ctx *pool_ctx = nullptr) {
    if (!out_members || !out_offsets || !out_n_clusters) return GHIP_EINVAL;
    if (n_pairs && !pairs) return GHIP_EINVAL;
    const ghip_options opt = pool_ctx ? pool_ctx->opt : ghip_process_options();
    const bool dbg = ghip_dbg(opt, GHIP_DEBUG_CLUSTER);
    auto tp0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[cluster] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tp0).count()); tp0 = t; } };
    if (out_requested) *out_requested = 0;
    // an ANI source is only needed when there is an edge to ask about: unrelated genomes (no precluster pair) are
    // singleton clusters whatever the clusterer (clusterer.rs:182-259 never calls calculate_ani then)
    if (n_pairs && !skip_clusterer && !pair_ani && !ani_cb && !batch_cb) return GHIP_EINVAL;
    for (size_t e = 0; e < n_pairs; e++)
        if (pairs[e].i >= n || pairs[e].j >= n || pairs[e].i == pairs[e].j) return GHIP_EINVAL;

    lap("validate");
    // ---- partition_sketches (clusterer.rs:452-487): single linkage over cache keys ----
    Dsu dsu(n);
    for (size_t e = 0; e < n_pairs; e++) dsu.join(pairs[e].i, pairs[e].j);
    // sets enumerated by first element; members ascending (clusterer.rs:67-76)
    std::vector<uint32_t> set_of(n), set_size;
    {
        std::vector<uint32_t> root_set(n, UINT32_MAX);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t r = dsu.find(i);
            if (root_set[r] == UINT32_MAX) { root_set[r] = (uint32_t)set_size.size(); set_size.push_back(0); }
            set_of[i] = root_set[r];
            set_size[set_of[i]]++;
        }
    }
    const size_t nsets = set_size.size();
    std::vector<uint64_t> set_start(nsets + 1, 0);
    for (size_t s = 0; s < nsets; s++) set_start[s + 1] = set_start[s] + set_size[s];
    std::vector<uint32_t> members(n), local(n);
    {
        std::vector<uint64_t> fill(set_start.begin(), set_start.end() - 1);
        for (uint32_t i = 0; i < n; i++) {
            uint64_t pos = fill[set_of[i]]++;
            members[pos] = i;
            local[i] = (uint32_t)(pos - set_start[set_of[i]]);
        }
    }
    // bigger preclusters first (clusterer.rs:79), stable
    std::vector<uint32_t> order(nsets);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return set_size[a] > set_size[b]; });

    lap("partition");
    // adjacency in CSR form over local indices (== transform_ids, cache.rs:47-58)
    std::vector<uint64_t> adj_start(n + 1, 0);
    for (size_t e = 0; e < n_pairs; e++) { adj_start[pairs[e].i + 1]++; adj_start[pairs[e].j + 1]++; }
    for (size_t i = 0; i < n; i++) adj_start[i + 1] += adj_start[i];
    std::vector<Adj> adj(2 * n_pairs);
    {
        std::vector<uint64_t> fill(adj_start.begin(), adj_start.end() - 1);
        for (size_t e = 0; e < n_pairs; e++) {
            adj[fill[pairs[e].i]++] = {local[pairs[e].j], (uint32_t)e};
            adj[fill[pairs[e].j]++] = {local[pairs[e].i], (uint32_t)e};
        }
    }
    lap("adjacency");
    // a duplicate key would be a BTreeMap overwrite; the GPU stage never emits one
    // clusterer ANI per edge: state 0 = not computed, 1 = None, 2 = Some(value)
    std::vector<uint8_t> ani_state(n_pairs, 0);
    std::vector<float> ani_val(n_pairs, 0.0f);
    if (skip_clusterer) {
        for (size_t e = 0; e < n_pairs; e++) { ani_state[e] = 2; ani_val[e] = pairs[e].ani; }
    } else if (pair_ani) {
        for (size_t e = 0; e < n_pairs; e++) {
            if (std::isnan(pair_ani[e])) ani_state[e] = 1;
            else { ani_state[e] = 2; ani_val[e] = pair_ani[e]; }
        }
    }
    std::atomic<bool> aborted{false};  // the callback returned < 0: the host's calculate_ani failed (a Rust panic, a Python exception)
    std::vector<uint8_t> is_rep(n, 0);       // indexed by genome
    if (batch_cb && !skip_clusterer && !pair_ani && n_pairs) {
        // ---- lazy ANI in batches.  The greedy rules only ever look at edges that touch a REPRESENTATIVE (candidate
        // representatives of a genome, clusterer.rs:194-204; representatives of a non-representative, :377-405), and which
        // genomes are representatives unfolds in genome order.  So: all preclusters advance in lock step; a genome that
        // no known representative covers becomes one and ALL its edges are requested; a precluster waits when its next
        // genome has an unanswered edge to a representative; one callback per round answers every request of every
        // precluster.  Rounds = (most representatives in one precluster) + 1; edges asked = those touching a
        // representative -- what the reference computes one `skani dist` at a time (without its stop-early luck),
        // typically a fifth to a half of all precluster pairs.
        std::vector<uint32_t> scan(nsets, 0), req;
        std::vector<uint8_t> requested(n_pairs, 0);
        std::vector<float> answers;
        // A round costs the callee one launch's latency however few edges it holds (ani_pairs: ~0.25 ms for anything up
        // to ~800 pairs), so a SHORT round is topped up (below): a short edge list is thereby asked for whole in its
        // first round, and the tail of a long one is one round instead of several tiny ones.
        const size_t flush_below = opt.lazy_flush_below;
        for (;;) {
            req.clear();
            for (size_t s = 0; s < nsets; s++) {
                const uint32_t *orig = members.data() + set_start[s];
                const uint32_t m = set_size[s];
                while (scan[s] < m) {
                    const uint32_t li = scan[s], gi = orig[li];
                    bool unknown = false, covered = false;
                    for (uint64_t a = adj_start[gi]; a < adj_start[gi + 1]; a++) {
                        const uint32_t lj = adj[a].nbr, e = adj[a].edge;
                        if (lj >= li || !is_rep[orig[lj]]) continue;
                        if (ani_state[e] == 0) unknown = true;
                        else if (ani_state[e] == 2 && ani_val[e] >= ani_threshold) covered = true;
                    }
                    if (unknown && !covered) break;   // wait for this round's answers
                    if (!covered) {
                        is_rep[gi] = 1;
                        for (uint64_t a = adj_start[gi]; a < adj_start[gi + 1]; a++) {
                            const uint32_t e = adj[a].edge;
                            if (ani_state[e] == 0 && !requested[e]) { requested[e] = 1; req.push_back(e); }
                        }
                    }
                    scan[s]++;
                }
            }
            if (req.empty()) break;
            // Once a round asks for fewer than `flush_below` edges -- a small input's first round, or the tail of a large one,
            // where only a few preclusters with long chains of representatives are still open (10 000 genomes: rounds of
            // 468, 85, 12 and 3 pairs after the first three) -- every further round would cost a launch's latency for a
            // handful of pairs.  Ask for everything the open preclusters still lack, once: they finish without another round.
            // (1 000 genomes, 4 500 edges: rounds of 900, 520 and 476 = 1.3 ms; everything in one round 1.6 ms.)
            if (req.size() < flush_below)
                for (size_t s = 0; s < nsets; s++) {
                    const uint32_t *orig = members.data() + set_start[s];
                    for (uint32_t li = scan[s]; li < set_size[s]; li++)
                        for (uint64_t a = adj_start[orig[li]]; a < adj_start[orig[li] + 1]; a++) {
                            const uint32_t e = adj[a].edge;
                            if (ani_state[e] == 0 && !requested[e]) { requested[e] = 1; req.push_back(e); }
                        }
                }
            answers.assign(req.size(), 0.0f);
            if (batch_cb(user, req.data(), req.size(), answers.data()) != 0) return GHIP_ECALLBACK;
            if (out_requested) *out_requested += req.size();
            for (size_t x = 0; x < req.size(); x++) {
                if (std::isnan(answers[x])) ani_state[req[x]] = 1;
                else { ani_state[req[x]] = 2; ani_val[req[x]] = answers[x]; }
            }
        }
        std::fill(is_rep.begin(), is_rep.end(), 0);   // the loops below re-derive it from the answers (same decisions)
    }
    auto edge_ani = [&](uint32_t e, uint32_t rep_genome, uint32_t genome) {
        if (ani_state[e] == 0 && !aborted && !ani_cb) { aborted = true; return; }   // (lazy form: cannot happen)
        if (ani_state[e] == 0 && !aborted) {  // ClusterDistanceFinder::calculate_ani(rep, genome)
            float v = 0.0f;
            int has = ani_cb(user, rep_genome, genome, &v);
            if (has < 0) { aborted = true; has = 0; }
            ani_state[e] = has ? 2 : 1;
            ani_val[e] = v;
        }
    };

    lap("ani states / lazy rounds");
    // The preclusters are independent (clusterer.rs:88 walks them with par_iter): ranges of `order` go to worker threads,
    // each writing its preclusters' members straight to their final place in out_m (a precluster's clusters occupy the
    // slots after those of the preclusters before it in `order`) and its cluster end offsets to a list of its own; the
    // lists are concatenated in order afterwards.  A callback back-end (ani_cb) stays on the calling thread: the host's
    // calculate_ani need not be re-entrant from threads it did not make.
    std::vector<uint32_t> out_m(n);
    std::vector<uint64_t> pre_base(nsets + 1, 0);   // first output slot of the oi-th precluster in `order`
    for (size_t oi = 0; oi < nsets; oi++) pre_base[oi + 1] = pre_base[oi] + set_size[order[oi]];
    std::vector<uint32_t> assign(n, 0);      // genome -> representative genome
    struct Cand { uint32_t nbr_local; uint32_t edge; float pre; };
    std::atomic<int> failure{GHIP_OK};

    auto run_range = [&](size_t o_lo, size_t o_hi, std::vector<uint64_t> &ends) {
    std::vector<Cand> cand;
    std::vector<uint32_t> rep_pos, cnt;
    std::vector<uint64_t> cstart, fill;
    for (size_t oi = o_lo; oi < o_hi && failure.load(std::memory_order_relaxed) == GHIP_OK; oi++) {
        const uint32_t s = order[oi];
        const uint32_t *orig = members.data() + set_start[s];
        const uint32_t m = set_size[s];
        if (m == 1) {   // a genome with no precluster pair: its own cluster (most of a diverse collection)
            is_rep[orig[0]] = 1; assign[orig[0]] = orig[0];
            out_m[pre_base[oi]] = orig[0];
            ends.push_back(pre_base[oi] + 1);
            continue;
        }
        // ---- find_precluster_cluster_representatives (clusterer.rs:182-259) ----
        for (uint32_t li = 0; li < m; li++) {
            const uint32_t gi = orig[li];
            cand.clear();
            for (uint64_t a = adj_start[gi]; a < adj_start[gi + 1]; a++) {
                const uint32_t lj = adj[a].nbr;
                if (lj < li && is_rep[orig[lj]]) cand.push_back({lj, adj[a].edge, pairs[adj[a].edge].ani});
            }
            // ascending by precluster ANI (clusterer.rs:200; sort_unstable -> ties by index here)
            if (cand.size() > 1)
                std::sort(cand.begin(), cand.end(), [](const Cand &x, const Cand &y) {
                    if (x.pre != y.pre) return x.pre < y.pre;
                    return x.nbr_local < y.nbr_local;
                });
            bool rep = true;
            for (const Cand &c : cand) {
                edge_ani(c.edge, orig[c.nbr_local], gi);
                if (ani_state[c.edge] == 2 && ani_val[c.edge] >= ani_threshold) {
                    rep = false;
                    if (!skip_clusterer && !pair_ani && !batch_cb) break;  // find_any stops on the first hit
                }
            }
            is_rep[gi] = rep ? 1 : 0;
            if (aborted) { failure = GHIP_ECALLBACK; return; }
        }
        // ---- find_precluster_cluster_memberships (clusterer.rs:350-449) ----
        for (uint32_t li = 0; li < m; li++) {
            const uint32_t gi = orig[li];
            if (is_rep[gi]) { assign[gi] = gi; continue; }
            bool have = false;
            float best = 0.0f;
            uint32_t best_local = 0;
            for (uint64_t a = adj_start[gi]; a < adj_start[gi + 1]; a++) {
                const uint32_t lj = adj[a].nbr;
                if (!is_rep[orig[lj]]) continue;
                edge_ani(adj[a].edge, orig[lj], gi);
                if (ani_state[adj[a].edge] != 2) continue;
                const float v = ani_val[adj[a].edge];
                // reps are visited in ascending order by the reference; strictly-greater wins,
                // so among equal ANIs the lowest-index representative is kept.
                if (!have || v > best || (v == best && lj < best_local)) { have = true; best = v; best_local = lj; }
            }
            if (aborted) { failure = GHIP_ECALLBACK; return; }
            if (!have) { failure = GHIP_EINVAL; return; }  // reference: best_rep.unwrap() panics (clusterer.rs:444)
            assign[gi] = orig[best_local];
        }
        // clusters of this precluster: representative first, then members ascending
        const size_t base = pre_base[oi];
        rep_pos.assign(m, UINT32_MAX);
        uint32_t nreps = 0;
        for (uint32_t li = 0; li < m; li++) if (is_rep[orig[li]]) { rep_pos[li] = nreps++; }
        cnt.assign(nreps, 0);
        for (uint32_t li = 0; li < m; li++) cnt[rep_pos[local[assign[orig[li]]]]]++;
        cstart.assign(nreps + 1, 0);
        for (uint32_t r = 0; r < nreps; r++) cstart[r + 1] = cstart[r] + cnt[r];
        fill.assign(cstart.begin(), cstart.end() - 1);
        for (uint32_t li = 0; li < m; li++)  // representatives first
            if (is_rep[orig[li]]) out_m[base + fill[rep_pos[li]]++] = orig[li];
        for (uint32_t li = 0; li < m; li++)
            if (!is_rep[orig[li]]) out_m[base + fill[rep_pos[local[assign[orig[li]]]]]++] = orig[li];
        for (uint32_t r = 0; r < nreps; r++) ends.push_back(base + cstart[r + 1]);
    }
    };
    std::vector<uint64_t> out_o{0};
    {
        // threads only where they pay (spawning one costs ~30 us): ranges of >= 2 000 genomes, at most 8 workers
        size_t workers = ani_cb ? 1 : std::min<size_t>({(size_t)8, n / 2000, (size_t)std::max(1u, std::thread::hardware_concurrency())});
        // (a callback back-end stays on ONE thread whatever the option says: the host's calculate_ani need not be re-entrant)
        if (opt.cluster_threads && !ani_cb) workers = opt.cluster_threads;
        workers = std::max<size_t>(1, std::min(workers, nsets));
        std::vector<std::vector<uint64_t>> ends(workers);
        if (workers == 1) run_range(0, nsets, ends[0]);
        else {
            // equal shares of the GENOMES (the big preclusters come first in `order`)
            std::vector<size_t> cut(workers + 1, nsets);
            cut[0] = 0;
            for (size_t w = 1; w < workers; w++)
                cut[w] = std::lower_bound(pre_base.begin(), pre_base.end(), (uint64_t)(n * w / workers)) - pre_base.begin();
            for (size_t w = 1; w < workers; w++) cut[w] = std::min(std::max(cut[w], cut[w - 1]), nsets);
            // the context's persistent I/O workers when the caller has a context and no ingest holds them (a wake-up is ~10 us,
            // a fresh thread ~30 us to spawn and as much to join); fresh threads otherwise
            if (pool_ctx && pool_ctx->ingest_mu.try_lock()) {
                pool_ctx->io.run((int)workers, [&](int w) { run_range(cut[w], cut[w + 1], ends[w]); });
                pool_ctx->ingest_mu.unlock();
            } else {
                std::vector<std::thread> pool;
                for (size_t w = 1; w < workers; w++) pool.emplace_back([&, w] { run_range(cut[w], cut[w + 1], ends[w]); });
                run_range(cut[0], cut[1], ends[0]);
                for (auto &t : pool) t.join();
            }
        }
        if (failure.load() != GHIP_OK) return failure.load();
        size_t total = 1;
        for (auto &e : ends) total += e.size();
        out_o.reserve(total);
        for (auto &e : ends) out_o.insert(out_o.end(), e.begin(), e.end());
    }

    lap("preclusters");
    uint32_t *om = (uint32_t *)malloc(std::max<size_t>(out_m.size(), 1) * sizeof(uint32_t));
    uint64_t *oo = (uint64_t *)malloc(out_o.size() * sizeof(uint64_t));
    if (!om || !oo) { free(om); free(oo); return GHIP_ENOMEM; }
    if (!out_m.empty()) memcpy(om, out_m.data(), out_m.size() * sizeof(uint32_t));   // (n = 0: data() may be null, which memcpy may not be given)
    memcpy(oo, out_o.data(), out_o.size() * sizeof(uint64_t));
    *out_members = om;
    *out_offsets = oo;
    *out_n_clusters = out_o.size() - 1;
    return GHIP_OK;
}

extern "C" int ghip_cluster(size_t n, const ghip_pair *pairs, size_t n_pairs, const float *pair_ani,
                            int skip_clusterer, float ani_threshold, ghip_ani_callback ani_cb, void *user,
                            uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters) {
    return cluster_impl(n, pairs, n_pairs, pair_ani, skip_clusterer, ani_threshold, ani_cb, nullptr, user, out_members, out_offsets,
                        out_n_clusters, nullptr);
}

extern "C" int ghip_cluster_lazy(size_t n, const ghip_pair *pairs, size_t n_pairs, float ani_threshold,
                                 ghip_ani_batch_callback batch_cb, void *user, uint32_t **out_members, uint64_t **out_offsets,
                                 size_t *out_n_clusters, uint64_t *out_pairs_requested) {
    if (n_pairs && !batch_cb) return GHIP_EINVAL;
    return cluster_impl(n, pairs, n_pairs, nullptr, 0, ani_threshold, nullptr, batch_cb, user, out_members, out_offsets, out_n_clusters,
                        out_pairs_requested);
}

// ---- clusterer::cluster with the device ANI index as the ClusterDistanceFinder, whole in native code ----
// (src/clusterer.rs:56-152 with SkaniClusterer, src/skani.rs:718-788): the lazy rounds of ghip_cluster_lazy answered by
// ghip_ani_pairs without a trip through the host language per round.  `order` (nullable) is galah's quality order
// (src/cluster_argument_parsing.rs:863-1157 sorts the genomes before anything else): order[x] = the genome that comes x-th.
// The sketches, the pair list and the ANI index stay where the genomes lie; the pair list is renumbered to positions and
// re-sorted (LSD radix over the bits two positions need), the ANI of an edge is asked for by its genomes.
namespace {
struct IndexAni {
    ghip_ctx *ctx;
    const ghip_ani_index *idx;
    const ghip_pair *by_genome;        // the caller's pair list (genome indices)
    const uint32_t *orig;              // sorted edge -> index into by_genome (null: identity)
    float min_af;
    // several ranks (ghip_cluster_index_comm): a request is answered by the rank that owns the pair's FIRST genome (where
    // ghip_exchange_ani_index put the slices it needs); local_ids maps genome -> position in this rank's index
    // a host's own ANI source instead of the device index (ghip_cluster_lazy_comm): called with the edges THIS rank answers,
    // as indices into the caller's pair list
    ghip_ani_batch_callback host_cb = nullptr;
    void *host_user = nullptr;
    std::vector<uint32_t> mine_edge;
    ghip_comm *comm = nullptr;
    const uint32_t *local_ids = nullptr;
    size_t block = 0;
    uint32_t rank = 0, world = 1;
    std::vector<uint32_t> buf, mine_at;
    std::vector<float> vals;
    std::vector<uint8_t> send, got;
    std::vector<uint64_t> sizes, cursor;
    uint64_t rounds = 0, ns_ani = 0, asked_here = 0;
    int rc = GHIP_OK;
};
int index_ani_batch(void *user, const uint32_t *edge, size_t n, float *out) {
    IndexAni *s = static_cast<IndexAni *>(user);
    const auto t0 = std::chrono::steady_clock::now();
    auto pair_of = [&](size_t x) -> const ghip_pair & { return s->by_genome[s->orig ? s->orig[edge[x]] : edge[x]]; };
    auto id = [&](uint32_t g) { return s->local_ids ? s->local_ids[g] : g; };
    auto edge_of = [&](size_t x) { return s->orig ? s->orig[edge[x]] : edge[x]; };   // index into the caller's pair list
    if (s->world == 1) {
        if (s->host_cb) {
            s->mine_edge.resize(n);
            for (size_t x = 0; x < n; x++) s->mine_edge[x] = edge_of(x);
            s->rc = s->host_cb(s->host_user, s->mine_edge.data(), n, out) == 0 ? GHIP_OK : GHIP_ECALLBACK;
        } else {
            s->buf.resize(2 * n);
            for (size_t x = 0; x < n; x++) { const ghip_pair &p = pair_of(x); s->buf[2 * x] = id(p.i); s->buf[2 * x + 1] = id(p.j); }
            s->rc = ghip_ani_pairs(s->ctx, s->idx, s->buf.data(), n, s->min_af, out, nullptr);
        }
        s->asked_here += n;
    } else {
        // The round's requests are the same list on every rank (the clusterer is deterministic and every rank runs it):
        // each rank answers the requests it owns, one variable-length gather -- sizes known to all, so ONE collective --
        // carries every rank's answers behind a status word, and every rank files them in request order.  A rank whose
        // launch failed says so in that word: all ranks leave the round together.
        s->buf.clear(); s->mine_at.clear(); s->mine_edge.clear();
        s->sizes.assign(s->world, sizeof(uint32_t));
        for (size_t x = 0; x < n; x++) {
            const ghip_pair &p = pair_of(x);
            const size_t owner = p.i / s->block;
            s->sizes[owner] += sizeof(float);
            if (owner == s->rank) {
                if (s->host_cb) s->mine_edge.push_back(edge_of(x));
                else { s->buf.push_back(id(p.i)); s->buf.push_back(id(p.j)); }
                s->mine_at.push_back((uint32_t)x);
            }
        }
        const size_t mine = s->mine_at.size();
        s->vals.assign(mine, 0.0f);
        int local_rc = GHIP_OK;
        for (size_t x = 0; x < s->buf.size() && !local_rc; x++)
            if (s->buf[x] == UINT32_MAX) local_rc = ghip_set_error(s->ctx, GHIP_EINVAL, "a requested pair's genome is not in this rank's ANI index");
        if (!local_rc && mine) {
            if (s->host_cb) { if (s->host_cb(s->host_user, s->mine_edge.data(), mine, s->vals.data()) != 0) local_rc = ghip_set_error(s->ctx, GHIP_ECALLBACK, "the host's batched calculate_ani failed on this rank"); }
            else local_rc = ghip_ani_pairs(s->ctx, s->idx, s->buf.data(), mine, s->min_af, s->vals.data(), nullptr);
        }
        if (!local_rc && ghip_comm_fault(s->comm, GHIP_FAULT_ANI_ROUND)) local_rc = ghip_set_error(s->ctx, GHIP_EHIP, "injected fault: ANI round");
        s->asked_here += mine;
        s->send.resize(sizeof(uint32_t) + mine * sizeof(float));
        const uint32_t status = (uint32_t)local_rc;
        memcpy(s->send.data(), &status, sizeof(status));
        if (mine) memcpy(s->send.data() + sizeof(uint32_t), s->vals.data(), mine * sizeof(float));
        int rc = ghip_comm_gatherv_known(s->comm, s->send.data(), s->send.size(), s->sizes, s->got);
        if (!rc) {
            s->cursor.assign(s->world, 0);
            uint64_t at = 0;
            for (uint32_t r = 0; r < s->world && !rc; r++) {
                uint32_t st;
                memcpy(&st, s->got.data() + at, sizeof(st));
                if (st) rc = r == s->rank ? local_rc : ghip_set_error(s->ctx, GHIP_EPEER, "rank " + std::to_string(r) + " failed in a lazy ANI round (code " + std::to_string(st) + ")");
                s->cursor[r] = at + sizeof(uint32_t);
                at += s->sizes[r];
            }
            if (!rc)
                for (size_t x = 0; x < n; x++) {
                    const size_t owner = pair_of(x).i / s->block;
                    memcpy(&out[x], s->got.data() + s->cursor[owner], sizeof(float));
                    s->cursor[owner] += sizeof(float);
                }
        }
        s->rc = rc ? rc : local_rc;
    }
    s->rounds++;
    s->ns_ani += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return s->rc == GHIP_OK ? 0 : 1;
}

int cluster_index_impl(ghip_ctx *ctx, ghip_comm *comm, const ghip_ani_index *idx, const uint32_t *local_ids, size_t n, const ghip_pair *pairs,
                       size_t n_pairs, const uint32_t *order, float ani_threshold, float min_aligned_fraction, uint32_t **out_members,
                       uint64_t **out_offsets, size_t *out_n_clusters, uint64_t *out_stats, size_t n_stats,
                       ghip_ani_batch_callback host_cb = nullptr, void *host_user = nullptr) {
    // (ctx may be NULL with a host callback: a host-payload communicator has no device)
    if ((!ctx && !host_cb) || !out_members || !out_offsets || !out_n_clusters || (n_pairs && (!pairs || (!idx && !host_cb)))) return GHIP_EINVAL;
    const auto t0 = std::chrono::steady_clock::now();
    if (out_stats) memset(out_stats, 0, n_stats * sizeof(uint64_t));
    int rc_in = GHIP_OK;
    for (size_t e = 0; e < n_pairs && !rc_in; e++)
        if (pairs[e].i >= n || pairs[e].j >= n || pairs[e].i == pairs[e].j) rc_in = ghip_set_error(ctx, GHIP_EINVAL, "pair list names a genome out of range");
    if (order && n_pairs && !rc_in) {
        std::vector<uint8_t> seen(n, 0);
        for (size_t x = 0; x < n && !rc_in; x++) {
            if (order[x] >= n || seen[order[x]]) rc_in = ghip_set_error(ctx, GHIP_EINVAL, "order is not a permutation of the genomes");
            else seen[order[x]] = 1;
        }
    }
    // several ranks: the rounds below are collectives whose sizes follow from lazy_flush_below and from the arguments -- the
    // ranks meet first, with their settings and with what each made of its arguments (ghip_comm_agree: GHIP_EINVAL on every
    // rank when the settings differ, GHIP_EPEER on the peers of a rank that rejects its arguments)
    if (comm && ghip_comm_world(comm) > 1) rc_in = ghip_comm_agree(comm, rc_in);
    if (rc_in) return rc_in;
    std::vector<ghip_pair> sorted;
    std::vector<uint32_t> orig;
    if (order && n_pairs) {
        std::vector<uint32_t> rank_of(n, UINT32_MAX);
        for (size_t x = 0; x < n; x++) {
            rank_of[order[x]] = (uint32_t)x;   // (a permutation: checked above)
        }
        uint32_t bits = 1;
        while (((uint64_t)1 << bits) < n) bits++;
        std::vector<uint64_t> key(n_pairs), key2(n_pairs);
        std::vector<uint32_t> ix(n_pairs), ix2(n_pairs);
        for (size_t e = 0; e < n_pairs; e++) {
            const uint32_t a = rank_of[pairs[e].i], b = rank_of[pairs[e].j];
            key[e] = ((uint64_t)std::min(a, b) << bits) | std::max(a, b);
            ix[e] = (uint32_t)e;
        }
        constexpr uint32_t DIGIT = 11;
        std::vector<uint32_t> hist((size_t)1 << DIGIT);
        for (uint32_t sh = 0; sh < 2 * bits; sh += DIGIT) {
            std::fill(hist.begin(), hist.end(), 0u);
            for (size_t e = 0; e < n_pairs; e++) hist[(key[e] >> sh) & ((1u << DIGIT) - 1)]++;
            uint32_t run = 0;
            for (uint32_t &h : hist) { const uint32_t c = h; h = run; run += c; }
            for (size_t e = 0; e < n_pairs; e++) {
                const uint32_t p = hist[(key[e] >> sh) & ((1u << DIGIT) - 1)]++;
                key2[p] = key[e]; ix2[p] = ix[e];
            }
            key.swap(key2); ix.swap(ix2);
        }
        sorted.resize(n_pairs);
        for (size_t e = 0; e < n_pairs; e++) {
            sorted[e] = pairs[ix[e]];   // common / total / ani are symmetric in the pair
            sorted[e].i = (uint32_t)(key[e] >> bits);
            sorted[e].j = (uint32_t)(key[e] & (((uint64_t)1 << bits) - 1));
        }
        orig.swap(ix);
    }
    IndexAni st{ctx, idx, pairs, orig.empty() ? nullptr : orig.data(), min_aligned_fraction};
    if (comm) {
        size_t first, count;
        st.comm = comm; st.rank = ghip_comm_rank(comm); st.world = ghip_comm_world(comm);
        ghip_shard_range(n, st.rank, st.world, &first, &count, &st.block);
    }
    st.local_ids = local_ids;
    st.host_cb = host_cb; st.host_user = host_user;
    uint64_t asked = 0;
    const int rc = cluster_impl(n, sorted.empty() ? pairs : sorted.data(), n_pairs, nullptr, 0, ani_threshold, nullptr,
                                n_pairs ? index_ani_batch : nullptr, &st, out_members, out_offsets, out_n_clusters, &asked, ctx);
    if (out_stats) {
        out_stats[0] = asked; out_stats[1] = st.rounds; out_stats[2] = st.ns_ani;
        out_stats[3] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (n_stats > 4) out_stats[4] = st.asked_here;
    }
    if (rc == GHIP_ECALLBACK && st.rc != GHIP_OK) return st.rc;   // the ANI launch failed (here or on a peer): its own code and message
    return rc;
}
}  // namespace

extern "C" int ghip_cluster_index(ghip_ctx *ctx, const ghip_ani_index *idx, size_t n, const ghip_pair *pairs, size_t n_pairs,
                                  const uint32_t *order, float ani_threshold, float min_aligned_fraction,
                                  uint32_t **out_members, uint64_t **out_offsets, size_t *out_n_clusters, uint64_t *out_stats) {
    return cluster_index_impl(ctx, nullptr, idx, nullptr, n, pairs, n_pairs, order, ani_threshold, min_aligned_fraction, out_members, out_offsets,
                              out_n_clusters, out_stats, 4);
}

// The same on several ranks (src/clusterer.rs:194-204, 276-296, 377-405: the reference asks lazily, and so does every world
// size here): EVERY rank calls it with the same pair list and runs the same deterministic clusterer; a round's requests are
// dealt to the ranks that own the pairs' first genomes (`idx` = what ghip_exchange_ani_index returned on this rank,
// `local_ids` its genome -> position map) and come back in one variable-length gather per round.  Every rank returns the
// same clusters.  out_stats[5] = pairs asked (all ranks), rounds, ns in the rounds, ns in all, pairs THIS rank computed.
extern "C" int ghip_cluster_index_comm(ghip_comm *comm, const ghip_ani_index *idx, const uint32_t *local_ids, size_t n,
                                       const ghip_pair *pairs, size_t n_pairs, const uint32_t *order, float ani_threshold,
                                       float min_aligned_fraction, uint32_t **out_members, uint64_t **out_offsets,
                                       size_t *out_n_clusters, uint64_t *out_stats) {
    if (!comm) return GHIP_EINVAL;
    ghip_ctx *ctx = ghip_comm_context(comm);
    if (!ctx) return GHIP_EINVAL;
    return ghip_comm_note_error(comm, cluster_index_impl(ctx, comm, idx, local_ids, n, pairs, n_pairs, order, ani_threshold, min_aligned_fraction,
                                                         out_members, out_offsets, out_n_clusters, out_stats, 5));
}

// The same rounds with the HOST's ClusterDistanceFinder answering them (the reference's calculate_ani, src/lib.rs:47-55, as the
// batched callback of ghip_cluster_lazy): every rank calls it with the same pair list and its own callback, which is handed
// the edges THIS rank answers -- those whose first genome lies in its block of ceil(n / world) genomes -- as indices into
// `pairs`; one variable-length gather per round; every rank returns the same clusters.  Needs no device: a communicator
// made by ghip_comm_init_callback(NULL, ...) will do (what tests/test_distributed_cpu.py drives over gloo).
extern "C" int ghip_cluster_lazy_comm(ghip_comm *comm, size_t n, const ghip_pair *pairs, size_t n_pairs, const uint32_t *order,
                                      float ani_threshold, ghip_ani_batch_callback batch_cb, void *user, uint32_t **out_members,
                                      uint64_t **out_offsets, size_t *out_n_clusters, uint64_t *out_stats) {
    if (!comm || (n_pairs && !batch_cb)) return GHIP_EINVAL;
    return ghip_comm_note_error(comm, cluster_index_impl(ghip_comm_context(comm), comm, nullptr, nullptr, n, pairs, n_pairs, order, ani_threshold, 0.0f,
                                                         out_members, out_offsets, out_n_clusters, out_stats, 5, batch_cb, user));
}
