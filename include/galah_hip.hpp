// galah_hip.hpp -- C++ host side above the C ABI (include/galah_hip.h): the reference's plug-in interface for
// the finch-precluster + ANI path, restated with the same names, argument meaning and error behaviour.
//
// galah is compiled Rust; this image has no Rust toolchain, so the host mirror is C++ (header-only, C++17) and the
// Rust shim a maintainer adds is in INTEGRATION.md.  Rust `panic!` / `expect` become std::runtime_error carrying the
// reference's message.
//
//   reference                                             here
//   ---------------------------------------------------   -----------------------------------------------
//   SortedPairGenomeDistanceCache                          galah::SortedPairGenomeDistanceCache
//     (src/sorted_pair_genome_distance_cache.rs:5-59)
//   trait PreclusterDistanceFinder (src/lib.rs:29-45)      galah::PreclusterDistanceFinder
//   trait ClusterDistanceFinder    (src/lib.rs:47-55)      galah::ClusterDistanceFinder
//   FinchPreclusterer (src/finch.rs:4-46), finch::         galah::FinchPreclusterer, galah::finch::distances
//     distances (src/finch.rs:48-97)
//   SkaniClusterer (src/skani.rs:689-716)                  galah::HipAniClusterer ("hipani": build-defined ANI)
//   clusterer::cluster (src/clusterer.rs:14-152)           galah::cluster
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "galah_hip.h"

namespace galah {

// ---------------------------------------------------------------------------------------------------------------
class SortedPairGenomeDistanceCache {  // src/sorted_pair_genome_distance_cache.rs
public:
    using Key = std::pair<size_t, size_t>;
    std::map<Key, std::optional<float>> internal;  // BTreeMap<(usize, usize), Option<f32>>

    void insert(Key pair, std::optional<float> distance) { internal[sorted(pair)] = distance; }          // :22-29
    // Some(&Option<f32>) / None: outer optional = key present
    std::optional<std::optional<float>> get(Key pair) const {                                              // :31-37
        auto it = internal.find(sorted(pair));
        if (it == internal.end()) return std::nullopt;
        return it->second;
    }
    bool contains_key(Key pair) const { return internal.count(sorted(pair)) != 0; }                        // :39-45
    // keep pairs whose both members are in `ids`, renumbered by their position in `ids`                   // :47-58
    SortedPairGenomeDistanceCache transform_ids(const std::vector<size_t> &ids) const {
        SortedPairGenomeDistanceCache out;
        for (size_t i = 0; i < ids.size(); i++)
            for (size_t j = i + 1; j < ids.size(); j++) {
                auto v = get({ids[i], ids[j]});
                if (v) out.insert({i, j}, *v);
            }
        return out;
    }
    size_t len() const { return internal.size(); }
    bool operator==(const SortedPairGenomeDistanceCache &o) const { return internal == o.internal; }

private:
    static Key sorted(Key p) { return p.first < p.second ? p : Key{p.second, p.first}; }
};

// ---------------------------------------------------------------------------------------------------------------
struct PreclusterDistanceFinder {  // src/lib.rs:29-45
    virtual ~PreclusterDistanceFinder() = default;
    virtual SortedPairGenomeDistanceCache distances(const std::vector<std::string> &genome_fasta_paths) = 0;
    virtual SortedPairGenomeDistanceCache distances_contigs(const std::vector<std::string> &genome_fasta_paths,
                                                            const std::vector<std::string> &contig_names) = 0;
    virtual SortedPairGenomeDistanceCache distances_with_references(const std::vector<std::string> &genome_fasta_paths,
                                                                    const std::vector<std::string> &reference_genomes) = 0;
    virtual std::string method_name() const = 0;
};

struct ClusterDistanceFinder {  // src/lib.rs:47-55
    virtual ~ClusterDistanceFinder() = default;
    virtual void initialise() const = 0;
    virtual std::string method_name() const = 0;
    virtual float get_ani_threshold() const = 0;
    virtual std::optional<float> calculate_ani(const std::string &fasta1, const std::string &fasta2) = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// One HIP context shared by the back-ends of a run (ghip_init / ghip_destroy).
class HipContext {
public:
    explicit HipContext(int device = 0) {
        if (ghip_init(device, &ctx_) != GHIP_OK)
            throw std::runtime_error(std::string("Failed to initialise the HIP back-end: ") + ghip_last_error(nullptr));
    }
    ~HipContext() { ghip_destroy(ctx_); }
    HipContext(const HipContext &) = delete;
    HipContext &operator=(const HipContext &) = delete;
    ghip_ctx *get() const { return ctx_; }
    std::string last_error() const { return ghip_last_error(ctx_); }
    // ghip_options of this context (every switch of the library; the GHIP_* environment only seeds the defaults, once)
    ghip_options options() const {
        ghip_options o{};
        ghip_get_options(ctx_, &o);
        return o;
    }
    void set_options(ghip_options o) {
        o.struct_size = sizeof(ghip_options);
        if (ghip_set_options(ctx_, &o) != GHIP_OK) throw std::runtime_error("ghip_set_options: " + last_error());
    }

private:
    ghip_ctx *ctx_ = nullptr;
};

namespace detail {
inline std::vector<const char *> c_paths(const std::vector<std::string> &paths) {
    std::vector<const char *> out;
    out.reserve(paths.size());
    for (auto &p : paths) out.push_back(p.c_str());
    return out;
}
}  // namespace detail

// ---------------------------------------------------------------------------------------------------------------
namespace finch {
// finch::distances (src/finch.rs:48-97): sketch every file (k-mers never span records, seed 0, no filtering,
// no_strict), then every pair i<j whose Mash ANI >= min_ani (compared in f64) enters the cache as `ANI as f32`.
// `edges` (optional) receives the same entries as the sorted edge list ghip_cluster consumes.
inline SortedPairGenomeDistanceCache distances(HipContext &hip, const std::vector<std::string> &genome_fasta_paths,
                                               float min_ani, size_t num_kmers, uint8_t kmer_length, int threads = 1,
                                               std::vector<ghip_pair> *edges = nullptr) {
    auto paths = detail::c_paths(genome_fasta_paths);
    ghip_sketches *sk = nullptr;
    if (ghip_sketch_files(hip.get(), paths.data(), paths.size(), kmer_length, (uint32_t)num_kmers, 0, threads, &sk) != GHIP_OK)
        throw std::runtime_error("Failed to sketch genomes with finch: " + hip.last_error());        // finch.rs:72
    ghip_pair *pairs = nullptr;
    size_t n = 0;
    const int rc = ghip_precluster(hip.get(), sk, min_ani, &pairs, &n);
    ghip_sketches_free(sk);
    if (rc != GHIP_OK) throw std::runtime_error("Failed to compare finch sketches: " + hip.last_error());  // finch.rs:80-85
    SortedPairGenomeDistanceCache cache;
    for (size_t x = 0; x < n; x++) cache.insert({pairs[x].i, pairs[x].j}, pairs[x].ani);              // finch.rs:91-93
    if (edges) edges->assign(pairs, pairs + n);
    ghip_free(pairs);
    return cache;
}
// Incremental dereplication on a persisted sketch matrix (include/galah_hip.h "Persisted sketch matrix"; no counterpart
// in the reference's finch back-end -- docs/preludes/cluster_prelude.md:13-15 is the workflow it serves).
//   save: distances() whose sketch matrix and genome names are also written to `matrix_path`
//   incremental: genome list = [the saved genomes ..., the new files ...]; only the NEW files are read and sketched, the
//   pair stage runs on the (new x all) rectangle; with `saved_edges` (the earlier run's edge list) `edges` / the cache
//   are what distances() over all the files would produce.  `names` receives the genome list.
inline SortedPairGenomeDistanceCache distances_and_save(HipContext &hip, const std::vector<std::string> &genome_fasta_paths, float min_ani,
                                                        size_t num_kmers, uint8_t kmer_length, const std::string &matrix_path, int threads = 1,
                                                        std::vector<ghip_pair> *edges = nullptr) {
    auto paths = detail::c_paths(genome_fasta_paths);
    ghip_sketches *sk = nullptr;
    if (ghip_sketch_files(hip.get(), paths.data(), paths.size(), kmer_length, (uint32_t)num_kmers, 0, threads, &sk) != GHIP_OK)
        throw std::runtime_error("Failed to sketch genomes with finch: " + hip.last_error());
    ghip_pair *pairs = nullptr;
    size_t n = 0;
    int rc = ghip_sketches_save_named(hip.get(), sk, paths.data(), 0, matrix_path.c_str());
    if (rc == GHIP_OK) rc = ghip_precluster(hip.get(), sk, min_ani, &pairs, &n);
    ghip_sketches_free(sk);
    if (rc != GHIP_OK) throw std::runtime_error("Failed to compare / save finch sketches: " + hip.last_error());
    SortedPairGenomeDistanceCache cache;
    for (size_t x = 0; x < n; x++) cache.insert({pairs[x].i, pairs[x].j}, pairs[x].ani);
    if (edges) edges->assign(pairs, pairs + n);
    ghip_free(pairs);
    return cache;
}

inline SortedPairGenomeDistanceCache distances_incremental(HipContext &hip, const std::string &matrix_path,
                                                           const std::vector<std::string> &new_genome_fasta_paths, float min_ani,
                                                           size_t num_kmers, uint8_t kmer_length, const std::vector<ghip_pair> *saved_edges,
                                                           std::vector<std::string> *names, int threads = 1,
                                                           std::vector<ghip_pair> *edges = nullptr) {
    ghip_sketches *saved = nullptr, *fresh = nullptr, *both = nullptr;
    char *blob = nullptr;
    size_t blob_bytes = 0;
    uint64_t seed = 0;
    if (ghip_sketches_load_named(hip.get(), matrix_path.c_str(), &saved, &blob, &blob_bytes, &seed) != GHIP_OK)
        throw std::runtime_error("Failed to load the sketch matrix: " + hip.last_error());
    const size_t n_old = ghip_sketches_count(saved);
    std::vector<std::string> all;
    for (size_t at = 0, g = 0; g < n_old && at < blob_bytes; g++) { all.emplace_back(blob + at); at += all.back().size() + 1; }
    ghip_free(blob);
    if (ghip_sketches_kmer(saved) != kmer_length || ghip_sketches_size(saved) != num_kmers || seed != 0) {
        ghip_sketches_free(saved);
        throw std::runtime_error("sketch matrix " + matrix_path + " was made with other sketch parameters");
    }
    auto paths = detail::c_paths(new_genome_fasta_paths);
    int rc = ghip_sketch_files(hip.get(), paths.data(), paths.size(), kmer_length, (uint32_t)num_kmers, 0, threads, &fresh);
    if (rc != GHIP_OK) { ghip_sketches_free(saved); throw std::runtime_error("Failed to sketch genomes with finch: " + hip.last_error()); }
    rc = ghip_sketches_concat(hip.get(), saved, fresh, &both);
    ghip_sketches_free(saved);
    ghip_sketches_free(fresh);
    ghip_pair *pairs = nullptr;
    size_t n = 0;
    if (rc == GHIP_OK) rc = ghip_precluster_from(hip.get(), both, n_old, min_ani, &pairs, &n);
    ghip_sketches_free(both);
    if (rc != GHIP_OK) throw std::runtime_error("Failed to compare finch sketches: " + hip.last_error());
    std::vector<ghip_pair> merged;
    if (saved_edges) merged = *saved_edges;
    merged.insert(merged.end(), pairs, pairs + n);
    ghip_free(pairs);
    std::sort(merged.begin(), merged.end(), [](const ghip_pair &a, const ghip_pair &b) { return a.i != b.i ? a.i < b.i : a.j < b.j; });
    SortedPairGenomeDistanceCache cache;
    for (const ghip_pair &p : merged) cache.insert({p.i, p.j}, p.ani);
    all.insert(all.end(), new_genome_fasta_paths.begin(), new_genome_fasta_paths.end());
    if (names) *names = all;
    if (edges) *edges = merged;
    return cache;
}
}  // namespace finch

class FinchPreclusterer : public PreclusterDistanceFinder {  // src/finch.rs:4-46
public:
    float min_ani;         // a fraction, not a percentage (finch.rs:5-6)
    size_t num_kmers;      // 1000 (src/cluster_argument_parsing.rs:1301)
    uint8_t kmer_length;   // 21   (src/cluster_argument_parsing.rs:1302)
    bool low_memory;
    int threads;

    FinchPreclusterer(std::shared_ptr<HipContext> hip, float min_ani, size_t num_kmers, uint8_t kmer_length,
                      bool low_memory = false, int threads = 1)
        : min_ani(min_ani), num_kmers(num_kmers), kmer_length(kmer_length), low_memory(low_memory), threads(threads),
          hip_(std::move(hip)) {}

    SortedPairGenomeDistanceCache distances(const std::vector<std::string> &genome_fasta_paths) override {
        if (low_memory)                                                                                 // finch.rs:14-15
            throw std::runtime_error("Low-memory clustering currently only supported with skani preclusterer");
        return finch::distances(*hip_, genome_fasta_paths, min_ani, num_kmers, kmer_length, threads, &last_edges);
    }
    SortedPairGenomeDistanceCache distances_contigs(const std::vector<std::string> &,
                                                    const std::vector<std::string> &) override {
        return SortedPairGenomeDistanceCache();                                                         // finch.rs:26-33
    }
    SortedPairGenomeDistanceCache distances_with_references(const std::vector<std::string> &,
                                                            const std::vector<std::string> &) override {
        throw std::runtime_error("Reference genome clustering currently only supported with skani preclusterer");  // :40
    }
    std::string method_name() const override { return "finch"; }  // the string matters: src/clusterer.rs:33,39

    std::vector<ghip_pair> last_edges;  // the last distances() result as a sorted edge list
    const std::shared_ptr<HipContext> &context() const { return hip_; }

private:
    std::shared_ptr<HipContext> hip_;
};

// ---------------------------------------------------------------------------------------------------------------
// The clusterer-stage ANI back-end (replaces SkaniClusterer, src/skani.rs:689-716): device-resident FracMinHash index
// built once per genome list; calculate_ani answers from it.  Build-defined estimator.
class HipAniClusterer : public ClusterDistanceFinder {
public:
    float threshold;              // percent, like SkaniClusterer::threshold
    float min_aligned_threshold;  // fraction
    bool small_genomes;
    int threads;

    HipAniClusterer(std::shared_ptr<HipContext> hip, float threshold, float min_aligned_threshold,
                    bool small_genomes = false, int threads = 1)
        : threshold(threshold), min_aligned_threshold(min_aligned_threshold), small_genomes(small_genomes),
          threads(threads), hip_(std::move(hip)) {}
    ~HipAniClusterer() override { ghip_ani_index_free(index_); }

    void initialise() const override {                                                                  // skani.rs:696-698
        if (!(threshold > 1.0f)) throw std::runtime_error("assertion failed: self.threshold > 1.0");
    }
    std::string method_name() const override { return "hipani"; }
    float get_ani_threshold() const override { return threshold; }

    // index the genome list once (what a per-pair `skani dist` subprocess re-does for both genomes every call)
    void prepare(const std::vector<std::string> &genomes) {
        ghip_ani_index_free(index_);
        index_ = nullptr;
        path_index_.clear();
        auto paths = detail::c_paths(genomes);
        ghip_genomes *g = nullptr;
        if (ghip_genomes_from_files(hip_->get(), paths.data(), paths.size(), threads, &g) != GHIP_OK)
            throw std::runtime_error("Failed to read genomes for ANI: " + hip_->last_error());
        const int rc = ghip_ani_index_build(hip_->get(), g, 15, small_genomes ? 30 : 125, 20000, &index_);
        ghip_genomes_free(g);
        if (rc != GHIP_OK) throw std::runtime_error("Failed to build the ANI index: " + hip_->last_error());
        for (size_t i = 0; i < genomes.size(); i++) path_index_.emplace(genomes[i], (uint32_t)i);
        order_ = genomes;
    }
    // the index serves `genomes` only if genome i of the list is genome i of the index: same paths, same order (an
    // index grown by calculate_ani()/prepare_missing, or built for another list of the same length, does not)
    bool prepared_for(const std::vector<std::string> &genomes) const { return index_ && order_ == genomes; }
    // take an index built elsewhere (galah::cluster builds sketches and index from one pass over the bases)
    void adopt_index(ghip_ani_index *index, const std::vector<std::string> &genomes) {
        ghip_ani_index_free(index_);
        index_ = index;
        path_index_.clear();
        for (size_t i = 0; i < genomes.size(); i++) path_index_.emplace(genomes[i], (uint32_t)i);
        order_ = genomes;
    }
    uint32_t seed_compression() const { return small_genomes ? 30 : 125; }
    const std::shared_ptr<HipContext> &context() const { return hip_; }

    // Called concurrently by the reference's rayon workers (src/clusterer.rs:267-270,283-293,375-399; the `C: Sync`
    // bound of clusterer.rs:14): lookups share the index, only indexing a genome not seen before is exclusive.
    std::optional<float> calculate_ani(const std::string &fasta1, const std::string &fasta2) override {
        std::shared_lock<std::shared_mutex> shared(mu_);
        while (!path_index_.count(fasta1) || !path_index_.count(fasta2)) {
            shared.unlock();
            {
                std::unique_lock<std::shared_mutex> exclusive(mu_);
                if (!path_index_.count(fasta1) || !path_index_.count(fasta2)) prepare_missing(fasta1, fasta2);
            }
            shared.lock();
        }
        const uint32_t pair[2] = {path_index_.at(fasta1), path_index_.at(fasta2)};
        float ani = 0.0f;
        if (ghip_ani_pairs(hip_->get(), index_, pair, 1, min_aligned_threshold, &ani, nullptr) != GHIP_OK)
            throw std::runtime_error("ANI failed: " + hip_->last_error());
        return ani;                                                  // always Some(..): src/skani.rs:709
    }
    // the batch form galah::cluster uses: pairs = flat (a, b) genome indices of the prepared list
    std::vector<float> calculate_ani_indices(const std::vector<uint32_t> &pairs) {
        std::vector<float> out(pairs.size() / 2);
        if (!out.empty() && ghip_ani_pairs(hip_->get(), index_, pairs.data(), out.size(), min_aligned_threshold,
                                           out.data(), nullptr) != GHIP_OK)
            throw std::runtime_error("ANI failed: " + hip_->last_error());
        return out;
    }

private:
    void prepare_missing(const std::string &f1, const std::string &f2) {  // calculate_ani on an un-prepared pair
        std::vector<std::string> g = order_;  // keep the indices of the genomes already prepared
        if (!path_index_.count(f1)) g.push_back(f1);
        if (!path_index_.count(f2) && f2 != f1) g.push_back(f2);
        prepare(g);
    }
    std::shared_ptr<HipContext> hip_;
    ghip_ani_index *index_ = nullptr;
    std::unordered_map<std::string, uint32_t> path_index_;
    std::vector<std::string> order_;  // genome i of the index
    std::shared_mutex mu_;  // calculate_ani: shared for lookups, exclusive while (re)indexing
};

// ---------------------------------------------------------------------------------------------------------------
// clusterer::cluster (src/clusterer.rs:14-152).  Returns Vec<Vec<usize>>, the representative first in each cluster.
// The O(N^2) host loops (partition_sketches, transform_ids) run on the sorted edge list inside ghip_cluster; a
// HipAniClusterer is asked in batches for the pairs that touch a representative (ghip_cluster_lazy), any other
// ClusterDistanceFinder through calculate_ani -- the clusters are the same because calculate_ani is a pure function of
// the pair.
inline uint64_t &last_ani_pairs_requested() {  // how many precluster pairs the last cluster() asked a HipAniClusterer for
    static thread_local uint64_t v = 0;
    return v;
}
inline std::vector<std::vector<size_t>> cluster(const std::vector<std::string> &genomes,
                                                PreclusterDistanceFinder &preclusterer, ClusterDistanceFinder &clusterer,
                                                bool cluster_contigs = false,
                                                const std::vector<std::string> *contig_names = nullptr,
                                                const std::vector<std::string> *reference_genomes = nullptr) {
    clusterer.initialise();
    const std::string preclusterer_name = preclusterer.method_name(), clusterer_name = clusterer.method_name();
    bool skip_clusterer = false;
    if (clusterer_name == preclusterer_name) skip_clusterer = true;                          // clusterer.rs:32-36
    if (cluster_contigs) {                                                                    // clusterer.rs:38-44
        if (preclusterer_name == "finch") throw std::runtime_error(preclusterer_name + " does not support contig comparisons.");
        skip_clusterer = true;
    }
    SortedPairGenomeDistanceCache cache;                                                      // clusterer.rs:47-54
    auto *hip_finch = dynamic_cast<FinchPreclusterer *>(&preclusterer);
    auto *hip_ani = dynamic_cast<HipAniClusterer *>(&clusterer);
    if (reference_genomes) cache = preclusterer.distances_with_references(genomes, *reference_genomes);
    else if (cluster_contigs) cache = preclusterer.distances_contigs(genomes, *contig_names);
    else if (!skip_clusterer && hip_finch && hip_ani && !hip_finch->low_memory && hip_finch->context() == hip_ani->context()) {
        // both back-ends are HIP: read every FASTA once, MinHash sketches AND ANI index from one pass over the bases
        HipContext &hip = *hip_finch->context();
        auto paths = detail::c_paths(genomes);
        ghip_sketches *sk = nullptr;
        ghip_ani_index *idx = nullptr;
        int rc = ghip_sketch_and_index_files(hip.get(), paths.data(), paths.size(), hip_finch->kmer_length, (uint32_t)hip_finch->num_kmers,
                                             0, 15, hip_ani->seed_compression(), 20000, std::max(hip_finch->threads, hip_ani->threads),
                                             0, &sk, &idx, nullptr);
        if (rc != GHIP_OK) throw std::runtime_error("Failed to sketch genomes with finch: " + hip.last_error());   // finch.rs:72
        ghip_pair *pairs = nullptr;
        size_t np = 0;
        rc = ghip_precluster(hip.get(), sk, hip_finch->min_ani, &pairs, &np);
        ghip_sketches_free(sk);
        if (rc != GHIP_OK) { ghip_ani_index_free(idx); throw std::runtime_error("Failed to compare finch sketches: " + hip.last_error()); }
        for (size_t x = 0; x < np; x++) cache.insert({pairs[x].i, pairs[x].j}, pairs[x].ani);
        hip_finch->last_edges.assign(pairs, pairs + np);
        ghip_free(pairs);
        hip_ani->adopt_index(idx, genomes);
    } else cache = preclusterer.distances(genomes);
    const size_t n = cluster_contigs ? contig_names->size() : genomes.size();

    std::vector<ghip_pair> edges;
    edges.reserve(cache.len());
    for (auto &kv : cache.internal) {
        if (!kv.second) throw std::runtime_error("precluster cache holds None: not produced by a precluster back-end");
        edges.push_back(ghip_pair{(uint32_t)kv.first.first, (uint32_t)kv.first.second, 0, 0, *kv.second});
    }
    struct CbState {
        ClusterDistanceFinder *c; const std::vector<std::string> *g; HipAniClusterer *hip; const std::vector<ghip_pair> *edges;
        std::exception_ptr failed;
    } st{&clusterer, &genomes, nullptr, &edges, nullptr};
    ghip_ani_callback cb = nullptr;
    ghip_ani_batch_callback batch_cb = nullptr;
    if (!skip_clusterer) {
        if (auto *hipani = dynamic_cast<HipAniClusterer *>(&clusterer)) {
            // the HIP clusterer is asked in BATCHES, and only for the pairs the greedy rules look at (those touching a
            // representative) -- the reference's laziness (clusterer.rs:178-200, 301-334) without one launch per pair
            if (!hipani->prepared_for(genomes)) hipani->prepare(genomes);
            st.hip = hipani;
            batch_cb = [](void *user, const uint32_t *edge_idx, size_t n_edges, float *out) -> int {
                auto *s = static_cast<CbState *>(user);
                try {
                    std::vector<uint32_t> idx(2 * n_edges);
                    for (size_t x = 0; x < n_edges; x++) {
                        idx[2 * x] = (*s->edges)[edge_idx[x]].i;
                        idx[2 * x + 1] = (*s->edges)[edge_idx[x]].j;
                    }
                    auto v = s->hip->calculate_ani_indices(idx);
                    std::copy(v.begin(), v.end(), out);
                    return 0;
                } catch (...) {
                    s->failed = std::current_exception();
                    return -1;
                }
            };
        } else {
            cb = [](void *user, uint32_t a, uint32_t b, float *out) -> int {
                auto *s = static_cast<CbState *>(user);
                try {  // no exception may cross the C ABI: stop the clusterer (< 0) and rethrow below
                    auto v = s->c->calculate_ani((*s->g)[a], (*s->g)[b]);
                    if (!v) return 0;
                    *out = *v;
                    return 1;
                } catch (...) {
                    s->failed = std::current_exception();
                    return -1;
                }
            };
        }
    }
    uint32_t *members = nullptr;
    uint64_t *offsets = nullptr;
    size_t n_clusters = 0;
    const int rc = batch_cb ? ghip_cluster_lazy(n, edges.data(), edges.size(), clusterer.get_ani_threshold(), batch_cb, &st, &members,
                                                &offsets, &n_clusters, &last_ani_pairs_requested())
                            : ghip_cluster(n, edges.data(), edges.size(), nullptr, skip_clusterer ? 1 : 0,
                                           clusterer.get_ani_threshold(), cb, &st, &members, &offsets, &n_clusters);
    if (st.failed) std::rethrow_exception(st.failed);
    if (rc != GHIP_OK)  // best_rep.unwrap() on None (src/clusterer.rs:444)
        throw std::runtime_error("called `Option::unwrap()` on a `None` value: a genome has no representative");
    std::vector<std::vector<size_t>> out(n_clusters);
    for (size_t c = 0; c < n_clusters; c++) out[c].assign(members + offsets[c], members + offsets[c + 1]);
    ghip_free(members);
    ghip_free(offsets);
    return out;
}

// GalahClusterer (src/cluster_argument_parsing.rs:108-115, 1514-1530): the library entry CoverM and `galah process` use
// (src/process.rs:128-145) -- the genome list and the two back-ends; cluster() is clusterer::cluster, element 0 of every
// inner vector the representative (src/cluster_argument_parsing.rs:730).
struct GalahClusterer {
    std::vector<std::string> genome_fasta_paths;
    PreclusterDistanceFinder *preclusterer = nullptr;
    ClusterDistanceFinder *clusterer = nullptr;
    bool cluster_contigs = false;
    const std::vector<std::string> *contig_names = nullptr;       // Option<Vec<&str>>
    const std::vector<std::string> *reference_genomes = nullptr;  // Option<Vec<String>>
    std::vector<std::vector<size_t>> cluster() const {
        if (!preclusterer || !clusterer) throw std::invalid_argument("GalahClusterer needs a preclusterer and a clusterer");
        return galah::cluster(genome_fasta_paths, *preclusterer, *clusterer, cluster_contigs, contig_names, reference_genomes);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// The same `cluster` on SEVERAL GPUs driven by this one process (galah's CLI is a single process): one context and one
// thread per device, genomes in contiguous blocks, sketch matrix all-gathered by peer copies over xGMI, pair work dealt
// over the devices, the native clusterer's lazy ANI rounds with each round's pairs computed where their first genome lives
// (ghip_cluster_files_multi -> ghip_cluster_ranks: the algorithm one device runs).  Same clusters as cluster(genomes, FinchPreclusterer, HipAniClusterer) on one device.
inline std::vector<std::vector<size_t>> cluster_multi_gpu(const std::vector<std::shared_ptr<HipContext>> &hips,
                                                          const std::vector<std::string> &genomes, const FinchPreclusterer &pre,
                                                          const HipAniClusterer &cl) {
    cl.initialise();
    if (pre.low_memory) throw std::runtime_error("Low-memory clustering currently only supported with skani preclusterer");
    if (hips.empty()) throw std::runtime_error("cluster_multi_gpu needs at least one HIP context");
    std::vector<ghip_ctx *> ctxs;
    for (auto &h : hips) ctxs.push_back(h->get());
    auto paths = detail::c_paths(genomes);
    uint32_t *members = nullptr;
    uint64_t *offsets = nullptr;
    size_t n_clusters = 0;
    const int rc = ghip_cluster_files_multi(ctxs.data(), (uint32_t)ctxs.size(), paths.data(), paths.size(), pre.kmer_length,
                                            (uint32_t)pre.num_kmers, pre.min_ani, cl.threshold, cl.min_aligned_threshold,
                                            cl.seed_compression(), std::max(pre.threads, cl.threads), &members, &offsets, &n_clusters);
    if (rc != GHIP_OK) throw std::runtime_error("Failed to sketch genomes with finch: " + hips[0]->last_error());
    std::vector<std::vector<size_t>> out(n_clusters);
    for (size_t c = 0; c < n_clusters; c++) out[c].assign(members + offsets[c], members + offsets[c + 1]);
    ghip_free(members);
    ghip_free(offsets);
    return out;
}

}  // namespace galah
