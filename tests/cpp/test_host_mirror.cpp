// The reference's own unit tests for this path, restated against the C++ host mirror (include/galah_hip.hpp) and run
// on the GPU through the C ABI.  Usage: test_host_mirror <tests/golden/fasta dir>.  Exit code 0 = all passed.
//   src/finch.rs:111-128                       test_finch_hello_world, test_finch_no_pairs
//   src/sorted_pair_genome_distance_cache.rs   insert / get / contains_key / transform_ids semantics
//   src/clusterer.rs:631-690                   test_minhash_skani_hello_world, .._two_clusters_same_ani
//                                              (the reference needs the skani binary; the build-defined ANI reproduces both)
//   tests/test_cmdline.rs:262-302              --min-aligned-fraction 0.2 -> one representative, 0.6 -> two
//   tests/test_cmdline.rs:161-181, 417-440     representative list order; github issue 7
//   src/clusterer.rs:14,267-296,375-399        `C: Sync`: calculate_ani called from many threads at once
//   src/finch.rs:14-15,40, src/clusterer.rs:38-44   refusals (panics -> std::runtime_error with the same text)
#include <cstdlib>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "galah_hip.hpp"

static int failures = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

template <typename F>
static bool panics_with(F &&f, const char *needle) {
    try { f(); } catch (const std::runtime_error &e) { return std::strstr(e.what(), needle) != nullptr; }
    return false;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::printf("usage: %s <fasta dir>\n", argv[0]); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    auto fa = [&](const char *name) { return d + name + ".fna.gz"; };
    auto hip = std::make_shared<galah::HipContext>(0);

    {   // a persisted sketch matrix + new files = the cache of a full run, only the new files sketched
        const std::vector<std::string> old = {fa("abisko_S1X13"), fa("abisko_S2D19"), fa("set1_1mbp")}, fresh = {fa("abisko_S3X12"), fa("set1_500kb")};
        std::vector<std::string> all = old;
        all.insert(all.end(), fresh.begin(), fresh.end());
        std::vector<ghip_pair> full_edges, old_edges, inc_edges;
        std::vector<std::string> names;
        const char *tmp = std::getenv("TMPDIR");
        const std::string path = std::string(tmp ? tmp : "/tmp") + "/ghip_host_mirror_matrix.ghipsk";
        auto full = galah::finch::distances(*hip, all, 0.9f, 1000, 21, 4, &full_edges);
        galah::finch::distances_and_save(*hip, old, 0.9f, 1000, 21, path, 4, &old_edges);
        auto inc = galah::finch::distances_incremental(*hip, path, fresh, 0.9f, 1000, 21, &old_edges, &names, 4, &inc_edges);
        CHECK(names == all);
        CHECK(inc_edges.size() == full_edges.size() && full_edges.size() >= 4);
        bool same = inc_edges.size() == full_edges.size();
        for (size_t x = 0; same && x < full_edges.size(); x++) same = std::memcmp(&inc_edges[x], &full_edges[x], sizeof(ghip_pair)) == 0;
        CHECK(same);
        CHECK(inc == full);
        std::remove(path.c_str());
    }

    {   // src/finch.rs:111-119 test_finch_hello_world
        galah::SortedPairGenomeDistanceCache expected;
        expected.insert({0, 1}, 0.9808188f);
        auto got = galah::finch::distances(*hip, {fa("set1_1mbp"), fa("set1_500kb")}, 0.9f, 1000, 21);
        CHECK(got == expected);
    }
    {   // src/finch.rs:121-128: nothing at 0.99
        auto got = galah::finch::distances(*hip, {fa("set1_1mbp"), fa("set1_500kb")}, 0.99f, 1000, 21);
        CHECK(got == galah::SortedPairGenomeDistanceCache());
    }
    {   // cache semantics
        galah::SortedPairGenomeDistanceCache c;
        c.insert({3, 1}, 0.5f);
        c.insert({2, 5}, std::nullopt);
        CHECK(c.contains_key({1, 3}) && c.contains_key({3, 1}) && !c.contains_key({1, 2}));
        CHECK(c.get({1, 3}).has_value() && **c.get({1, 3}) == 0.5f);
        CHECK(c.get({5, 2}).has_value() && !c.get({5, 2})->has_value());   // present, None
        CHECK(!c.get({0, 9}).has_value());
        auto t = c.transform_ids({5, 1, 3});                               // (1,3) -> positions (1,2)
        CHECK(t.len() == 1 && t.contains_key({1, 2}) && **t.get({2, 1}) == 0.5f);
    }
    const std::vector<std::string> abisko = {fa("abisko_S1X13"), fa("abisko_S2D19"), fa("abisko_S3X12"), fa("abisko_S2D13")};
    auto sorted = [](std::vector<std::vector<size_t>> c) { std::sort(c.begin(), c.end()); return c; };
    {   // src/clusterer.rs:631-659: finch 0.9 + skani-equivalent at 95, min aligned fraction 0.2 -> [[0,1,2,3]]
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        galah::HipAniClusterer cl(hip, 95.0f, 0.2f);
        auto clusters = sorted(galah::cluster(abisko, pre, cl));
        CHECK((clusters == std::vector<std::vector<size_t>>{{0, 1, 2, 3}}));
        CHECK(pre.last_edges.size() == 6);
        // a round this short is topped up with every edge still open (one batch); with that off the HIP clusterer is asked lazily:
        // genome 0 is the representative, only its 3 edges are ever looked at
        CHECK(galah::last_ani_pairs_requested() == 6);
        ghip_options opt;   // (ghip_cluster_lazy has no context: the process-wide options)
        CHECK(ghip_get_options(nullptr, &opt) == GHIP_OK && opt.struct_size == sizeof(ghip_options) && opt.lazy_flush_below == 512);
        opt.lazy_flush_below = 0;
        CHECK(ghip_set_options(nullptr, &opt) == GHIP_OK);
        CHECK((sorted(galah::cluster(abisko, pre, cl)) == std::vector<std::vector<size_t>>{{0, 1, 2, 3}}));
        CHECK(galah::last_ani_pairs_requested() == 3);
        opt.lazy_flush_below = 512;
        CHECK(ghip_set_options(nullptr, &opt) == GHIP_OK);
        // the library entry (GalahClusterer, src/cluster_argument_parsing.rs:108-115,1514-1530): same call, same clusters
        galah::GalahClusterer gc;
        gc.genome_fasta_paths = abisko; gc.preclusterer = &pre; gc.clusterer = &cl;
        CHECK((sorted(gc.cluster()) == std::vector<std::vector<size_t>>{{0, 1, 2, 3}}));
    }
    {   // src/clusterer.rs:661-690: at 99 -> [[0,1,3],[2]]
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        galah::HipAniClusterer cl(hip, 99.0f, 0.2f);
        auto clusters = sorted(galah::cluster(abisko, pre, cl));
        CHECK((clusters == std::vector<std::vector<size_t>>{{0, 1, 3}, {2}}));
        // calculate_ani through the trait object, one pair at a time, gives the batch's values
        galah::ClusterDistanceFinder &tr = cl;
        auto a01 = tr.calculate_ani(abisko[0], abisko[1]);
        CHECK(a01.has_value() && *a01 >= 99.0f && *tr.calculate_ani(abisko[0], abisko[2]) < 99.0f);
    }
    {   // skip_clusterer: preclusterer and clusterer of the same name reuse the precluster ANI (src/clusterer.rs:32-36)
        struct SameName : galah::ClusterDistanceFinder {
            void initialise() const override {}
            std::string method_name() const override { return "finch"; }
            float get_ani_threshold() const override { return 0.95f; }   // finch ANI is a fraction
            std::optional<float> calculate_ani(const std::string &, const std::string &) override {
                throw std::runtime_error("must not be called");
            }
        } same;
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        auto clusters = sorted(galah::cluster(abisko, pre, same));
        // finch ANIs: (0,1) .9894 (0,2) .9793 (0,3) .9973 (1,2) .9847 (1,3) .9893 (2,3) .9791 -> one cluster at 0.95
        CHECK((clusters == std::vector<std::vector<size_t>>{{0, 1, 2, 3}}));
    }
    {   // tests/test_cmdline.rs:262-302
        const std::vector<std::string> g = {fa("set2_1mbp"), fa("set2_half")};
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        galah::HipAniClusterer lo(hip, 95.0f, 0.2f), hi(hip, 95.0f, 0.6f);
        CHECK((sorted(galah::cluster(g, pre, lo)) == std::vector<std::vector<size_t>>{{0, 1}}));
        CHECK((sorted(galah::cluster(g, pre, hi)) == std::vector<std::vector<size_t>>{{0}, {1}}));
    }
    {   // tests/test_cmdline.rs:161-181 (representative list) and :417-440 (github issue 7, --min-aligned-fraction 60)
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        galah::HipAniClusterer a(hip, 95.0f, 0.15f), b(hip, 95.0f, 0.6f);
        CHECK((galah::cluster({fa("clash_500kb"), fa("set1_500kb"), fa("set1_1mbp")}, pre, a) == std::vector<std::vector<size_t>>{{1, 2}, {0}}));
        CHECK((galah::cluster({fa("antonio_MAG52"), fa("antonio_MAG189")}, pre, b) == std::vector<std::vector<size_t>>{{0, 1}}));
    }
    {   // refusals
        galah::FinchPreclusterer lowmem(hip, 0.9f, 1000, 21, true);
        CHECK(panics_with([&] { lowmem.distances(abisko); }, "Low-memory clustering currently only supported with skani preclusterer"));
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        CHECK(panics_with([&] { pre.distances_with_references(abisko, abisko); },
                          "Reference genome clustering currently only supported with skani preclusterer"));
        CHECK(pre.distances_contigs(abisko, abisko).len() == 0);
        galah::HipAniClusterer cl(hip, 95.0f, 0.2f);
        std::vector<std::string> names = {"a"};
        CHECK(panics_with([&] { galah::cluster(abisko, pre, cl, true, &names); }, "finch does not support contig comparisons."));
        galah::HipAniClusterer bad(hip, 0.95f, 0.2f);   // threshold given as a fraction: skani.rs:696-698 asserts > 1
        CHECK(panics_with([&] { bad.initialise(); }, "self.threshold > 1.0"));
        CHECK(panics_with([&] { galah::finch::distances(*hip, {d + "does_not_exist.fna"}, 0.9f, 1000, 21); },
                          "Failed to sketch genomes with finch"));
    }
    {   // calculate_ani is called from rayon workers (src/clusterer.rs:267-270,283-293,375-399): 8 threads ask for every
        // ordered pair of six genomes at once -- two of them not indexed yet, so some calls re-index while others
        // look up -- and must all get the values of a quiet, sequential pass
        std::vector<std::string> six = abisko;
        six.push_back(fa("set1_1mbp"));
        six.push_back(fa("set1_500kb"));
        galah::HipAniClusterer quiet(hip, 95.0f, 0.15f), busy(hip, 95.0f, 0.15f);
        quiet.prepare(six);
        busy.prepare(abisko);
        std::vector<float> want(36), got(36 * 8, -1.0f);
        for (int x = 0; x < 36; x++) want[x] = *quiet.calculate_ani(six[x / 6], six[x % 6]);
        std::vector<std::thread> pool;
        for (int t = 0; t < 8; t++)
            pool.emplace_back([&, t] {
                for (int y = 0; y < 36; y++) {
                    const int x = (y * 7 + t * 5) % 36;   // every thread its own order
                    got[t * 36 + x] = *busy.calculate_ani(six[x / 6], six[x % 6]);
                }
            });
        for (auto &th : pool) th.join();
        bool same = true;
        for (int t = 0; t < 8; t++)
            for (int x = 0; x < 36; x++) same = same && got[t * 36 + x] == want[x];
        CHECK(same);
        CHECK(want[0 * 6 + 1] > 95.0f && want[4 * 6 + 5] > 95.0f && want[0 * 6 + 4] == 0.0f);
    }
    {   // unrelated genomes: no precluster pair at all -> one singleton cluster each, through the batched ANI clusterer
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        galah::HipAniClusterer cl(hip, 95.0f, 0.15f);
        CHECK((galah::cluster({fa("abisko_S1X13"), fa("antonio_MAG52"), fa("set1_1mbp")}, pre, cl) ==
               std::vector<std::vector<size_t>>{{0}, {1}, {2}}));
    }
    {   // a clusterer prepared for ANOTHER list of the same length (or the same genomes in another order) must be
        // re-prepared: edge indices are positions in the list given to cluster()
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21);
        galah::HipAniClusterer fresh(hip, 95.0f, 0.15f), stale(hip, 95.0f, 0.15f);
        std::vector<std::string> fwd = {fa("set1_500kb"), fa("set1_1mbp"), fa("antonio_MAG52"), fa("antonio_MAG189")};
        std::vector<std::string> rev(fwd.rbegin(), fwd.rend());
        stale.prepare(rev);
        CHECK(!stale.prepared_for(fwd) && stale.prepared_for(rev));
        galah::FinchPreclusterer other(std::make_shared<galah::HipContext>(0), 0.9f, 1000, 21);  // other context: no fused ingest
        CHECK((galah::cluster(fwd, other, stale) == galah::cluster(fwd, pre, fresh)));
        CHECK(stale.prepared_for(fwd));
    }
    {   // one process, several GPUs (here: three contexts on the one device, the peer-copy transport): same clusters
        galah::FinchPreclusterer pre(hip, 0.9f, 1000, 21, false, 4);
        galah::HipAniClusterer cl(hip, 95.0f, 0.15f, false, 4);
        std::vector<std::string> all = abisko;
        for (const char *x : {"antonio_MAG52", "antonio_MAG189", "set1_1mbp", "set1_500kb", "abisko_S1D21", "abisko_S2M16"}) all.push_back(fa(x));
        const auto want = galah::cluster(all, pre, cl);
        std::vector<std::shared_ptr<galah::HipContext>> three = {std::make_shared<galah::HipContext>(0), std::make_shared<galah::HipContext>(0),
                                                                 std::make_shared<galah::HipContext>(0)};
        CHECK((galah::cluster_multi_gpu(three, all, pre, cl) == want));
        CHECK((galah::cluster_multi_gpu({hip}, all, pre, cl) == want));
        // the communicator entry points directly: two ranks on threads, a device all-gather and a candidate-list merge
        ghip_ctx *ctxs[2] = {three[0]->get(), three[1]->get()};
        ghip_comm *comms[2] = {nullptr, nullptr};
        CHECK(ghip_comm_init_local(ctxs, 2, comms) == GHIP_OK);
        CHECK(std::string(ghip_comm_transport(comms[0])) == "local-peer-copy" && ghip_comm_world(comms[1]) == 2 && ghip_comm_rank(comms[1]) == 1);
        bool ok[2] = {false, false};
        auto rank = [&](int r) {
            ghip_pair mine[2] = {{(uint32_t)r, 5, 1, 2, 0.9f}, {(uint32_t)(r + 2), 7, 3, 4, 0.95f}};   // sorted shares
            ghip_pair *allp = nullptr;
            size_t n = 0;
            if (ghip_allgather_pairs(comms[r], mine, 2, &allp, &n) != GHIP_OK || n != 4) return;
            ok[r] = allp[0].i == 0 && allp[1].i == 1 && allp[2].i == 2 && allp[3].i == 3 && allp[3].j == 7;
            ghip_free(allp);
            uint64_t v = 40 + r, got[2] = {0, 0};
            ok[r] = ok[r] && ghip_comm_allgather_host(comms[r], &v, sizeof v, got) == GHIP_OK && got[0] == 40 && got[1] == 41;
        };
        std::thread t1(rank, 1);
        rank(0);
        t1.join();
        CHECK(ok[0] && ok[1]);
        ghip_comm_destroy(comms[0]);
        ghip_comm_destroy(comms[1]);
    }
    std::printf(failures ? "%d check(s) failed\n" : "host mirror: all reference tests passed\n", failures);
    return failures ? 1 : 0;
}
