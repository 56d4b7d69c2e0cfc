#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <functional>
#include <string>
#include <unordered_map>
#include "galah_hip.h"
#include "ghip_internal.h"
// host-only stand-ins for what cluster.cpp takes from the rest of the library: the error setter, the worker pool (run inline)
// and ghip_ani_pairs, which here looks the pair's ANI up in a table the test fills (keyed by genome indices)
static std::unordered_map<uint64_t, float> g_ani;
int ghip_set_error(ghip_ctx *, int code, const std::string &) { return code; }
void ghip_io_pool::run(int n, std::function<void(int)> fn) { for (int w = 0; w < n; w++) fn(w); }
extern "C" int ghip_ani_pairs(ghip_ctx *, const ghip_ani_index *, const uint32_t *pairs, size_t n, float, float *out_ani, float *) {
    for (size_t x = 0; x < n; x++) out_ani[x] = g_ani.at(((uint64_t)pairs[2 * x] << 32) | pairs[2 * x + 1]);
    return 0;
}
int main() {
    std::mt19937 rng(3);
    for (int rep = 0; rep < 300; rep++) {
        size_t n = 1 + rng() % 300;
        std::vector<ghip_pair> pairs; std::vector<float> ani;
        for (uint32_t i = 0; i < n; i++) for (uint32_t j = i + 1; j < n; j++)
            if (rng() % 100 < 5 || (i / 7 == j / 7 && rng() % 100 < 80)) { pairs.push_back({i, j, 0, 0, 0.9f + (rng() % 1000) * 1e-4f}); ani.push_back(90.0f + (rng() % 1000) * 0.01f); }
        uint32_t *m = nullptr; uint64_t *o = nullptr; size_t nc = 0;
        int skip = rng() % 3 == 0;
        int rc = ghip_cluster(n, pairs.data(), pairs.size(), skip ? nullptr : ani.data(), skip, skip ? 0.95f : 95.0f, nullptr, nullptr, &m, &o, &nc);
        if (rc == 0) { size_t tot = o[nc]; if (tot != n) { printf("BAD total %zu != %zu\n", tot, n); return 1; } }
        if (!skip) {   // the lazy, batched form: same clusters, asked in rounds (with and without the short-list shortcut)
            struct St { const std::vector<float> *ani; size_t rounds; } st{&ani, 0};
            auto cb = [](void *u, const uint32_t *e, size_t k, float *out) -> int {
                St *s = static_cast<St *>(u);
                s->rounds++;
                for (size_t x = 0; x < k; x++) out[x] = (*s->ani)[e[x]];
                return 0;
            };
            setenv("GHIP_LAZY_FLUSH_BELOW", rep % 3 == 0 ? "0" : (rep % 3 == 1 ? "16" : "100000"), 1);
            uint32_t *m2 = nullptr; uint64_t *o2 = nullptr; size_t nc2 = 0; uint64_t asked = 0;
            const int rc2 = ghip_cluster_lazy(n, pairs.data(), pairs.size(), 95.0f, cb, &st, &m2, &o2, &nc2, &asked);
            if (rc2 != rc) { printf("BAD lazy rc %d != %d\n", rc2, rc); return 1; }
            if (rc2 == 0) {
                if (nc2 != nc || asked > pairs.size()) { printf("BAD lazy clusters %zu != %zu\n", nc2, nc); return 1; }
                for (size_t x = 0; x < n; x++) if (m2[x] != m[x]) { printf("BAD lazy member at %zu\n", x); return 1; }
                for (size_t c = 0; c <= nc; c++) if (o2[c] != o[c]) { printf("BAD lazy offset\n"); return 1; }
                ghip_free(m2); ghip_free(o2);
            }
        }
        if (!skip && rc == 0) {   // ghip_cluster_index: the same rounds answered "inside the library", in a random genome order
            static ghip_ctx *ctx = new ghip_ctx();
            static ghip_ani_index *idx = reinterpret_cast<ghip_ani_index *>(ctx);   // (opaque to cluster.cpp)
            g_ani.clear();
            for (size_t e = 0; e < pairs.size(); e++) g_ani[((uint64_t)pairs[e].i << 32) | pairs[e].j] = ani[e];
            std::vector<uint32_t> order(n);
            for (uint32_t x = 0; x < n; x++) order[x] = x;
            for (size_t x = n; x > 1; x--) std::swap(order[x - 1], order[rng() % x]);
            uint32_t *m3 = nullptr; uint64_t *o3 = nullptr; size_t nc3 = 0; uint64_t stats[4];
            const int rc3 = ghip_cluster_index(ctx, idx, n, pairs.data(), pairs.size(), rep % 2 ? order.data() : nullptr, 95.0f, 0.15f, &m3, &o3, &nc3, stats);
            if (rc3 != 0 && rc3 != GHIP_EINVAL) { printf("BAD cluster_index rc %d\n", rc3); return 1; }   // (EINVAL: a genome without a representative, as the reference panics)
            if (rc3 == 0) {
                if (o3[nc3] != n) { printf("BAD cluster_index total\n"); return 1; }
                if (!(rep % 2)) for (size_t x = 0; x < n; x++) if (m3[x] != m[x]) { printf("BAD cluster_index member\n"); return 1; }
                ghip_free(m3); ghip_free(o3);
            }
        }
        if (rc == 0) { ghip_free(m); ghip_free(o); }
    }
    printf("cluster asan ok\n");
}
