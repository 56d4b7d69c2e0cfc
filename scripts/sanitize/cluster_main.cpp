#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "galah_hip.h"
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
        if (rc == 0) { size_t tot = o[nc]; if (tot != n) { printf("BAD total %zu != %zu\n", tot, n); return 1; } ghip_free(m); ghip_free(o); }
    }
    printf("cluster asan ok\n");
}
