// Host-side MODEL of the dense probe form's arranged variant (galah_amd/csrc/pairs_probe.hip), run in the CPU suite
// (tests/test_abi.py builds it with g++): the kernels' pure functions are SHARED with this file through
// galah_amd/csrc/probe_common.h (bucket choices with the constrained second bucket, the 31-bit tags); the algorithms
// around them -- random-walk cuckoo insertion, the arrangement of a row by bucket residue, the tag probe, the exact recount --
// are restated serially.  What it checks is what makes the form EXACT whatever the arrangement does to speed:
//   1. every hash of a sketch sits in one of its two buckets of the table, for every number of constrained bits;
//   2. an arranged row is a permutation of the sketch's hashes (holes are 2^64 - 1), overfull residue classes included;
//   3. the tag probe over the arranged row counts common' >= |A n B| (a superset list), the full-key recount counts exactly.
// It cannot check the kernel code itself (round 4 had no GPU for most of its length); it pins the design.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "../../galah_amd/csrc/probe_common.h"

using namespace ghip_probe;
static const uint64_t EMPTY = ~0ull;
static uint64_t rs = 88172645463325252ull;
static uint64_t rnd64() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs * 0x2545F4914F6CDD1Dull; }
static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

// pair_table_build_kernel, one thread
static bool build_table(const std::vector<uint64_t> &row, uint32_t buckets, uint32_t cbits, std::vector<uint64_t> &tab) {
    const uint32_t mask = buckets - 1;
    tab.assign(2 * buckets, EMPTY);
    for (size_t i = 0; i < row.size(); i++) {
        uint64_t x = row[i];
        uint32_t b = bucket1(x, mask), r = (uint32_t)(x >> 40) ^ (uint32_t)x ^ (uint32_t)(i * 0x9E3779B9u);
        bool placed = false;
        for (int it = 0; it < 4000 && !placed; it++) {
            if (tab[2 * b] == EMPTY) { tab[2 * b] = x; placed = true; break; }
            if (tab[2 * b + 1] == EMPTY) { tab[2 * b + 1] = x; placed = true; break; }
            r = r * 1664525u + 1013904223u;
            std::swap(x, tab[2 * b + (r >> 31)]);
            const uint32_t b1 = bucket1(x, mask), b2 = bucket2(x, mask, cbits);
            b = (b == b1) ? b2 : b1;
        }
        if (!placed) return false;
    }
    return true;
}

// pair_arrange_kernel, one thread (the kernel's atomics give SOME order of arrival; any order must do)
static std::vector<uint64_t> arrange(const std::vector<uint64_t> &row, uint32_t buckets, int ns) {
    const uint32_t mask = buckets - 1;
    std::vector<uint64_t> slot((size_t)ns * 64, EMPTY), over;
    uint32_t cnt[32] = {0};
    for (uint64_t x : row) {
        const uint32_t r = bucket1(x, mask) & 31u, k = cnt[r]++;
        if (k < 2u * (uint32_t)ns) slot[(k >> 1) * 64 + (k & 1u) * 32 + r] = x;
        else over.push_back(x);
    }
    size_t at = 0;
    for (uint64_t x : over) {
        while (at < slot.size() && slot[at] != EMPTY) at++;
        if (at < slot.size()) slot[at++] = x;
    }
    return slot;
}

int main() {
    struct Case { uint32_t s, n; };
    for (Case c : {Case{1000, 60}, Case{1024, 20}, Case{256, 80}, Case{200, 60}, Case{12, 40}, Case{700, 40}}) {
        uint32_t buckets = 1;
        while (buckets < c.s) buckets <<= 1;
        const int ns = c.s <= 256 ? 4 : 16;
        // families of sketches sharing hashes; ragged lengths
        std::vector<std::vector<uint64_t>> rows(c.n);
        std::vector<uint64_t> pool(3 * c.s);
        for (size_t g = 0; g < c.n; g++) {
            if (g % 6 == 0) for (auto &p : pool) p = rnd64() >> 1;
            const size_t len = 1 + rnd64() % c.s;
            std::set<uint64_t> u;
            while (u.size() < len) u.insert(rnd64() % 5 < 3 ? pool[rnd64() % pool.size()] : rnd64() >> 1);
            rows[g].assign(u.begin(), u.end());
        }
        for (uint32_t cbits : {0u, 2u, 3u, 4u}) {
            const uint32_t mask = buckets - 1;
            std::vector<std::vector<uint64_t>> tabs(c.n);
            std::vector<bool> ok(c.n);
            size_t unplaced = 0;
            for (size_t g = 0; g < c.n; g++) {
                ok[g] = build_table(rows[g], buckets, cbits, tabs[g]);
                unplaced += !ok[g];
                if (!ok[g]) continue;
                for (uint64_t x : rows[g]) {   // (1) found in one of its two buckets
                    const uint32_t b1 = bucket1(x, mask), b2 = bucket2(x, mask, cbits);
                    CHECK(tabs[g][2 * b1] == x || tabs[g][2 * b1 + 1] == x || tabs[g][2 * b2] == x || tabs[g][2 * b2 + 1] == x);
                    CHECK(b1 <= mask && b2 <= mask && (buckets <= 2 || mask <= 2 * ((1u << cbits) - 1) || cbits == 0 || b1 != b2));
                    if (cbits && mask > 2 * ((1u << cbits) - 1)) CHECK(((b1 ^ b2) & ((1u << cbits) - 1)) == 0);   // the shared low bits
                }
            }
            CHECK(unplaced <= 1);   // (2e-5 .. 0 per sketch at 1 024 buckets; small tables at 4 bits keep the free choice or fail rarely)
            for (size_t g = 0; g < c.n; g++) {   // (2) permutation
                auto arr = arrange(rows[g], buckets, ns);
                std::vector<uint64_t> got;
                for (uint64_t x : arr) if (x != EMPTY) got.push_back(x);
                std::sort(got.begin(), got.end());
                CHECK(got == rows[g]);
            }
            for (size_t a = 0; a + 1 < c.n; a += 3)      // (3) tag probe >= exact == true
                for (size_t b = a + 1; b < std::min<size_t>(c.n, a + 8); b++) {
                    if (!ok[a]) continue;
                    std::vector<uint64_t> inter;
                    std::set_intersection(rows[a].begin(), rows[a].end(), rows[b].begin(), rows[b].end(), std::back_inserter(inter));
                    auto arr = arrange(rows[b], buckets, ns);
                    size_t tag_hits = 0, exact = 0;
                    for (uint64_t x : arr) {
                        if (x == EMPTY) continue;
                        const uint32_t t = tag_of(x), b1 = bucket1(x, mask), b2 = bucket2(x, mask, cbits);
                        bool th = false, eh = false;
                        for (uint32_t bk : {b1, b2})
                            for (int sl = 0; sl < 2; sl++) {
                                const uint64_t y = tabs[a][2 * bk + sl];
                                th |= (y != EMPTY && tag_of(y) == t);
                                eh |= (y == x);
                            }
                        tag_hits += th; exact += eh;
                    }
                    CHECK(exact == inter.size());
                    CHECK(tag_hits >= inter.size() && tag_hits <= inter.size() + 1);
                }
        }
    }
    if (failures) { printf("%d check(s) failed\n", failures); return 1; }
    printf("probe model ok\n");
    return 0;
}
