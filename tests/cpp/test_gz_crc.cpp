// TEST INFRASTRUCTURE (CPU suite: tests/test_abi.py).  The GF(2) arithmetic the device-side gzip path checksums its texts
// with (galah_amd/csrc/gz_common.h -- the SAME header the kernels of gz_inflate.hip compile) against zlib's crc32: every
// work-item takes the remainder of its own span with register 0, multiplies it by x^(8 * bytes behind the span), the sums
// are added and ghip_gz::crc_finish turns the total into the CRC-32 of gzip's trailer.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "../../galah_amd/csrc/gz_common.h"

int main() {
    std::mt19937_64 rng(7);
    uint32_t table[256];
    for (uint32_t b = 0; b < 256; b++) table[b] = ghip_gz::crc_byte(0, b);
    size_t cases = 0;
    for (size_t n : {0u, 1u, 3u, 4u, 5u, 255u, 2047u, 2048u, 2049u, 4096u, 65537u, 1000003u}) {
        for (size_t span : {1u, 7u, 2048u, 100000u}) {
            if (n / span > 200000) continue;
            std::vector<uint8_t> text(n);
            for (auto &c : text) c = (uint8_t)rng();
            uint32_t acc = 0;
            for (size_t from = 0; from < n; from += span) {
                const size_t m = std::min(span, n - from);
                uint32_t c = 0;
                for (size_t i = 0; i < m; i++) c = table[(c ^ text[from + i]) & 0xffu] ^ (c >> 8);
                acc ^= ghip_gz::gf_mul(ghip_gz::gf_x_pow_bytes(n - from - m), c);
            }
            const uint32_t got = ghip_gz::crc_finish(acc, n), want = (uint32_t)crc32(0L, text.data(), (uInt)n);
            if (got != want) { printf("n=%zu span=%zu: %08x, zlib says %08x\n", n, span, got, want); return 1; }
            cases++;
        }
    }
    // x^(8n) for n beyond 2^32 bytes wraps as the order of x says; spot-check the multiplication's algebra
    for (int i = 0; i < 1000; i++) {
        const uint32_t a = (uint32_t)rng(), b = (uint32_t)rng(), c = (uint32_t)rng();
        if (ghip_gz::gf_mul(a, b) != ghip_gz::gf_mul(b, a) || ghip_gz::gf_mul(ghip_gz::gf_mul(a, b), c) != ghip_gz::gf_mul(a, ghip_gz::gf_mul(b, c)) ||
            ghip_gz::gf_mul(a, b ^ c) != (ghip_gz::gf_mul(a, b) ^ ghip_gz::gf_mul(a, c)) || ghip_gz::gf_mul(a, 0x80000000u) != a) { printf("gf_mul algebra\n"); return 1; }
    }
    const uint64_t p = 123456789, q = 987654321;
    if (ghip_gz::gf_mul(ghip_gz::gf_x_pow_bytes(p), ghip_gz::gf_x_pow_bytes(q)) != ghip_gz::gf_x_pow_bytes(p + q)) { printf("x^(8(p+q))\n"); return 1; }
    printf("gz crc ok: %zu cases\n", cases);
    return 0;
}
