// The pure functions of the dense probe form that the kernels (pairs_probe.hip) and a host-side model of them
// (tests/cpp/test_probe_model.cpp, run in the CPU suite) share: bucket choices of the cuckoo sets and the 31-bit tags.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define GHIP_HD __host__ __device__ __forceinline__
#else
#define GHIP_HD inline
#endif

namespace ghip_probe {

GHIP_HD uint32_t bucket1(uint64_t x, uint32_t mask) { return (uint32_t)x & mask; }
// The arranged form (ghip_options.probe_arranged) CONSTRAINS the second cuckoo choice: it keeps the first one's low `cbits`
// bucket bits (cbits = 0: the free form).  A bucket is 8 bytes of tags, and a ds_read_b64 is served in two groups of 32 lanes
// that conflict when two lanes read different addresses of one bank PAIR = bucket mod 32: a B row whose elements are dealt to
// the lanes by the residue of their FIRST bucket (pair_arrange_kernel) reads its first buckets in one LDS cycle per group, and
// its second buckets -- 5 - cbits random bank bits left -- in ~3.4 (cbits 0), ~3 (2), ~2.7 (3), 2 (4) instead of ~3.4.
// How far the constraint can go is a matter of the cuckoo tables, simulated on the CPU (1 000-hash sketches, 2 x 1 024
// slots, 300 000 tables per form): with cbits = 4 a residue class is a cuckoo table of 64 buckets of its own and 2e-5 of
// the sketches cannot be placed (each one a fall-back of the whole call to the merge kernel); 3e-6 with cbits = 3; none
// seen with 2 or 0.  And a constrained second choice EQUAL to the first (probability 2^cbits / buckets instead of 1 /
// buckets) leaves an element one bucket -- three of those in one bucket cannot be placed (8e-4 of the sketches at cbits = 4
// until the equal case flips the lowest free bit).  Default: 3 bits for tables of >= 1 024 buckets, 2 below.
GHIP_HD uint32_t bucket2(uint64_t x, uint32_t mask, uint32_t cbits) {
    const uint32_t free2 = (uint32_t)(x >> 20) & mask;
    const uint32_t lm = (1u << cbits) - 1u;
    if (cbits == 0 || mask <= 2u * lm) return free2;   // (tables too small for the constraint keep the free choice)
    uint32_t c = (free2 & ~lm) | ((uint32_t)x & lm);
    if (c == ((uint32_t)x & mask)) c ^= lm + 1u;        // never the first bucket again
    return c;
}
// 31 bits of the hash that neither bucket index can use (bits 10..19 and 30..50; s <= 1024: at most 10-bit indices) under a
// presence bit: an empty slot's tag (0) matches nothing
GHIP_HD uint32_t tag_of(uint64_t x) {
    return 0x80000000u | (((uint32_t)(x >> 30) & 0x1fffffu) << 10) | (((uint32_t)x >> 10) & 0x3ffu);
}


}  // namespace ghip_probe
