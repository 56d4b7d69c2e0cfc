// MurmurHash3_x64_128 of a 21-byte key whose first-stage multiplies come from tables (sketch.hip), in
// hand-written gfx950 ISA.  Shared by sketch.hip and scripts/ubench/int_ops.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// x * C mod 2^64 = three v_mad_u64_u32 (gfx950 wants even-aligned 64-bit tuples, so the cross terms are accumulated
// in their own pair) and one add of the cross terms to the high word.
//
// MurmurHash3_x64_128(canonical 21-mer, seed).h1 from the table terms A = k1*c1, B = k2*c2,
// T = rotl(tail*c1,31)*c2 ^ 21.  S1/S2/S3 are the three instructions a non-zero seed adds.
//
// Order: the hash is two independent halves (k1 -> h1, k2 -> h2) that meet three times; its 47 instructions are
// issued with the halves INTERLEAVED -- the six multiplies of a stage stand together, then the two adds, the shifts,
// the xors -- so that no instruction waits on the one before it.  Worth about 1 % (scripts/ubench/hash_variants, warm
// clocks: 205-207 SIMD-cycles per wave half after half, 204-205 interleaved, 200 with two evaluations interleaved on
// top; sketch_kmers 91.5 -> 90.2 ms per 50 Gbases): the SIMD is busy either way, every instruction of this mix costs
// about 4.3 cycles whatever its class, and the instruction count is what there is to save.
#define GHIP_MURMUR21_BODY(S1, S2, S3)                                                   \
    "v_alignbit_b32 v48, %[a0], %[a1], 1\n"                                             \
    "v_alignbit_b32 v49, %[a1], %[a0], 1\n" /* rotl(A,31) */                            \
    "v_alignbit_b32 v50, %[b1], %[b0], 31\n"                                            \
    "v_alignbit_b32 v51, %[b0], %[b1], 31\n" /* rotl(B,33) */                           \
    "v_mad_u64_u32 v[42:43], vcc, v49, %[c2lo], 0\n" /* k1 = rotl(A,31)*c2 -> v[40:41] */ \
    "v_mad_u64_u32 v[46:47], vcc, v51, %[c1lo], 0\n" /* k2 = rotl(B,33)*c1 -> v[44:45] */ \
    "v_mad_u64_u32 v[40:41], vcc, v48, %[c2lo], 0\n"                                    \
    "v_mad_u64_u32 v[44:45], vcc, v50, %[c1lo], 0\n"                                    \
    "v_mad_u64_u32 v[42:43], vcc, v48, %[c2hi], v[42:43]\n"                             \
    "v_mad_u64_u32 v[46:47], vcc, v50, %[c1hi], v[46:47]\n"                             \
    "v_add_u32 v41, v41, v42\n"                                                         \
    "v_add_u32 v45, v45, v46\n"                                                         \
    S1                                       /* h1 = seed ^ k1 */                       \
    S3                                       /* h2 = seed ^ k2 */                       \
    "v_alignbit_b32 v48, v40, v41, 5\n"                                                 \
    "v_alignbit_b32 v49, v41, v40, 5\n"     /* rotl(h1,27) */                           \
    "v_alignbit_b32 v50, v44, v45, 1\n"                                                 \
    "v_alignbit_b32 v51, v45, v44, 1\n"     /* rotl(h2,31) */                           \
    S2                                       /* h1 += h2 (= seed) */                    \
    "v_lshl_add_u64 v[48:49], v[48:49], 2, v[48:49]\n"                                  \
    "v_lshl_add_u64 v[48:49], v[48:49], 0, %[k52]\n"                                    \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, v[48:49]\n"                                  \
    "v_lshl_add_u64 v[50:51], v[50:51], 2, v[50:51]\n"                                  \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, %[k38]\n"                                    \
    "v_xor_b32 v48, v48, %[t0]\n"                                                       \
    "v_xor_b32 v49, v49, %[t1]\n"           /* h1 ^= tail ^ 21 */                       \
    "v_xor_b32 v50, 21, v50\n"              /* h2 ^= 21 */                              \
    "v_lshl_add_u64 v[48:49], v[48:49], 0, v[50:51]\n"                                  \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, v[48:49]\n"                                  \
    "v_lshrrev_b32 v52, 1, v49\n"           /* fmix64 of both, short of the last multiply: x ^= x >> 33 */ \
    "v_lshrrev_b32 v42, 1, v51\n"                                                       \
    "v_xor_b32 v48, v48, v52\n"                                                         \
    "v_xor_b32 v50, v50, v42\n"                                                         \
    "v_mad_u64_u32 v[42:43], vcc, v49, %[f1lo], 0\n" /* x *= f1 */                      \
    "v_mad_u64_u32 v[46:47], vcc, v51, %[f1lo], 0\n"                                    \
    "v_mad_u64_u32 v[40:41], vcc, v48, %[f1lo], 0\n"                                    \
    "v_mad_u64_u32 v[44:45], vcc, v50, %[f1lo], 0\n"                                    \
    "v_mad_u64_u32 v[42:43], vcc, v48, %[f1hi], v[42:43]\n"                             \
    "v_mad_u64_u32 v[46:47], vcc, v50, %[f1hi], v[46:47]\n"                             \
    "v_add_u32 v41, v41, v42\n"                                                         \
    "v_add_u32 v45, v45, v46\n"                                                         \
    "v_lshrrev_b32 v52, 1, v41\n"           /* x ^= x >> 33 */                          \
    "v_lshrrev_b32 v42, 1, v45\n"                                                       \
    "v_xor_b32 v40, v40, v52\n"             /* a = v[40:41]: fmix64(h1) short of its last multiply */ \
    "v_xor_b32 v44, v44, v42\n"             /* b = v[44:45]: the same for h2 */         \
    /* a*f2 + b*f2 = (a + b)*f2: ONE multiply gives the high word the filter needs */   \
    "v_lshl_add_u64 v[48:49], v[40:41], 0, v[44:45]\n"                                  \
    "v_mad_u64_u32 v[42:43], vcc, v49, %[f2lo], 0\n"                                    \
    "v_mad_u64_u32 v[50:51], vcc, v48, %[f2lo], 0\n"                                    \
    "v_mad_u64_u32 v[42:43], vcc, v48, %[f2hi], v[42:43]\n"                             \
    "v_add3_u32 %[s1], v51, v42, 1\n"

// The hash is h = fin(a*f2) + fin(b*f2) with fin(x) = x ^ (x >> 33), f2 = 0xc4ceb9fe1a85ec53 and a, b the two halves
// short of their last multiply.  fin() only changes the low 31 bits, so hi32(h) = hi32(a*f2) + hi32(b*f2) + carry, and
// a*f2 + b*f2 = (a + b)*f2 has the high word G = hi32(a*f2) + hi32(b*f2) + carry': hi32(h) is G - 1, G or G + 1
// (mod 2^32).  The filter therefore needs ONE 64-bit multiply instead of two: it returns s1 = G + 1 (mod 2^32), and
// h <= thr implies s1 <= murmur21_filter_bound(thr).  The two real multiplies, the xor-shifts and the 64-bit add are
// only spent on the ~1/2000 hashes that pass this one 32-bit compare (murmur21_finish).
template <bool SEED0>
__device__ __forceinline__ uint32_t murmur21_filter(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t t0,
                                                     uint32_t t1, uint32_t seed, uint64_t &A, uint64_t &B) {
    uint32_t s1;
    const uint64_t k52 = 0x52dce729ull, k38 = 0x38495ab5ull, seed64 = seed;
#define GHIP_MURMUR21_OPERANDS                                                                                       \
    [s1] "=&v"(s1), "=&{v[40:41]}"(A), "=&{v[44:45]}"(B)                                                              \
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [t0] "v"(t0), [t1] "v"(t1),                        \
        [c1lo] "s"(0x114253d5u), [c1hi] "s"(0x87c37b91u), [c2lo] "s"(0x2745937fu), [c2hi] "s"(0x4cf5ad43u),          \
        [f1lo] "s"(0xed558ccdu), [f1hi] "s"(0xff51afd7u), [f2lo] "s"(0x1a85ec53u), [f2hi] "s"(0xc4ceb9feu),          \
        [k52] "s"(k52), [k38] "s"(k38), [seed] "s"(seed), [seed64] "s"(seed64)                                       \
        : "v42", "v43", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "vcc"
    if constexpr (SEED0) {
        asm volatile(GHIP_MURMUR21_BODY("", "", "") : GHIP_MURMUR21_OPERANDS);
    } else {
        asm volatile(GHIP_MURMUR21_BODY("v_xor_b32 v40, %[seed], v40\n", "v_lshl_add_u64 v[48:49], v[48:49], 0, %[seed64]\n",
                               "v_xor_b32 v44, %[seed], v44\n")
            : GHIP_MURMUR21_OPERANDS);
    }
#undef GHIP_MURMUR21_OPERANDS
    return s1;
}

__device__ __forceinline__ uint64_t murmur21_finish(uint64_t A, uint64_t B) {
    const uint64_t F1 = A * 0xc4ceb9fe1a85ec53ull, F2 = B * 0xc4ceb9fe1a85ec53ull;
    return (F1 ^ (F1 >> 33)) + (F2 ^ (F2 >> 33));
}

// Largest s1 for which a hash <= thr is possible: hi32(h) <= hi32(thr) = t needs G in {2^32 - 1} u [0, t + 1],
// i.e. s1 = G + 1 (mod 2^32) in [0, t + 2].
__device__ __forceinline__ uint32_t murmur21_filter_bound(uint64_t thr) {
    const uint32_t th = (uint32_t)(thr >> 32);
    return th >= 0xfffffffeu ? 0xffffffffu : th + 2u;
}

template <bool SEED0>
__device__ __forceinline__ uint64_t murmur21_core(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t t0,
                                                   uint32_t t1, uint32_t seed) {
    uint64_t A, B;
    murmur21_filter<SEED0>(a0, a1, b0, b1, t0, t1, seed, A, B);
    return murmur21_finish(A, B);
}
