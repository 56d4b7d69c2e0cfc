// MinHash sketching on gfx950: replaces finch::sketch_files (reference src/finch.rs:55-69).
//
//   sketch_kmers  : one pass over the normalised base stream (k = 21: sketch_kmers21_kernel, below; any other k:
//                   sketch_kmers_kernel_rt).  Each lane owns 64 consecutive k-mer starts, rolls the 2-bit
//                   codes of both strands, hashes the canonical k-mer with MurmurHash3_x64_128 (seed, first
//                   u64) and keeps hashes <= a per-genome threshold as candidates; with SEEDS the same pass
//                   emits the FracMinHash seeds of the ANI index.  VALU-issue bound (~98 instructions per
//                   base in the fused form), not HBM bound.
//   sketch_select : per genome, sort the candidates (bitonic, LDS), drop duplicates, write the
//                   s smallest into the packed u64[n][s] matrix.  Exact: the host re-runs a
//                   genome with a wider threshold / larger list if fewer than s distinct
//                   hashes survived or the list overflowed.
//   synth_genomes : counter-based synthetic genomes written straight into HBM (bench input).
#include <type_traits>

#include "ghip_internal.h"
#include "seed_common.h"
#include "murmur21_asm.h"

namespace {

using ghip_seed::base_code;

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// x * C mod 2^64 as three v_mad_u64_u32 (32x32+64): on gfx950 every VOP3 integer op issues in ~4.4
// cycles per wave64, multiplies included (scripts/ubench/int_ops.hip), so instruction COUNT is what
// matters; hipcc's default expansion is five instructions (mad64, 2x mul_lo, 2x add).
template <uint64_t C>
__device__ __forceinline__ uint64_t mulc(uint64_t x) {
    constexpr uint32_t c_lo = (uint32_t)C, c_hi = (uint32_t)(C >> 32);
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint64_t r0 = (uint64_t)lo * c_lo;
    uint64_t acc = (uint64_t)lo * c_hi + (r0 >> 32);
    acc = (uint64_t)hi * c_lo + acc;
    return (uint64_t)(uint32_t)r0 | (acc << 32);
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k = mulc<0xff51afd7ed558ccdULL>(k);
    k ^= k >> 33;
    k = mulc<0xc4ceb9fe1a85ec53ULL>(k);
    k ^= k >> 33;
    return k;
}

// 8 bases (2 bits each, base j at bits [2j+1:2j]) -> 8 ASCII bytes, base j in byte j.
__device__ __forceinline__ uint64_t expand8(uint32_t f16) {
    const uint32_t lut = 0x54474341u;  // 'A','C','G','T' in bytes 0..3
    uint32_t lo = f16 & 0xffu, hi = (f16 >> 8) & 0xffu;
    lo = (lo | (lo << 12)) & 0x000f000fu;
    lo = (lo | (lo << 6)) & 0x03030303u;
    hi = (hi | (hi << 12)) & 0x000f000fu;
    hi = (hi | (hi << 6)) & 0x03030303u;
    uint32_t alo = __builtin_amdgcn_perm(lut, lut, lo);
    uint32_t ahi = __builtin_amdgcn_perm(lut, lut, hi);
    return ((uint64_t)ahi << 32) | alo;
}

// ---- K = 21 fast path ---------------------------------------------------------------------------
// Cost model (scripts/ubench/int_ops.hip, cycles per wave64 instruction per SIMD): add/sub/and/or/xor/
// 32-bit shifts/v_bitop3/v_mov ~2.4; everything else (all multiplies, v_mad_u64_u32 ~5, 64-bit shifts,
// v_alignbit, v_lshl_add_u64, SDWA forms, compares) ~4.4.  The kernel is VALU-issue bound, so the loop
// is written to that model:
//  * first-stage multiplies of MurmurHash3 by table look-up.  k1 = bytes 0..7, k2 = bytes 8..15,
//    tail = bytes 16..20 of the canonical k-mer.  A 64-bit multiply is linear mod 2^64:
//    k1*c1 = A4(x0)*c1 + ((A4(x1)*c1) << 32) with A4(x) the 4 ASCII bytes of the 8-bit 2-bit-code x.
//    256-entry LDS tables of A4(x)*c1 and A4(x)*c2 replace the 2-bit->ASCII expansion and the two first
//    multiplies; a 1024-entry table holds the whole tail term rotl(tail*c1,31)*c2 ^ 21.  12 KiB of LDS.
//  * the rest of the hash (55 instructions) in hand-written ISA: x*C mod 2^64 is three v_mad_u64_u32 and
//    an add (hipcc emits six instructions), rotations are v_alignbit pairs, x*5+c is two v_lshl_add_u64.
//  * rolling state = the little-endian 2-bit codes of the forward strand and of the reverse complement;
//    the canonical k-mer (needletail: the lexicographically smaller strand) is the numerically smaller
//    of the two: rc_be < fwd_be  <=>  mask - fwd_le < mask - rc_le  <=>  rc_le < fwd_le.
//  * validity (no non-ACGT byte in the window) is only evaluated for the ~1/2000 hashes under the
//    threshold: per lane the position of the last bad byte, refreshed per 4-byte word under a
//    wave-uniform branch.  Device base streams hold A,C,G,T or bytes with bit 3 set ('N', '-').
struct KmerLuts {
    uint64_t c1[256];    // A4(x) * c1
    uint64_t c2[256];    // A4(x) * c2
    uint64_t tail[1024]; // rotl64(A5(x) * c1, 31) * c2 ^ 21   (the length xor of h1 folded in)
};

__device__ __forceinline__ uint32_t ascii4(uint32_t x8) {
    const uint32_t lut = 0x54474341u;
    uint32_t v = (x8 | (x8 << 12)) & 0x000f000fu;
    v = (v | (v << 6)) & 0x03030303u;
    return __builtin_amdgcn_perm(lut, lut, v);
}

// the tables do not depend on the input: built once per context into global memory, copied to LDS per block
__global__ __launch_bounds__(256) void build_kmer_luts_kernel(KmerLuts *L) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (uint32_t x = threadIdx.x; x < 256; x += blockDim.x) {
        const uint64_t a = ascii4(x);
        L->c1[x] = a * c1;
        L->c2[x] = a * c2;
    }
    for (uint32_t x = threadIdx.x; x < 1024; x += blockDim.x) {
        const uint64_t a = (uint64_t)ascii4(x & 0xffu) | ((uint64_t)((0x54474341u >> (8 * (x >> 8))) & 0xffu) << 32);
        L->tail[x] = (rotl64(a * c1, 31) * c2) ^ 21ull;
    }
}

__device__ __forceinline__ void load_kmer_luts(KmerLuts &L, const KmerLuts *__restrict__ g) {
    static_assert(sizeof(KmerLuts) % 16 == 0, "copied in 16-byte pieces");
    const uint4 *src = reinterpret_cast<const uint4 *>(g);
    uint4 *dst = reinterpret_cast<uint4 *>(&L);
    for (uint32_t i = threadIdx.x; i < sizeof(KmerLuts) / 16; i += blockDim.x) dst[i] = src[i];
}

struct Win21 {  // per-lane window state, refreshed once per 4-byte word
    uint32_t wlo, whi;  // W (48 bits): forward 2-bit codes of the last 24 bases, the oldest in bits 1:0
    uint32_t rlo, rhi;  // R: complement codes of the same bases, the NEWEST in bits 1:0 (bits >= 48 hold stale bases)
    int32_t lim21;      // last bad byte + 21: the 21-mer ending at byte b is clean iff lim21 <= b (and no bad byte of b's own word precedes it)
    int32_t lim15;      // last bad byte + seed k
};

// SEEDS = true: the same pass also emits the FracMinHash seeds of the ANI index (ani.hip), whose
// k <= 16 rolling codes are bit fields of the 21-mer codes already in registers -- one read of the
// bases, one byte decode, for both sketches.  A lane owns the 21-mers STARTING at its 64 bytes, i.e.
// ending at bytes 20..83; in the fused pass it owns the seeds ending at the same bytes (starts
// 21-k .. 84-k of its range), and the first lane of a genome adds the seeds ending at bytes k-1..19.
template <bool SEEDS, bool SEED0>
__global__ __launch_bounds__(GHIP_SKETCH_THREADS) void sketch_kmers21_kernel(
    const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ starts,
    const uint64_t *__restrict__ lens, const uint32_t *__restrict__ slot_genome,
    const uint64_t *__restrict__ slot_thr, const uint64_t *__restrict__ slot_cand_start,
    const uint32_t *__restrict__ slot_cand_cap, const ghip_sketch_work *__restrict__ work,
    uint32_t seed, uint64_t *__restrict__ cand, uint32_t *__restrict__ cand_count, ghip_seed::SeedOut so,
    const KmerLuts *__restrict__ g_luts) {
    constexpr int K = 21;
    __shared__ __attribute__((aligned(16))) KmerLuts luts;
    __shared__ __attribute__((aligned(16))) unsigned char sl_raw[SEEDS ? sizeof(ghip_seed::SeedLds) : 16];
    ghip_seed::SeedLds &sl = *reinterpret_cast<ghip_seed::SeedLds *>(sl_raw);  // only touched when SEEDS
    load_kmer_luts(luts, g_luts);
    __syncthreads();
    const ghip_sketch_work wk = work[blockIdx.x];
    const uint32_t slot = wk.slot;
    const uint32_t g = slot_genome[slot];
    const uint64_t L = lens[g];
    const uint64_t blk0 = (uint64_t)wk.chunk * GHIP_SKETCH_CHUNK;
    const uint32_t toff = threadIdx.x * GHIP_SKETCH_POS_PER_THREAD;
    const uint64_t p0 = blk0 + toff;
    const bool live = p0 < L;  // no early exit: the neighbour lane shuffles this lane's bytes
    ghip_seed::SeedBlock sb;
    uint32_t ashift = 0, sthr = 0;
    int ak = 1;
    if constexpr (SEEDS) {
        sthr = so.seed_thr[g];   // this genome's density
        sb = ghip_seed::seed_block_begin(sl, so, g, blk0);
        ashift = 2 * (K - so.k);  // the newest so.k bases of the 21-mer
        ak = (int)so.k;
    }
    const uint64_t thr = slot_thr[slot];
    const uint32_t thr_bound = murmur21_filter_bound(thr);
    const uint64_t cstart = slot_cand_start[slot];
    const uint32_t ccap = slot_cand_cap[slot];
    const uint4 *src = reinterpret_cast<const uint4 *>(bytes + starts[g] + p0);  // 16-B aligned

    // Every base is fetched from HBM once: a lane loads only its own 64 bytes; the 20 bytes that run
    // into the next lane's range come from that lane's registers (wave shuffle), and only lane 63
    // reads them from memory.  Lanes past the stream end hold 'N'.
    const uint4 NNNN = make_uint4(0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu);
    uint4 q0 = NNNN, q1 = NNNN, q2 = NNNN, q3 = NNNN;
    if (live) { q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; }
    uint4 q4;
    q4.x = __shfl_down(q0.x, 1, 64); q4.y = __shfl_down(q0.y, 1, 64); q4.z = __shfl_down(q0.z, 1, 64); q4.w = __shfl_down(q0.w, 1, 64);
    uint32_t q5x = __shfl_down(q1.x, 1, 64);
    if ((threadIdx.x & 63u) == 63u) {  // next wave's bytes (or the 'N' tail padding of the genome)
        q4 = NNNN; q5x = NNNN.x;
        if (live) { q4 = src[4]; q5x = src[5].x; }
    }

    constexpr uint32_t CAND_WAVE_CAP = 64;  // expected 64*64*2.5*s/L ~ 2 per wave at the default threshold
    __shared__ uint64_t cand_lds[GHIP_SKETCH_THREADS / 64][CAND_WAVE_CAP];
    __shared__ uint32_t cand_wave_n[GHIP_SKETCH_THREADS / 64], cand_base;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t cand_n = 0;  // wave-uniform
    Win21 st{0, 0, 0, 0, K - 1, ak - 1};   // "last bad byte" = -1
    struct Pend { uint64_t A, B, T; uint32_t ax, bx; } pend{};  // table terms of the position whose hash is pending
    // hash one position from its table terms and append it to the genome's candidate list if it is under the
    // threshold and its window holds no bad byte (b = the byte it ends at, inword = bad bytes of its own word up to b)
    auto finish = [&](const Pend &p, const int b, const uint32_t inword, const bool slow) __attribute__((always_inline)) {
        const uint32_t a1 = (uint32_t)(p.A >> 32) + p.ax, b1 = (uint32_t)(p.B >> 32) + p.bx;
        uint64_t F1, F2;  // the two halves of the hash short of their last multiply (murmur21_asm.h)
#ifdef GHIP_DBG_NOHASH  // timing experiment only: wrong results
        F1 = ((p.A ^ p.B ^ p.T) + (((uint64_t)a1 << 32) | b1)) * 0x9e3779b97f4a7c15ull; F2 = 0;
        const uint32_t s1 = (uint32_t)(F1 >> 32) + 1u;
#else
        const uint32_t s1 = murmur21_filter<SEED0>((uint32_t)p.A, a1, (uint32_t)p.B, b1, (uint32_t)p.T, (uint32_t)(p.T >> 32), seed, F1, F2);
#endif
        // rare (~2.5*s survivors per genome): wave-private LDS buffer, one global atomic per block at the end --
        // the ~300 concurrent blocks of a genome would otherwise serialise on its counter inside the hot loop
        bool hit = s1 <= thr_bound;  // necessary for h <= thr (murmur21_asm.h)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit) != 0, 0)) {   // cold: keep it out of the straight-line loop body
            const uint64_t h = murmur21_finish(F1, F2);
            hit = hit && h <= thr && (!slow || (st.lim21 <= b && inword == 0));
            const unsigned long long m = __ballot(hit);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (hit) {
                const uint32_t at = cand_n + rank;
                if (__builtin_expect(at < CAND_WAVE_CAP, 1)) cand_lds[wave][at] = h;
                else {  // wave buffer full (thresholds far above the default): straight to the global list
                    uint32_t idx = atomicAdd(&cand_count[slot], 1u);
                    if (idx < ccap) cand[cstart + idx] = h;
                }
            }
            cand_n += (uint32_t)__popcll(m);
        }
    };
    // after the position loop, by every thread of the block (contains __syncthreads)
    auto flush_candidates = [&]() {
        if ((threadIdx.x & 63u) == 0) cand_wave_n[wave] = min(cand_n, CAND_WAVE_CAP);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (uint32_t w = 0; w < GHIP_SKETCH_THREADS / 64; w++) { const uint32_t c = cand_wave_n[w]; cand_wave_n[w] = tot; tot += c; }
            cand_base = tot ? atomicAdd(&cand_count[slot], tot) : 0u;
        }
        __syncthreads();
        const uint32_t mine = min(cand_n, CAND_WAVE_CAP), base = cand_base + cand_wave_n[wave];
        for (uint32_t i = threadIdx.x & 63u; i < mine; i += 64)
            if (base + i < ccap) cand[cstart + base + i] = cand_lds[wave][i];
    };
    // One 4-byte word.  The four codes are decoded together (SWAR) and packed by two multiplies -- (t * 0x01041040) >> 24 =
    // c0 | c1<<2 | c2<<4 | c3<<6 and ((t ^ 3333) * 0x40100401) >> 24 = the complements in reverse order; the partial
    // products fall into distinct 2-bit fields, so nothing carries -- and shifted into the two 24-base windows ONCE; the
    // four 21-mers ending in this word are bit fields of the windows (one v_alignbit + one v_bfe per strand and position),
    // instead of two 64-bit shift/or/and chains per byte.  SLOW: some lane of the wave holds a non-ACGT byte in this word
    // (wave-uniform, rare): only then are the per-byte bad-byte masks evaluated; otherwise a window is clean iff
    // lim <= b, tested for the ~1/2000 hash survivors and the ~1/125 seed candidates only.
    auto word_impl = [&](const uint32_t w, const int b0, auto hash_tag, auto seed_tag, auto slow_tag, auto widx_tag, const bool seed_gate, const bool have_pend) __attribute__((always_inline)) {
        constexpr bool HASH = decltype(hash_tag)::value;
        constexpr int WIDX = decltype(widx_tag)::value;   // index of the word in its group of four (-1: a warm-up word)
        constexpr bool SEED_HERE = SEEDS && decltype(seed_tag)::value;
        constexpr bool SLOW = decltype(slow_tag)::value;
        const uint32_t t = ((w >> 1) ^ (w >> 2)) & 0x03030303u;   // A0 C1 G2 T3 per byte
        const uint32_t fm = t * 0x01041040u, rm = (t ^ 0x03030303u) * 0x40100401u;
        [[maybe_unused]] const uint32_t bad4 = w & 0x08080808u;    // bit 3: not one of A,C,G,T
        {   // W = (W >> 8) | (fpack << 40);  R = (R << 8) | rpack
            const uint32_t nwlo = __builtin_amdgcn_alignbit(st.whi, st.wlo, 8);
            const uint32_t nwhi = __builtin_amdgcn_perm(fm, st.whi, 0x0c0c0701u);   // byte 0 = whi.byte1, byte 1 = fm.byte3
            const uint32_t nrhi = __builtin_amdgcn_alignbit(st.rhi, st.rlo, 24);
            const uint32_t nrlo = __builtin_amdgcn_alignbit(st.rlo, rm, 24);
            st.wlo = nwlo; st.whi = nwhi; st.rlo = nrlo; st.rhi = nrhi;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = b0 + j;
            [[maybe_unused]] const uint32_t inword = SLOW ? (bad4 & (0xffffffffu >> (8 * (3 - j)))) : 0u;  // bad bytes 0..j of this word
            const int fs = 2 * j, rs = 6 - 2 * j;
            const uint32_t flo = fs ? __builtin_amdgcn_alignbit(st.whi, st.wlo, fs) : st.wlo;
            const uint32_t rlo = rs ? __builtin_amdgcn_alignbit(st.rhi, st.rlo, rs) : st.rlo;
            if constexpr (SEED_HERE) {
                // so.k-mer ending here: the low 2 so.k bits of rlo hold the COMPLEMENT of its forward big-endian code, those
                // of W >> (fs + ashift) the complement of its reverse complement's; their sum is the selection key
                // (seed_common.h) -- no masks, no min: add, multiply, compare
                const uint32_t wsh = (uint32_t)((((uint64_t)st.whi << 32) | st.wlo) >> (fs + ashift));
                const bool pass = seed_gate && (rlo + wsh) * so.mul < sthr;
                if (pass) {  // ~1/125 of the lanes; only the raw forward code is stored, seed_canon() finishes it in the flush
                    // validity only in the SLOW variant: the fast one runs when no lane of the wave saw a bad byte in
                    // this word or the six before it
                    if (!SLOW || (st.lim15 <= b && inword == 0)) {
#ifndef GHIP_DBG_NOAPPEND   // timing experiment only: no seeds come out
                        if constexpr (WIDX >= 0) ghip_seed::seed_mark(sl, sb, rlo, 4 * WIDX + j);
                        else {  // warm-up words (first lane of a genome): positions before the lane's 64
                            uint32_t canon, strand;
                            ghip_seed::seed_canon<true>(rlo, so.k, canon, strand);
                            ghip_seed::seed_emit_global(so, sb, canon, strand, toff + (uint32_t)(b - (ak - 1)));
                        }
#endif
                    }
                }
            }
            if constexpr (HASH) {
                // software pipeline: issue this position's five table reads, then hash the PREVIOUS position
                // (its reads were issued one step ago), so LDS latency hides behind ~240 cycles of hashing
                const uint32_t fhi = __builtin_amdgcn_ubfe(st.whi, fs, 10), rhi = __builtin_amdgcn_ubfe(st.rhi, rs, 10);
                const bool use_rc = (((uint64_t)rhi << 32) | rlo) < (((uint64_t)fhi << 32) | flo);
                const uint32_t lo = use_rc ? rlo : flo, hi = use_rc ? rhi : fhi;
                Pend cur;
#ifdef GHIP_DBG_NOLDS  // timing experiment only (scripts/sketch_variants.sh): wrong results
                cur.A = lo ^ 0x1234u; cur.B = lo + 77u; cur.T = hi >> 3; cur.ax = lo; cur.bx = hi;
#else
                cur.A = luts.c1[lo & 0xffu]; cur.B = luts.c2[(lo >> 16) & 0xffu]; cur.T = luts.tail[hi];
                cur.ax = reinterpret_cast<const uint32_t *>(luts.c1)[2 * ((lo >> 8) & 0xffu)];  // low words only
                cur.bx = reinterpret_cast<const uint32_t *>(luts.c2)[2 * (lo >> 24)];
#endif
                // the previous position is byte j-1 of this word, or byte 3 of the word before (whose bad
                // bytes are already folded into lim21)
                if (j > 0 || have_pend) finish(pend, b - 1, (SLOW && j > 0) ? (bad4 & (0xffffffffu >> (8 * (4 - j)))) : 0u, SLOW);
                pend = cur;
            }
        }
        if constexpr (SLOW) {
            if (bad4) {
                const int lastbad = b0 + 3 - (int)(__builtin_clz(bad4) >> 3);
                st.lim21 = lastbad + K;
                st.lim15 = lastbad + ak;
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // The careful variant runs for a word in which some lane of the wave holds a bad byte AND for the five words after
    // it (a 21-mer reaches 20 bytes back, and the pending hash of a word's first position belongs to the word before):
    // the fast variant never looks at lim21 / lim15.  The warm-up words are always careful (lim starts at k - 1).
    uint32_t slow_left = 0;  // wave-uniform
    auto word = [&](const uint32_t w, const int b0, auto hash_tag, auto seed_tag, auto widx_tag, const bool seed_gate, const bool have_pend, const bool careful) __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64((w & 0x08080808u) != 0)) slow_left = 7;
        if (__builtin_expect(careful || slow_left, 0)) {   // the fast variant is the fall-through
            slow_left = slow_left ? slow_left - 1 : 0;
            word_impl(w, b0, hash_tag, seed_tag, T_{}, widx_tag, seed_gate, have_pend);
        } else word_impl(w, b0, hash_tag, seed_tag, F_{}, widx_tag, seed_gate, have_pend);
    };
    using W_ = std::integral_constant<int, -1>;
    // warm-up: bytes 0..19 only fill the windows -- except in the first wave of a genome, whose lane 0
    // owns the seeds that end before byte 20
    const uint32_t warm[5] = {q0.x, q0.y, q0.z, q0.w, q1.x};
    if (SEEDS && wk.chunk == 0 && threadIdx.x < 64) {
        const bool first = threadIdx.x == 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            // bytes b < ak-1 cannot end a seed: lim15 = ak - 1 says so
            word(warm[i], 4 * i, F_{}, T_{}, W_{}, first, false, true);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 5; i++) word(warm[i], 4 * i, F_{}, F_{}, W_{}, false, false, false);   // (windows only: the fast variant unless a bad byte shows up)
    }
    uint32_t w0 = q1.y, w1 = q1.z, w2 = q1.w, w3 = q2.x;
#pragma unroll 1
    for (int v = 0; v < 4; v++) {
        const int b0 = 20 + 16 * v;
        word(w0, b0, T_{}, std::integral_constant<bool, SEEDS>{}, std::integral_constant<int, 0>{}, true, v > 0, false);
        word(w1, b0 + 4, T_{}, std::integral_constant<bool, SEEDS>{}, std::integral_constant<int, 1>{}, true, true, false);
        word(w2, b0 + 8, T_{}, std::integral_constant<bool, SEEDS>{}, std::integral_constant<int, 2>{}, true, true, false);
        word(w3, b0 + 12, T_{}, std::integral_constant<bool, SEEDS>{}, std::integral_constant<int, 3>{}, true, true, false);
        if constexpr (SEEDS) ghip_seed::seed_group_end(sb, (uint32_t)v);
        w0 = q2.y; w1 = q2.z; w2 = q2.w; w3 = q3.x;
        q2 = q3; q3 = q4; q4.x = q5x;
    }
    finish(pend, 83, 0u, true);  // the last position (ends at byte 83)
    flush_candidates();
    if constexpr (SEEDS) ghip_seed::seed_block_flush<true>(sl, so, sb, toff + (uint32_t)(K - ak), bytes + starts[g] + blk0);
}

// Generic-k fallback (k in [1,32], not 21): same algorithm, runtime k.
__device__ __forceinline__ uint64_t murmur3_h1_rt(const uint64_t (&w)[4], int K, uint32_t seed) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    int nblocks = K / 16;
    for (int b = 0; b < nblocks; b++) {
        uint64_t k1 = w[2 * b], k2 = w[2 * b + 1];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    int tail = K & 15;
    if (tail > 8) {
        uint64_t k2 = w[(2 * nblocks + 1) & 3];
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (tail > 0) {
        uint64_t k1 = w[(2 * nblocks) & 3];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

__global__ __launch_bounds__(GHIP_SKETCH_THREADS) void sketch_kmers_kernel_rt(
    const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ starts,
    const uint64_t *__restrict__ lens, const uint32_t *__restrict__ slot_genome,
    const uint64_t *__restrict__ slot_thr, const uint64_t *__restrict__ slot_cand_start,
    const uint32_t *__restrict__ slot_cand_cap, const ghip_sketch_work *__restrict__ work,
    int K, uint32_t seed, uint64_t *__restrict__ cand, uint32_t *__restrict__ cand_count) {
    const ghip_sketch_work wk = work[blockIdx.x];
    const uint32_t slot = wk.slot;
    const uint32_t g = slot_genome[slot];
    const uint64_t L = lens[g];
    const uint64_t p0 = (uint64_t)wk.chunk * GHIP_SKETCH_CHUNK + (uint64_t)threadIdx.x * GHIP_SKETCH_POS_PER_THREAD;
    if (p0 >= L) return;
    const uint64_t thr = slot_thr[slot];
    const uint64_t cstart = slot_cand_start[slot];
    const uint32_t ccap = slot_cand_cap[slot];
    const uint8_t *src = bytes + starts[g] + p0;
    const uint64_t mask = (K < 32) ? ((1ull << (2 * K)) - 1) : ~0ull;
    uint64_t fwd_be = 0, fwd_le = 0;
    uint32_t good = 0;
    const int NB = GHIP_SKETCH_POS_PER_THREAD + K - 1;
    for (int b = 0; b < NB; b++) {
        uint32_t code = base_code(src[b]);
        if (code > 3u) { good = 0; continue; }
        fwd_be = ((fwd_be << 2) | code) & mask;
        fwd_le = (fwd_le >> 2) | ((uint64_t)code << (2 * (K - 1)));
        good++;
        if (b < K - 1 || good < (uint32_t)K) continue;
        if (p0 + (uint64_t)(b - (K - 1)) + K > L) continue;
        uint64_t rc_be = (~fwd_le) & mask;
        uint64_t canon_le = (rc_be < fwd_be) ? ((~fwd_be) & mask) : fwd_le;
        uint64_t w[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            int nb = K - 8 * i;
            if (nb <= 0) break;
            uint64_t x = expand8((uint32_t)(canon_le >> (16 * i)) & 0xffffu);
            if (nb < 8) x &= (1ull << (8 * nb)) - 1;
            w[i] = x;
        }
        uint64_t h = murmur3_h1_rt(w, K, seed);
        if (h <= thr) {
            uint32_t idx = atomicAdd(&cand_count[slot], 1u);
            if (idx < ccap) cand[cstart + idx] = h;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// sketch_select: one 1024-thread block per pending genome.
// status bits: 1 = candidate list overflowed (count > cap), 2 = fewer than s distinct hashes.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t SELECT_THREADS = 1024;
constexpr uint32_t SELECT_LDS_ELEMS = 8192;  // 64 KiB

__device__ __forceinline__ void bitonic_sort(uint64_t *buf, uint32_t m /* pow2 */) {
    for (uint32_t size = 2; size <= m; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                uint32_t lo = 2 * t - (t & (stride - 1));
                uint32_t hi = lo + stride;
                bool up = ((lo & size) == 0);
                uint64_t a = buf[lo], b = buf[hi];
                if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(SELECT_THREADS) void sketch_select_kernel(
    const uint32_t *__restrict__ slot_genome, uint64_t *__restrict__ cand,
    const uint32_t *__restrict__ cand_count, const uint64_t *__restrict__ slot_cand_start,
    const uint32_t *__restrict__ slot_cand_cap, uint32_t s, uint64_t *__restrict__ hashes,
    uint32_t *__restrict__ out_lens, uint32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *lds = reinterpret_cast<uint64_t *>(smem_raw);
    __shared__ uint32_t wave_tot[SELECT_THREADS / 64];

    const uint32_t slot = blockIdx.x;
    const uint32_t g = slot_genome[slot];
    const uint32_t cap = slot_cand_cap[slot];
    const uint32_t count = cand_count[slot];
    uint64_t *gbuf = cand + slot_cand_start[slot];
    if (count > cap) {  // host re-runs this genome with cap >= count
        if (threadIdx.x == 0) status[slot] = 1u;
        return;
    }
    uint32_t m = 1;
    while (m < count) m <<= 1;
    if (m < 2) m = 2;
    uint64_t *buf;
    if (m <= SELECT_LDS_ELEMS) {
        buf = lds;
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) buf[i] = (i < count) ? gbuf[i] : ~0ull;
    } else {
        buf = gbuf;  // cap is a power of two >= m on this path
        for (uint32_t i = count + threadIdx.x; i < m; i += blockDim.x) buf[i] = ~0ull;
    }
    bitonic_sort(buf, m);

    // distinct rank of every element; first s distinct go to the output row
    const uint32_t per = (m + blockDim.x - 1) / blockDim.x;
    const uint32_t i0 = threadIdx.x * per;
    uint32_t local = 0;
    for (uint32_t i = i0; i < i0 + per && i < count; i++) local += (i == 0 || buf[i] != buf[i - 1]) ? 1u : 0u;
    // block exclusive scan of `local`
    uint32_t incl = local;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; w++) {
        uint32_t t = wave_tot[w];
        if (w < wave) base += t;
        total += t;
    }
    uint32_t rank = base + incl - local;
    uint64_t *row = hashes + (uint64_t)g * s;
    for (uint32_t i = i0; i < i0 + per && i < count; i++) {
        if (i == 0 || buf[i] != buf[i - 1]) {
            if (rank < s) row[rank] = buf[i];
            rank++;
        }
    }
    const uint32_t len = total < s ? total : s;
    for (uint32_t i = len + threadIdx.x; i < s; i += blockDim.x) row[i] = ~0ull;
    if (threadIdx.x == 0) {
        out_lens[g] = len;
        status[slot] = (total < s) ? 2u : 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// synth_genomes: same definition as oracle go_synth_genome (build-defined).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t synth_key(uint64_t seed, uint32_t species, uint32_t stream) {
    return splitmix64(splitmix64(seed ^ ((uint64_t)species << 20)) ^ ((uint64_t)stream * 0xd1b54a32d192ed03ULL));
}

// each thread writes 16 bases (one 16-B store)
__global__ __launch_bounds__(256) void synth_genomes_kernel(uint8_t *__restrict__ bytes,
                                                            const uint64_t *__restrict__ starts,
                                                            uint64_t length, uint32_t first,
                                                            uint32_t members, uint64_t seed,
                                                            uint32_t sub_thr) {
    const uint32_t g = blockIdx.y;  // local index; the series index is first + g
    const uint32_t species = (first + g) / members, member = (first + g) % members;
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (p0 >= length) return;
    const uint64_t ka = synth_key(seed, species, 0), km = synth_key(seed, species, member + 1);
    const uint64_t w = splitmix64(ka + (p0 >> 5));
    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint64_t p = p0 + j;
        uint32_t base = (uint32_t)(w >> (2 * (p & 31))) & 3u;
        uint64_t u = splitmix64(km + p);
        if ((uint32_t)(u >> 32) < sub_thr) base = (base + 1 + (uint32_t)((u >> 8) % 3)) & 3u;
        uint32_t ch = (0x54474341u >> (8 * base)) & 0xffu;
        if (p >= length) ch = 'N';
        out[j >> 2] |= ch << (8 * (j & 3));
    }
    *reinterpret_cast<uint4 *>(bytes + starts[g] + p0) = make_uint4(out[0], out[1], out[2], out[3]);
}

// Device base streams hold A,C,G,T or a byte with bit 3 set (what sketch_kmers21 tests): caller-supplied
// streams (ghip_genomes_from_host) are rewritten in place, every other byte becomes 'N'.  16 bytes per thread.
__global__ __launch_bounds__(256) void sanitize_bases_kernel(uint4 *__restrict__ bytes, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = bytes[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t out = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t c = (w[k] >> (8 * j)) & 0xffu;
                const bool ok = c == 'A' || c == 'C' || c == 'G' || c == 'T';
                out |= (ok ? c : (uint32_t)'N') << (8 * j);
            }
            w[k] = out;
        }
        bytes[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// The sketch pass's own roof, measurable on any box: nothing but the filter form of MurmurHash3_x64_128 (the 47
// instructions that cannot be tabulated, murmur21_asm.h), `iters` dependent evaluations per wave, 8 waves per SIMD.
__global__ __launch_bounds__(256) void hash_floor_kernel(uint32_t *__restrict__ out, uint32_t iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = seed * 3u + blockIdx.x, b0 = a0 ^ 0x1234567u, b1 = a1 + 99u, t0 = a0 * 7u, t1 = a1 * 11u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        uint64_t A, B;
        const uint32_t s1 = murmur21_filter<true>(a0, a1, b0, b1, t0, t1, 0, A, B);
        acc += s1; a0 = (uint32_t)A; a1 ^= (uint32_t)B;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

}  // namespace

// Runs the hash-only kernel for `wave_positions` wave-level hash evaluations spread over every SIMD; *ms = its duration.
int ghip_launch_hash_floor(ghip_ctx *ctx, uint64_t wave_positions, double *ms) {
    const unsigned blocks = (unsigned)ctx->num_cus * 8u;   // 256 threads = 4 waves: 8 waves per SIMD
    const uint64_t waves = (uint64_t)blocks * 4u;
    const uint32_t iters = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(wave_positions / waves, 1), 1u << 30);
    uint32_t *d_out = (uint32_t *)ghip_pool_alloc(ctx, (size_t)blocks * 256 * sizeof(uint32_t));
    if (!d_out) return GHIP_EHIP;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(hash_floor_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_out, std::min(iters, 1000u), 1u);  // warm-up
    hipEventRecord(e0, ctx->stream);
    hipLaunchKernelGGL(hash_floor_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_out, iters, 2u);
    hipEventRecord(e1, ctx->stream);
    float t = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&t, e0, e1) == hipSuccess;
    hipEventDestroy(e0); hipEventDestroy(e1);
    ghip_pool_free(ctx, d_out);
    if (!ok) return ghip_set_error(ctx, GHIP_EHIP, "hash_floor kernel failed");
    // scale to exactly wave_positions (iters was rounded down to a whole number per wave)
    *ms = (double)t * (double)wave_positions / ((double)iters * (double)waves);
    return GHIP_OK;
}

void ghip_launch_synth(ghip_ctx *ctx, uint8_t *d_bytes, const uint64_t *d_starts, uint64_t length,
                       uint32_t first, uint32_t count, uint32_t members, uint64_t seed, uint32_t sub_thr) {
    uint64_t threads = (length + 15) / 16;
    dim3 grid((unsigned)((threads + 255) / 256), count);
    ghip_prof_begin(ctx, "synth_genomes");
    hipLaunchKernelGGL(synth_genomes_kernel, grid, dim3(256), 0, ctx->stream, d_bytes, d_starts, length, first, members, seed, sub_thr);
    ghip_prof_end(ctx);
}

// ---- packed ingest: files cross PCIe as 2-bit codes (a quarter of the bytes), the resident stream stays one byte per
// base.  packed[i / 4] holds base i in bits 2 (i % 4) .. +1 (A0 C1 G2 T3, what ingest.cpp: ghip_pack_stream writes);
// every stream byte that is not A/C/G/T (the 'N' after each record, ambiguity codes, gaps) travels as a run
// (start, length, byte) and is patched in afterwards.  Bytes from `len` to the end of the last 16-byte group get 'N',
// which is what the whole buffer was filled with.
__global__ __launch_bounds__(256) void unpack_bases_kernel(const uint32_t *__restrict__ packed, uint4 *__restrict__ out, uint64_t len) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // 16 bases per thread
    if (16 * t >= len) return;
    const uint32_t w = packed[t];
    uint32_t a[4] = {ascii4(w & 0xffu), ascii4((w >> 8) & 0xffu), ascii4((w >> 16) & 0xffu), ascii4(w >> 24)};
    const uint64_t left = len - 16 * t;
    if (left < 16) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int have = (int)left - 4 * q;   // valid bytes of word q
            if (have <= 0) a[q] = 0x4e4e4e4eu;
            else if (have < 4) a[q] = (a[q] & (0xffffffffu >> (8 * (4 - have)))) | (0x4e4e4e4eu << (8 * have));
        }
    }
    out[t] = make_uint4(a[0], a[1], a[2], a[3]);
}

__global__ __launch_bounds__(64) void patch_runs_kernel(const uint32_t *__restrict__ runs /* start, len, byte */, uint8_t *__restrict__ out) {
    const uint32_t start = runs[3 * blockIdx.x], n = runs[3 * blockIdx.x + 1];
    const uint8_t b = (uint8_t)runs[3 * blockIdx.x + 2];
    for (uint32_t k = threadIdx.x; k < n; k += 64) out[(uint64_t)start + k] = b;
}

// both on `stream`, in this order, after the copy that brought `d_packed` and `d_runs`
void ghip_launch_unpack_bases(hipStream_t stream, const uint8_t *d_packed, const uint32_t *d_runs, uint32_t n_runs, uint8_t *d_out, uint64_t len) {
    if (len == 0) return;
    const uint64_t groups = (len + 15) / 16;
    hipLaunchKernelGGL(unpack_bases_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(d_packed), reinterpret_cast<uint4 *>(d_out), len);
    if (n_runs) hipLaunchKernelGGL(patch_runs_kernel, dim3(n_runs), dim3(64), 0, stream, d_runs, d_out);
}

void ghip_launch_sanitize(ghip_ctx *ctx, uint8_t *d_bytes, uint64_t n_bytes /* multiple of 16 */) {
    const uint64_t n16 = n_bytes / 16;
    if (n16 == 0) return;
    const unsigned grid = (unsigned)std::min<uint64_t>((n16 + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(sanitize_bases_kernel, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<uint4 *>(d_bytes), n16);
}

// timing experiment only (scripts/sketch_variants.sh): unused dynamic LDS caps the workgroups per CU
static unsigned dbg_extra_lds() {
    static const unsigned v = [] { const char *e = getenv("GHIP_DBG_EXTRA_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
    return v;
}

void ghip_launch_sketch_kmers(ghip_ctx *ctx, const uint8_t *d_bytes, const uint64_t *d_starts,
                              const uint64_t *d_lens, const uint32_t *d_slot_genome,
                              const uint64_t *d_slot_thr, const uint64_t *d_slot_cand_start,
                              const uint32_t *d_slot_cand_cap, const ghip_sketch_work *d_work,
                              size_t n_work, uint32_t k, uint32_t seed, uint64_t *d_cand,
                              uint32_t *d_cand_count, const ghip_seed_args *seeds) {
    if (n_work == 0) return;
    ghip_seed::SeedOut so{};
    if (seeds)
        so = ghip_seed::SeedOut{seeds->k, ghip_seed::seed_mul(seeds->k), seeds->chunk, seeds->d_seed_thr, ghip_seed::seed_chunk_magic(seeds->chunk), seeds->d_seed_code, seeds->d_seed_loc,
                                seeds->d_seed_start, seeds->d_seg_count, seeds->d_chunk_total, seeds->d_chunk_start};
    const KmerLuts *luts = nullptr;
    if (k == 21) {
        if (!ctx->d_kmer_luts) {
            ctx->d_kmer_luts = (uint64_t *)ghip_pool_alloc(ctx, sizeof(KmerLuts));
            if (!ctx->d_kmer_luts) return;  // ctx->err is set; the caller's next HIP check reports it
            hipLaunchKernelGGL(build_kmer_luts_kernel, dim3(1), dim3(256), 0, ctx->stream, reinterpret_cast<KmerLuts *>(ctx->d_kmer_luts));
        }
        luts = reinterpret_cast<const KmerLuts *>(ctx->d_kmer_luts);
    }
    ghip_prof_begin(ctx, "sketch_kmers");
    for (size_t off = 0; off < n_work; off += GHIP_MAX_GRID) {  // one AQL dispatch holds < 2^32 work-items
        const unsigned grid = (unsigned)std::min<size_t>(n_work - off, GHIP_MAX_GRID);
        if (k == 21) {  // seeds != nullptr: fused MinHash + ANI seeding pass (first pass over all genomes only)
#define GHIP_LAUNCH21(SEEDS, SEED0)                                                                                  \
    hipLaunchKernelGGL((sketch_kmers21_kernel<SEEDS, SEED0>), dim3(grid), dim3(GHIP_SKETCH_THREADS), dbg_extra_lds(), ctx->stream, \
                       d_bytes, d_starts, d_lens, d_slot_genome, d_slot_thr, d_slot_cand_start, d_slot_cand_cap,     \
                       d_work + off, seed, d_cand, d_cand_count, so, luts)
            if (seeds && seed == 0) GHIP_LAUNCH21(true, true);
            else if (seeds) GHIP_LAUNCH21(true, false);
            else if (seed == 0) GHIP_LAUNCH21(false, true);
            else GHIP_LAUNCH21(false, false);
#undef GHIP_LAUNCH21
        } else {
            hipLaunchKernelGGL(sketch_kmers_kernel_rt, dim3(grid), dim3(GHIP_SKETCH_THREADS), 0, ctx->stream,
                               d_bytes, d_starts, d_lens, d_slot_genome, d_slot_thr, d_slot_cand_start,
                               d_slot_cand_cap, d_work + off, (int)k, seed, d_cand, d_cand_count);
        }
    }
    ghip_prof_end(ctx);
}

void ghip_launch_sketch_select(ghip_ctx *ctx, const uint32_t *d_slot_genome, size_t n_slots,
                               uint64_t *d_cand, const uint32_t *d_cand_count,
                               const uint64_t *d_slot_cand_start, const uint32_t *d_slot_cand_cap,
                               uint32_t s, uint64_t *d_hashes, uint32_t *d_lens, uint32_t *d_status) {
    if (n_slots == 0) return;
    ghip_prof_begin(ctx, "sketch_select");
    hipLaunchKernelGGL(sketch_select_kernel, dim3((unsigned)n_slots), dim3(SELECT_THREADS),
                       SELECT_LDS_ELEMS * sizeof(uint64_t), ctx->stream, d_slot_genome, d_cand, d_cand_count,
                       d_slot_cand_start, d_slot_cand_cap, s, d_hashes, d_lens, d_status);
    ghip_prof_end(ctx);
}
