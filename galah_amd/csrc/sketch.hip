// MinHash sketching on gfx950: replaces finch::sketch_files (reference src/finch.rs:55-69).
//
//   sketch_kmers  : one pass over the resident bases -- 2-bit codes + a validity bitmap, 3 bits per base (k = 21:
//                   sketch_kmers21_kernel, below; any other k: sketch_kmers_kernel_rt).  Each lane owns 64 consecutive
//                   k-mer starts; the codes of both strands are bit fields of the packed words, the canonical k-mer is
//                   hashed with MurmurHash3_x64_128 (seed, first u64) and hashes <= a per-genome threshold are kept as
//                   candidates; with SEEDS the same pass emits the FracMinHash seeds of the ANI index.  VALU-issue
//                   bound (MurmurHash3 alone is 47 instructions per base), not HBM bound.
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

// ---- resident base format (ghip_genomes) ------------------------------------------------------
// packed[i / 16] holds base i in bits 2 (i % 16) .. +1 (A0 C1 G2 T3: the form the files cross PCIe in, ingest.cpp);
// valid[i / 32] bit i % 32 says whether stream byte i is one of A/C/G/T (the 'N' after a record, ambiguity codes, gaps and
// everything past the genome's length are 0).  3 bits per base instead of 8: 50 000 x 5 Mb = 94 GB resident on one GPU.
// A genome starts at a base offset that is a multiple of 64 (16-byte packed words, 8-byte bitmap words) and is followed
// by at least 128 invalid positions (vector loads over-read).
__device__ __forceinline__ uint32_t packed_base(const uint32_t *__restrict__ packed, uint64_t i) { return (packed[i >> 4] >> (2u * ((uint32_t)i & 15u))) & 3u; }
__device__ __forceinline__ bool valid_base(const uint32_t *__restrict__ valid, uint64_t i) { return (valid[i >> 5] >> ((uint32_t)i & 31u)) & 1u; }

// the 16 bases of a packed word complemented and in reverse order (base 15 in bits 1:0)
__device__ __forceinline__ uint32_t revcomp16(uint32_t w) {
    const uint32_t x = __brev(~w);   // bases reversed, the two bits of each base swapped
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// SEEDS = true: the same pass also emits the FracMinHash seeds of the ANI index (ani.hip), whose
// k <= 16 codes are bit fields of the packed words already in registers -- one read of the bases for both sketches.
// A lane owns the 21-mers STARTING at its 64 bases, i.e. ending at bases 20..83; in the fused pass it owns the seeds
// ending at the same bases (starts 21-k .. 84-k of its range), and the first lane of a genome adds the seeds that end
// at bases k-1..19.
//
// The stream is consumed as it is stored: a lane loads ITS 64 bases as one 16-byte vector (and 8 bytes of validity);
// the 20 bases that run into the next lane's range come from that lane's registers.  The forward strand's little-endian
// code of the 21-mer starting at base p IS the packed stream shifted right by 2p; the reverse complement's is the
// word-wise reversed complement of the stream (revcomp16, once per 16 bases) shifted likewise -- one v_alignbit per
// strand and position for the low 32 bits, one v_bfe (v_alignbit + v_and in four of sixteen positions) for the 10 bits
// above them; no per-base decode, no rolling state, no warm-up.  Validity is per 16 positions: when some lane of the
// wave holds an invalid position among the 36 bases an iteration touches, the iteration runs the careful variant
// (per-position window tests); otherwise the fast one, which contains no validity test at all.
template <bool SEEDS, bool SEED0>
__global__ __launch_bounds__(GHIP_SKETCH_THREADS) void sketch_kmers21_kernel(
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ valid, const uint64_t *__restrict__ starts,
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
    const bool live = p0 < L;  // no early exit: the neighbour lane shuffles this lane's words
    ghip_seed::SeedBlock sb;
    uint32_t ashift = 0, sthr = 0;
    int ak = 1;
    if constexpr (SEEDS) {
        sthr = so.seed_thr[g];   // this genome's density
        sb = ghip_seed::seed_block_begin(sl, so, g, blk0);
        ashift = 2 * (K - so.k);  // the seed is the newest so.k bases of the 21-mer
        ak = (int)so.k;
    }
    const uint64_t thr = slot_thr[slot];
    const uint32_t thr_bound = murmur21_filter_bound(thr);
    const uint64_t cstart = slot_cand_start[slot];
    const uint32_t ccap = slot_cand_cap[slot];
    const uint64_t base0 = starts[g] + p0;   // a multiple of 64
    const uint32_t *pw = packed + (base0 >> 4), *vw = valid + (base0 >> 5);

    // Every base is fetched from HBM once: a lane loads only its own 64 bases; the 20 that run into the next lane's
    // range come from that lane's registers (wave shuffle), and only lane 63 reads them from memory.  Lanes past the
    // stream end hold invalid positions.
    uint4 q = make_uint4(0, 0, 0, 0);
    uint2 vb = make_uint2(0, 0);
    if (live) { q = *reinterpret_cast<const uint4 *>(pw); vb = *reinterpret_cast<const uint2 *>(vw); }
    uint32_t o0 = __shfl_down(q.x, 1, 64), o1 = __shfl_down(q.y, 1, 64), v2 = __shfl_down(vb.x, 1, 64);
    if ((threadIdx.x & 63u) == 63u) {  // next wave's bases (or the invalid tail of the genome)
        o0 = o1 = v2 = 0;
        if (live) { o0 = pw[4]; o1 = pw[5]; v2 = vw[2]; }
    }

    constexpr uint32_t CAND_WAVE_CAP = 64;  // expected 64*64*2.5*s/L ~ 2 per wave at the default threshold
    __shared__ uint64_t cand_lds[GHIP_SKETCH_THREADS / 64][CAND_WAVE_CAP];
    __shared__ uint32_t cand_wave_n[GHIP_SKETCH_THREADS / 64], cand_base;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t cand_n = 0;  // wave-uniform
    struct Pend { uint64_t A, B, T; uint32_t ax, bx; } pend{};  // table terms of the position whose hash is pending
    bool pend_ok = true;   // ... and whether its window holds only valid bases (maintained by the careful variant)
    // hash one position from its table terms and append it to the genome's candidate list if it is under the
    // threshold and its window is clean
    auto finish = [&](const Pend &p, const bool ok) __attribute__((always_inline)) {
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
            hit = hit && h <= thr && ok;
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
    // Sixteen positions: the 21-mers starting at bases 16 v + j of the lane, j = 0..15.  W0..W2 = the packed words holding
    // bases 16 v .. 16 v + 47, R0..R2 their reversed complements, vlo/vhi = the validity bits of bases 16 v .. 16 v + 63.
    auto body16 = [&](const uint32_t W0, const uint32_t W1, const uint32_t W2, const uint32_t R0, const uint32_t R1, const uint32_t R2,
                      const uint32_t vlo, const uint32_t vhi, auto slow_tag, const bool have_pend) __attribute__((always_inline)) {
        constexpr bool SLOW = decltype(slow_tag)::value;
        // the forward stream from the seed's first base on: {W2, W1, W0} >> ashift (run-time seed length, uniform)
        [[maybe_unused]] uint32_t S0 = 0, S1 = 0;
        if constexpr (SEEDS) {
            const uint32_t bs = ashift & 31u;
            if (ashift < 32) { S0 = __builtin_amdgcn_alignbit(W1, W0, bs); S1 = __builtin_amdgcn_alignbit(W2, W1, bs); }
            else { S0 = __builtin_amdgcn_alignbit(W2, W1, bs); S1 = W2 >> bs; }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t flo = j ? __builtin_amdgcn_alignbit(W1, W0, 2 * j) : W0;
            uint32_t rlo;
            if (j < 11) rlo = __builtin_amdgcn_alignbit(R0, R1, 2 * (11 - j));
            else if (j == 11) rlo = R1;
            else rlo = __builtin_amdgcn_alignbit(R1, R2, 2 * (27 - j));
            [[maybe_unused]] bool ok21 = true;
            if constexpr (SLOW) {   // bases j .. j + 20 of the window all valid
                const uint32_t vv = j ? __builtin_amdgcn_alignbit(vhi, vlo, j) : vlo;
                ok21 = (vv & 0x1fffffu) == 0x1fffffu;
            }
            if constexpr (SEEDS) {
                // so.k-mer ending here: the low 2 so.k bits of rlo hold the COMPLEMENT of its forward big-endian code, those
                // of the forward stream from its first base on the complement of its reverse complement's; their sum is the
                // selection key (seed_common.h) -- no masks, no min: add, multiply, compare
                const uint32_t wsh = j ? __builtin_amdgcn_alignbit(S1, S0, 2 * j) : S0;
                const bool pass = (rlo + wsh) * so.mul < sthr;
                if (pass) {  // ~1/125 of the lanes; only the raw forward code is stored, seed_canon() finishes it in the flush
                    bool okk = true;
                    if constexpr (SLOW) {   // bases j + 21 - k .. j + 20
                        const uint32_t sh = (uint32_t)(j + K - ak);
                        const uint64_t vv = ((((uint64_t)vhi << 32) | vlo) >> sh);
                        const uint32_t mk = ak >= 32 ? ~0u : ((1u << ak) - 1u);
                        okk = ((uint32_t)vv & mk) == mk;
                    }
#ifndef GHIP_DBG_NOAPPEND   // timing experiment only: no seeds come out
                    if (okk) ghip_seed::seed_mark(sl, sb, rlo, j);
#endif
                }
            }
            {
                // software pipeline: issue this position's five table reads, then hash the PREVIOUS position
                // (its reads were issued one step ago), so LDS latency hides behind ~240 cycles of hashing
                uint32_t fhi, rhi;
                if (j <= 11) { fhi = __builtin_amdgcn_ubfe(W1, 2 * j, 10); rhi = __builtin_amdgcn_ubfe(R0, 2 * (11 - j), 10); }
                else { fhi = __builtin_amdgcn_alignbit(W2, W1, 2 * j) & 0x3ffu; rhi = __builtin_amdgcn_alignbit(R0, R1, 2 * (27 - j)) & 0x3ffu; }
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
                if (j > 0 || have_pend) finish(pend, pend_ok);
                pend = cur;
                pend_ok = SLOW ? ok21 : true;
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (SEEDS && wk.chunk == 0 && threadIdx.x == 0) {
        // the seeds of a genome that end before base 20 belong to no lane's 64 positions: its first lane emits them (cold)
        const uint64_t gb = starts[g];
        for (int e = ak - 1; e < K - 1 && (uint64_t)e < L; e++) {
            uint32_t f = 0;
            bool okk = true;
            for (int i = e - ak + 1; i <= e; i++) { f = (f << 2) | packed_base(packed, gb + (uint64_t)i); okk = okk && valid_base(valid, gb + (uint64_t)i); }
            if (!okk) continue;
            const uint32_t r = ghip_seed::revcomp_code(f, so.k);
            if (ghip_seed::seed_selected(f, r, so.mul, sthr)) ghip_seed::seed_emit_global(so, sb, min(f, r), r < f ? 1u : 0u, (uint32_t)(e - ak + 1));
        }
    }
    uint32_t W0 = q.x, W1 = q.y, W2 = q.z, n0 = q.w, n1 = o0, n2 = o1;
    uint32_t R0 = revcomp16(W0), R1 = revcomp16(W1), R2 = revcomp16(W2);
    uint32_t va = vb.x, vbb = vb.y, vc = v2;   // validity of bases 16 v .. 16 v + 95 (shifted down 16 bits per iteration)
#pragma unroll 1
    for (int v = 0; v < 4; v++) {
        // clean = the 36 bases this iteration touches are all valid, in every lane of the wave
        const bool dirty = va != 0xffffffffu || (vbb & 0xfu) != 0xfu;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(dirty) != 0, 0)) body16(W0, W1, W2, R0, R1, R2, va, vbb, T_{}, v > 0);   // the fast variant is the fall-through
        else body16(W0, W1, W2, R0, R1, R2, va, vbb, F_{}, v > 0);
        if constexpr (SEEDS) ghip_seed::seed_group_end(sb, (uint32_t)v);
        W0 = W1; W1 = W2; W2 = n0; n0 = n1; n1 = n2;
        R0 = R1; R1 = R2; R2 = revcomp16(W2);
        va = __builtin_amdgcn_alignbit(vbb, va, 16); vbb = __builtin_amdgcn_alignbit(vc, vbb, 16); vc >>= 16;
    }
    finish(pend, pend_ok);  // the last position (ends at base 83)
    flush_candidates();
    if constexpr (SEEDS) ghip_seed::seed_block_flush<true>(sl, so, sb, toff + (uint32_t)(K - ak), packed + ((starts[g] + blk0) >> 4));
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
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ valid, const uint64_t *__restrict__ starts,
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
    const uint64_t base0 = starts[g] + p0;
    const uint64_t mask = (K < 32) ? ((1ull << (2 * K)) - 1) : ~0ull;
    uint64_t fwd_be = 0, fwd_le = 0;
    uint32_t good = 0;
    const int NB = GHIP_SKETCH_POS_PER_THREAD + K - 1;
    for (int b = 0; b < NB; b++) {
        if (!valid_base(valid, base0 + (uint64_t)b)) { good = 0; continue; }   // (positions past the genome's length are invalid)
        const uint32_t code = packed_base(packed, base0 + (uint64_t)b);
        fwd_be = ((fwd_be << 2) | code) & mask;
        fwd_le = (fwd_le >> 2) | ((uint64_t)code << (2 * (K - 1)));
        good++;
        if (b < K - 1 || good < (uint32_t)K) continue;
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
constexpr uint32_t SELECT_THREADS = 256;     // small blocks: eight per CU overlap one another's loads and barriers (1 024-thread blocks: two)
constexpr uint32_t SELECT_LDS_ELEMS = 8192;  // 64 KiB at most; a launch asks for what its longest list needs
constexpr uint32_t SELECT_BUCKETS = 2048, SELECT_PER_THREAD = 16, SELECT_BUCKET_MAX = 12;   // counting-sort form of the selection

// Thread t of a step exchanges elements lo = 2t - (t & (stride - 1)) and lo + stride: for stride <= 64 the 64 threads of a
// wavefront stay inside ITS 128 consecutive elements, so consecutive steps of small strides only need the wave's own
// writes to be visible -- a wavefront fence, not a block barrier.  Of the 66 steps of a 2 048-element sort 14 need the block
// (a stride above 64, or the step after one): the kernel was barrier-bound (16 SIMD-cycles per VALU instruction).
// LOCAL = the buffer is LDS (the global-memory path of very long candidate lists keeps the block barrier everywhere).
template <bool LOCAL>
__device__ __forceinline__ void bitonic_sort(uint64_t *buf, uint32_t m /* pow2 */) {
    uint32_t prev = ~0u;   // the fill before the sort was block-wide
    for (uint32_t size = 2; size <= m; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            if (!LOCAL || stride > 64 || prev > 64) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
            prev = stride;
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
    const uint32_t *__restrict__ slot_cand_cap, uint32_t s, uint32_t lds_elems, uint64_t *__restrict__ hashes,
    uint32_t *__restrict__ out_lens, uint32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // lds_elems x 8 bytes
    uint64_t *lds = reinterpret_cast<uint64_t *>(smem_raw);
    __shared__ uint32_t wave_tot[SELECT_THREADS / 64];
    __shared__ uint64_t wave_max[SELECT_THREADS / 64];
    __shared__ uint32_t worst;   // largest bucket of the counting sort if above SELECT_BUCKET_MAX, else 0

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
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (m <= lds_elems && count <= SELECT_PER_THREAD * SELECT_THREADS) {
        // The candidates are hashes below a threshold, i.e. spread evenly over [0, max]: a counting sort on their top 11
        // bits puts every one within a few places of its final position in ONE pass (histogram by LDS atomic, scan,
        // scatter), and the handful that share a bucket are ordered by insertion -- about a fifth of the LDS traffic of
        // the 66-step bitonic network, which remains for inputs that are not spread (a bucket of more than
        // SELECT_BUCKET_MAX: highly repetitive sequence) and for lists beyond the LDS.
        buf = lds;
        uint32_t *hist = reinterpret_cast<uint32_t *>(smem_raw + (size_t)lds_elems * sizeof(uint64_t));   // [SELECT_BUCKETS]
        uint64_t v[SELECT_PER_THREAD];
        uint32_t rk[SELECT_PER_THREAD];
        uint64_t vmax = 0;
#pragma unroll
        for (uint32_t u = 0; u < SELECT_PER_THREAD; u++) {
            const uint32_t i = threadIdx.x + u * SELECT_THREADS;
            v[u] = i < count ? gbuf[i] : 0ull;
            vmax = v[u] > vmax ? v[u] : vmax;
        }
        for (uint32_t b = threadIdx.x; b < SELECT_BUCKETS; b += SELECT_THREADS) hist[b] = 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint64_t o = __shfl_xor(vmax, off, 64); vmax = o > vmax ? o : vmax; }
        if (lane == 0) wave_max[wave] = vmax;
        if (threadIdx.x == 0) worst = 0;
        __syncthreads();
        for (uint32_t w = 0; w < SELECT_THREADS / 64; w++) vmax = wave_max[w] > vmax ? wave_max[w] : vmax;
        const uint32_t top = vmax ? 64u - (uint32_t)__builtin_clzll(vmax) : 1u, sh = top > 11u ? top - 11u : 0u;   // (vmax >> sh) < 2048
#pragma unroll
        for (uint32_t u = 0; u < SELECT_PER_THREAD; u++)
            if (threadIdx.x + u * SELECT_THREADS < count) rk[u] = atomicAdd(&hist[(uint32_t)(v[u] >> sh)], 1u);
        __syncthreads();
        {   // exclusive scan of the 2 048 counters: eight per thread, then the block's 256 partial sums
            constexpr uint32_t PER = SELECT_BUCKETS / SELECT_THREADS;
            uint32_t c[PER], sum = 0, big = 0;
#pragma unroll
            for (uint32_t x = 0; x < PER; x++) { c[x] = hist[threadIdx.x * PER + x]; sum += c[x]; big = c[x] > big ? c[x] : big; }
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += o; }
            if (lane == 63) wave_tot[wave] = incl;
            if (big > SELECT_BUCKET_MAX) atomicMax(&worst, big);
            __syncthreads();
            uint32_t base = incl - sum;
            for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
#pragma unroll
            for (uint32_t x = 0; x < PER; x++) { hist[threadIdx.x * PER + x] = base; base += c[x]; }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < SELECT_PER_THREAD; u++)
            if (threadIdx.x + u * SELECT_THREADS < count) buf[hist[(uint32_t)(v[u] >> sh)] + rk[u]] = v[u];
        __syncthreads();
        if (worst == 0) {
            // order inside the buckets: bucket b holds buf[hist[b] .. hist[b + 1]) (the last one up to count)
            for (uint32_t b = threadIdx.x; b < SELECT_BUCKETS; b += SELECT_THREADS) {
                const uint32_t lo = hist[b], hi = b + 1 < SELECT_BUCKETS ? hist[b + 1] : count;
                for (uint32_t i = lo + 1; i < hi; i++) {
                    const uint64_t x = buf[i];
                    uint32_t j = i;
                    while (j > lo && buf[j - 1] > x) { buf[j] = buf[j - 1]; j--; }
                    buf[j] = x;
                }
            }
            __syncthreads();
        } else {   // (block-uniform) not spread: the general network over the same LDS image
            for (uint32_t i = count + threadIdx.x; i < m; i += blockDim.x) buf[i] = ~0ull;
            bitonic_sort<true>(buf, m);
        }
    } else if (m <= lds_elems) {
        buf = lds;
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) buf[i] = (i < count) ? gbuf[i] : ~0ull;
        bitonic_sort<true>(buf, m);
    } else {
        buf = gbuf;  // cap is a power of two >= m on this path
        for (uint32_t i = count + threadIdx.x; i < m; i += blockDim.x) buf[i] = ~0ull;
        bitonic_sort<false>(buf, m);
    }

    // distinct rank of every element; first s distinct go to the output row
    const uint32_t per = (m + blockDim.x - 1) / blockDim.x;
    const uint32_t i0 = threadIdx.x * per;
    uint32_t local = 0;
    for (uint32_t i = i0; i < i0 + per && i < count; i++) local += (i == 0 || buf[i] != buf[i - 1]) ? 1u : 0u;
    // block exclusive scan of `local`
    uint32_t incl = local;
    __syncthreads();   // (wave_tot is reused)
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

// each thread writes 32 bases: two packed words and one validity word (positions past `length` are invalid)
__global__ __launch_bounds__(256) void synth_genomes_kernel(uint32_t *__restrict__ packed, uint32_t *__restrict__ valid,
                                                            const uint64_t *__restrict__ starts,
                                                            uint64_t length, uint32_t first,
                                                            uint32_t members, uint64_t seed,
                                                            uint32_t sub_thr) {
    const uint32_t g = blockIdx.y;  // local index; the series index is first + g
    const uint32_t species = (first + g) / members, member = (first + g) % members;
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 32;
    if (p0 >= length) return;
    const uint64_t ka = synth_key(seed, species, 0), km = synth_key(seed, species, member + 1);
    const uint64_t w = splitmix64(ka + (p0 >> 5));
    uint32_t out[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 32; j++) {
        uint64_t p = p0 + j;
        uint32_t base = (uint32_t)(w >> (2 * (p & 31))) & 3u;
        uint64_t u = splitmix64(km + p);
        if ((uint32_t)(u >> 32) < sub_thr) base = (base + 1 + (uint32_t)((u >> 8) % 3)) & 3u;
        out[j >> 4] |= base << (2 * (j & 15));
    }
    const uint64_t at = starts[g] + p0;   // a multiple of 32
    packed[at >> 4] = out[0];
    packed[(at >> 4) + 1] = out[1];
    const uint64_t left = length - p0;
    valid[at >> 5] = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
}

// ASCII stream bytes -> the resident form: `src` holds stream bytes [pos0, pos0 + n) of a genome whose first base sits at
// base offset `gbase` (a multiple of 64); pos0 is a multiple of 32.  Every byte other than A,C,G,T becomes an invalid
// position.  32 bases per thread.
__global__ __launch_bounds__(256) void pack_bases_kernel(const uint8_t *__restrict__ src, uint64_t n, uint64_t gbase_plus_pos0,
                                                         uint32_t *__restrict__ packed, uint32_t *__restrict__ valid) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; 32 * t < n; t += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t out[2] = {0, 0}, ok = 0;
        const uint64_t b0 = 32 * t;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const uint32_t c = b0 + j < n ? src[b0 + j] : 0u;
            const bool good = c == 'A' || c == 'C' || c == 'G' || c == 'T';
            out[j >> 4] |= (good ? (((c >> 1) ^ (c >> 2)) & 3u) : 0u) << (2 * (j & 15));
            ok |= (good ? 1u : 0u) << j;
        }
        const uint64_t at = gbase_plus_pos0 + b0;
        packed[at >> 4] = out[0];
        packed[(at >> 4) + 1] = out[1];
        valid[at >> 5] = ok;
    }
}

// the resident form -> stream bytes ('N' at every invalid position): ghip_genomes_to_host.  16 bases per thread.
__global__ __launch_bounds__(256) void unpack_bases_kernel(const uint32_t *__restrict__ packed, const uint32_t *__restrict__ valid,
                                                           uint64_t gbase, uint4 *__restrict__ out, uint64_t len) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // 16 bases per thread
    if (16 * t >= len) return;
    const uint64_t at = gbase + 16 * t;
    const uint32_t w = packed[at >> 4], ok = (valid[at >> 5] >> ((uint32_t)at & 16u)) & 0xffffu;
    uint32_t a[4] = {ascii4(w & 0xffu), ascii4((w >> 8) & 0xffu), ascii4((w >> 16) & 0xffu), ascii4(w >> 24)};
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        const uint32_t m4 = (ok >> (4 * qd)) & 0xfu;
        // spread the four validity bits to byte masks; invalid bytes become 'N'
        const uint32_t bm = ((m4 & 1u) ? 0xffu : 0u) | ((m4 & 2u) ? 0xff00u : 0u) | ((m4 & 4u) ? 0xff0000u : 0u) | ((m4 & 8u) ? 0xff000000u : 0u);
        a[qd] = (a[qd] & bm) | (0x4e4e4e4eu & ~bm);
    }
    out[t] = make_uint4(a[0], a[1], a[2], a[3]);
}

// Validity bitmaps of a GROUP of genomes that arrived as 2-bit codes + runs (the packed ingest ships many small files in
// one copy): gtab[2 m] = first bitmap word of member m relative to vw (the group's first bitmap word), gtab[2 m + 1] = its
// length in bases.  Bits [0, len) of every member are set, the rest of its words stay clear (the allocation is
// zero-filled), then the runs -- (start, length, member) triples, start relative to the member -- are cleared.
__global__ __launch_bounds__(256) void valid_fill_group_kernel(const uint32_t *__restrict__ gtab, uint32_t *__restrict__ vw) {
    const uint32_t off = gtab[2 * blockIdx.y], len = gtab[2 * blockIdx.y + 1];
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; 32ull * t < len; t += gridDim.x * blockDim.x) {
        const uint32_t left = len - 32 * t;
        vw[off + t] = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
    }
}
__global__ __launch_bounds__(64) void valid_clear_runs_group_kernel(const uint32_t *__restrict__ runs /* start, len, member */, const uint32_t *__restrict__ gtab,
                                                                    uint32_t *__restrict__ vw, int single) {
    const uint32_t start = runs[3 * blockIdx.x], n = runs[3 * blockIdx.x + 1], member = single ? 0u : runs[3 * blockIdx.x + 2];
    uint32_t *mine = vw + gtab[2 * member];
    // word-wise: a neighbouring run may share a word, hence atomicAnd
    const uint32_t w0 = start >> 5, w1 = (start + n - 1) >> 5;
    for (uint32_t w = w0 + threadIdx.x; w <= w1; w += 64) {
        const uint32_t lo = w == w0 ? (start & 31u) : 0u, hi = w == w1 ? ((start + n - 1) & 31u) : 31u;
        const uint32_t m = (hi == 31u ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
        atomicAnd(&mine[w], ~m);
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

void ghip_launch_synth(ghip_ctx *ctx, uint32_t *d_packed, uint32_t *d_valid, const uint64_t *d_starts, uint64_t length,
                       uint32_t first, uint32_t count, uint32_t members, uint64_t seed, uint32_t sub_thr) {
    uint64_t threads = (length + 31) / 32;
    dim3 grid((unsigned)((threads + 255) / 256), count);
    ghip_prof_begin(ctx, "synth_genomes");
    hipLaunchKernelGGL(synth_genomes_kernel, grid, dim3(256), 0, ctx->stream, d_packed, d_valid, d_starts, length, first, members, seed, sub_thr);
    ghip_prof_end(ctx);
}

// stream bytes [pos0, pos0 + n) of the genome at base offset gbase (device memory `d_src`) -> the resident form
void ghip_launch_pack_bases(hipStream_t stream, const uint8_t *d_src, uint64_t n, uint64_t gbase_plus_pos0, uint32_t *d_packed, uint32_t *d_valid) {
    if (n == 0) return;
    const uint64_t threads = (n + 31) / 32;
    hipLaunchKernelGGL(pack_bases_kernel, dim3((unsigned)std::min<uint64_t>((threads + 255) / 256, 1u << 20)), dim3(256), 0, stream, d_src, n,
                       gbase_plus_pos0, d_packed, d_valid);
}

// the resident form of bases [0, len) of the genome at base offset gbase -> stream bytes (d_out 16-byte aligned, room for
// len rounded up to 16)
void ghip_launch_unpack_bases(hipStream_t stream, const uint32_t *d_packed, const uint32_t *d_valid, uint64_t gbase, uint8_t *d_out, uint64_t len) {
    if (len == 0) return;
    const uint64_t groups = (len + 15) / 16;
    hipLaunchKernelGGL(unpack_bases_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, d_packed, d_valid, gbase,
                       reinterpret_cast<uint4 *>(d_out), len);
}

// The packed ingest: the 2-bit codes are copied straight to their place; this builds the validity bits of the group's
// members from the tables that travelled with them: gtab (2 u32 per member) and the runs (3 u32 each).
// Both kernels on `stream`, after the copy that brought the tables.  single: one member whose runs carry the byte value in
// their third field (the one-file parser's table), not a member index.
void ghip_launch_valid_from_runs(hipStream_t stream, const uint32_t *d_gtab, uint32_t n_members, uint64_t max_len, const uint32_t *d_runs,
                                 uint32_t n_runs, uint32_t *d_valid_group, bool single) {
    if (n_members == 0 || max_len == 0) return;
    const uint64_t words = (max_len + 31) / 32;
    hipLaunchKernelGGL(valid_fill_group_kernel, dim3((unsigned)std::min<uint64_t>((words + 255) / 256, 4096), n_members), dim3(256), 0, stream, d_gtab, d_valid_group);
    if (n_runs) hipLaunchKernelGGL(valid_clear_runs_group_kernel, dim3(n_runs), dim3(64), 0, stream, d_runs, d_gtab, d_valid_group, single ? 1 : 0);
}

// timing experiment only (scripts/sketch_variants.sh): unused dynamic LDS caps the workgroups per CU
static unsigned dbg_extra_lds() {
    static const unsigned v = [] { const char *e = getenv("GHIP_DBG_EXTRA_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
    return v;
}

void ghip_launch_sketch_kmers(ghip_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_valid, const uint64_t *d_starts,
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
                       d_packed, d_valid, d_starts, d_lens, d_slot_genome, d_slot_thr, d_slot_cand_start, d_slot_cand_cap, \
                       d_work + off, seed, d_cand, d_cand_count, so, luts)
            if (seeds && seed == 0) GHIP_LAUNCH21(true, true);
            else if (seeds) GHIP_LAUNCH21(true, false);
            else if (seed == 0) GHIP_LAUNCH21(false, true);
            else GHIP_LAUNCH21(false, false);
#undef GHIP_LAUNCH21
        } else {
            hipLaunchKernelGGL(sketch_kmers_kernel_rt, dim3(grid), dim3(GHIP_SKETCH_THREADS), 0, ctx->stream,
                               d_packed, d_valid, d_starts, d_lens, d_slot_genome, d_slot_thr, d_slot_cand_start,
                               d_slot_cand_cap, d_work + off, (int)k, seed, d_cand, d_cand_count);
        }
    }
    ghip_prof_end(ctx);
}

void ghip_launch_sketch_select(ghip_ctx *ctx, const uint32_t *d_slot_genome, size_t n_slots,
                               uint64_t *d_cand, const uint32_t *d_cand_count,
                               const uint64_t *d_slot_cand_start, const uint32_t *d_slot_cand_cap, uint32_t max_cap,
                               uint32_t s, uint64_t *d_hashes, uint32_t *d_lens, uint32_t *d_status) {
    if (n_slots == 0) return;
    // the LDS a block asks for follows the launch's longest candidate list (caps are powers of two): 32 KiB at the
    // default s = 1000, so that five blocks share a CU; lists beyond 8 192 entries are sorted in global memory
    uint32_t lds_elems = 2;
    while (lds_elems < max_cap && lds_elems < SELECT_LDS_ELEMS) lds_elems <<= 1;
    const size_t select_lds = lds_elems * sizeof(uint64_t) + SELECT_BUCKETS * sizeof(uint32_t);
    if (select_lds > 48 * 1024) ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(sketch_select_kernel), select_lds);
    ghip_prof_begin(ctx, "sketch_select");
    hipLaunchKernelGGL(sketch_select_kernel, dim3((unsigned)n_slots), dim3(SELECT_THREADS),
                       select_lds, ctx->stream, d_slot_genome, d_cand, d_cand_count,
                       d_slot_cand_start, d_slot_cand_cap, s, lds_elems, d_hashes, d_lens, d_status);
    ghip_prof_end(ctx);
}
