// ANI on candidate pairs, batched on gfx950: replaces the per-pair `skani dist` subprocess of
// SkaniClusterer::calculate_ani (reference src/skani.rs:708-788).  Build-defined estimator in
// skani's style (FracMinHash seeds k=15 c=125, 20 kb chunks, seed matches colinear within a chunk,
// containment^(1/k), aligned-fraction gate, two-decimal percent) -- skani parity is UNPINNED, see
// DESIGN.md "ANI" and oracle/galah_oracle_ani.c (the definition this file reproduces bit for bit).
// The device does integer work only; the host finishes pow/rounding.
//
//   ani_seeds : pass over the base stream; 2-bit codes of the k-mer and of its reverse complement (k <= 16, fit u32);
//               selected by a multiplicative mix of their strand-symmetric sum (seed_common.h); the seeds are appended
//               as (canonical code u32, loc u32 = chunk << 16 | strand << 15 | offset in chunk) and counted per chunk.
//   ani_bin   : per genome, counting sort of the seed list by the top 14 bits of a second
//               multiplicative hash of the code; writes the binned list and its bin offsets
//               (CSR).  Every genome uses the same bin function, so two genomes can be joined
//               bin by bin -- no hash table, no atomics on the device-wide memory.
//   ani_pairs : one workgroup per PAIR (both directions at once): a wave stages the seeds of 64
//               consecutive bins of both genomes in LDS and joins them; every anchor (equal codes)
//               gives an orientation and a diagonal, i.e. one of 16 bands; per chunk and band the
//               seeds with an anchor there are counted in LDS (u16 votes); M_c = the votes of the
//               bands that hold >= 3 seeds (colinear matches only), capped at T_c; a chunk is
//               aligned iff M_c*10000 >= 510*T_c; emits the (M_c, T_c) of the lower-median containment
//               over the aligned chunks of the shorter genome (of both at equal length; rank selection, exact),
//               their number, and the aligned bases per direction.
#include "ghip_internal.h"
#include "seed_common.h"

namespace {

using namespace ghip_seed;

// Standalone seeding pass (the fused form lives in sketch.hip: sketch_kmers<K, true>) over the resident bases (2-bit
// codes + validity bitmap, ghip_internal.h).
__global__ __launch_bounds__(GHIP_SKETCH_THREADS) void ani_seeds_kernel(
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ valid, const uint64_t *__restrict__ starts,
    const uint64_t *__restrict__ lens, const ghip_sketch_work *__restrict__ work, SeedOut so) {
    __shared__ SeedLds sl;
    const ghip_sketch_work wk = work[blockIdx.x];
    const uint32_t g = wk.slot;
    const uint64_t L = lens[g];
    const uint64_t blk0 = (uint64_t)wk.chunk * GHIP_SKETCH_CHUNK;
    const uint32_t toff = threadIdx.x * GHIP_SKETCH_POS_PER_THREAD;
    const uint64_t p0 = blk0 + toff;
    const uint32_t K = so.k, sthr = so.seed_thr[g];
    SeedBlock sb = seed_block_begin(sl, so, g, blk0);
    {
        // every base is fetched once: own 64 bases (one 16-byte vector, 8 bytes of validity) from memory, the K-1 <= 15
        // overlap bases from the next lane's registers (lane 63: from memory); lanes past the stream end hold invalid positions
        const bool live = p0 < L;
        const uint64_t base0 = starts[g] + p0;
        const uint32_t *pw = packed + (base0 >> 4), *vw = valid + (base0 >> 5);
        uint4 q = make_uint4(0, 0, 0, 0);
        uint2 vb = make_uint2(0, 0);
        if (live) { q = *reinterpret_cast<const uint4 *>(pw); vb = *reinterpret_cast<const uint2 *>(vw); }
        uint32_t o0 = __shfl_down(q.x, 1, 64), v2 = __shfl_down(vb.x, 1, 64);
        if ((threadIdx.x & 63u) == 63u) { o0 = live ? pw[4] : 0u; v2 = live ? vw[2] : 0u; }
        const uint32_t words[5] = {q.x, q.y, q.z, q.w, o0}, vbits[3] = {vb.x, vb.y, v2};
        const uint32_t mask = (K < 16) ? ((1u << (2 * K)) - 1) : ~0u;  // K <= 16: codes fit 32 bits
        const uint32_t top = 2 * (K - 1);
        uint32_t fwd = 0, rev = 0;
        uint32_t good = 0;
        const int NB = GHIP_SKETCH_POS_PER_THREAD + K - 1;  // K <= 16 -> at most 79 bases = 5 words
#pragma unroll
        for (int b = 0; b < 80; b++) {
            uint32_t code = (words[b >> 4] >> (2 * (b & 15))) & 3u;
            const bool ok = (vbits[b >> 5] >> (b & 31)) & 1u;
            fwd = ((fwd << 2) | code) & mask;
            rev = (rev >> 2) | ((3u - code) << top);
            good = ok ? good + 1 : 0;
            const bool pass = b < NB && b >= (int)K - 1 && good >= K && seed_selected(fwd, rev, so.mul, sthr);
            if (pass) seed_mark_at(sl, sb, fwd, (uint32_t)(b - ((int)K - 1)));
        }
    }
    seed_block_flush<false>(sl, so, sb, toff, packed + ((starts[g] + blk0) >> 4));
}


constexpr uint32_t BIN_COUNT = GHIP_ANI_BIN_COUNT, BIN_THREADS = 1024;
constexpr uint32_t SEG_BINS = BIN_COUNT / SEGMENTS;          // bins of one segment
constexpr uint32_t BIN_WINDOW = 8192;                        // seeds sorted inside LDS at a time (64 KiB)
constexpr uint32_t BIN_PER_THREAD = BIN_WINDOW / BIN_THREADS;
constexpr size_t BIN_LDS = SEG_BINS * sizeof(uint32_t) + (size_t)BIN_WINDOW * 8;

// Counting sort of ONE SEGMENT of one genome's seed list (seed_common.h: the seeding pass files every seed under the
// top bits of its bin) by the remaining bin bits, in LDS; the order inside a bin is arbitrary.  A segment of a genome up
// to ~8 Mb at c = 125 fits the window: it is read once (the seeds wait in registers between the histogram and the
// placement) and written once, contiguously.  Longer segments take the windowed form (destination slots to a scratch
// array, every window re-reads them).  The segments' outputs are concatenated: bin_start[g][b] is relative to the
// genome's first seed slot, so shards can be concatenated.
__global__ __launch_bounds__(BIN_THREADS) void ani_bin_kernel(
    const uint32_t *__restrict__ in_code, const uint32_t *__restrict__ in_loc,
    uint32_t *__restrict__ out_code, uint32_t *__restrict__ out_loc,
    const uint64_t *__restrict__ seed_start, const uint32_t *__restrict__ seg_count,
    uint32_t *__restrict__ bin_start, uint32_t *__restrict__ pos_tmp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem_raw);               // SEG_BINS counters, then cursors
    uint32_t *w_code = hist + SEG_BINS, *w_loc = w_code + BIN_WINDOW;      // the output window
    __shared__ uint32_t wave_tot[BIN_THREADS / 64];
    const uint32_t g = blockIdx.x / SEGMENTS, seg = blockIdx.x % SEGMENTS;
    const uint64_t s0 = seed_start[g];
    const uint32_t segcap = (uint32_t)(seed_start[g + 1] - s0) / SEGMENTS;
    uint32_t out_base = 0, total = 0, n = 0;
    for (uint32_t s = 0; s < SEGMENTS; s++) {
        const uint32_t c = min(seg_count[(uint64_t)g * SEGMENTS + s], segcap);
        if (s < seg) out_base += c;
        if (s == seg) n = c;
        total += c;
    }
    const uint64_t in0 = s0 + (uint64_t)seg * segcap, out0 = s0 + out_base;
    uint32_t *bstart = bin_start + (uint64_t)g * (BIN_COUNT + 1) + seg * SEG_BINS;
    const bool one_window = n <= BIN_WINDOW;
    for (uint32_t i = threadIdx.x; i < SEG_BINS; i += BIN_THREADS) hist[i] = 0;
    __syncthreads();
    uint32_t code_r[BIN_PER_THREAD], loc_r[BIN_PER_THREAD];
    if (one_window) {
#pragma unroll
        for (uint32_t k = 0; k < BIN_PER_THREAD; k++) {
            const uint32_t i = threadIdx.x + k * BIN_THREADS;
            code_r[k] = 0; loc_r[k] = 0;
            if (i < n) { code_r[k] = in_code[in0 + i]; loc_r[k] = in_loc[in0 + i]; atomicAdd(&hist[code_bin(code_r[k]) & (SEG_BINS - 1)], 1u); }
        }
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += BIN_THREADS) atomicAdd(&hist[code_bin(in_code[in0 + i]) & (SEG_BINS - 1)], 1u);
    }
    __syncthreads();
    // exclusive scan of hist: consecutive bins per thread
    constexpr uint32_t PER = SEG_BINS / BIN_THREADS;
    uint32_t local[PER], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) { local[j] = hist[threadIdx.x * PER + j]; sum += local[j]; }
    uint32_t incl = sum;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
    uint32_t run = base + incl - sum;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        hist[threadIdx.x * PER + j] = run;
        bstart[threadIdx.x * PER + j] = out_base + run;
        run += local[j];
    }
    if (seg == SEGMENTS - 1 && threadIdx.x == BIN_THREADS - 1) bstart[SEG_BINS] = total;  // bin_start[g][BIN_COUNT] == n
    __syncthreads();
    if (one_window) {
#pragma unroll
        for (uint32_t k = 0; k < BIN_PER_THREAD; k++) {
            const uint32_t i = threadIdx.x + k * BIN_THREADS;
            if (i < n) {
                const uint32_t p = atomicAdd(&hist[code_bin(code_r[k]) & (SEG_BINS - 1)], 1u);
                w_code[p] = code_r[k]; w_loc[p] = loc_r[k];
            }
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += BIN_THREADS) { out_code[out0 + k] = w_code[k]; out_loc[out0 + k] = w_loc[k]; }
        return;
    }
    // destination slot of every seed (one LDS atomic each), kept in a scratch array (re-read once per window)
    for (uint32_t i = threadIdx.x; i < n; i += BIN_THREADS) pos_tmp[in0 + i] = atomicAdd(&hist[code_bin(in_code[in0 + i]) & (SEG_BINS - 1)], 1u);
    __syncthreads();
    // Scatter through the LDS window and write each window out contiguously: scattering 4-byte
    // stores straight to HBM costs a whole sector per seed (PMC: 3.3 GB written for 0.27 GB).
    for (uint32_t w0 = 0; w0 < n; w0 += BIN_WINDOW) {
        for (uint32_t i = threadIdx.x; i < n; i += BIN_THREADS) {
            const uint32_t p = pos_tmp[in0 + i] - w0;
            if (p < BIN_WINDOW) { w_code[p] = in_code[in0 + i]; w_loc[p] = in_loc[in0 + i]; }
        }
        __syncthreads();
        const uint32_t m = min(BIN_WINDOW, n - w0);
        for (uint32_t k = threadIdx.x; k < m; k += BIN_THREADS) {
            out_code[out0 + w0 + k] = w_code[k];
            out_loc[out0 + w0 + k] = w_loc[k];
        }
        __syncthreads();
    }
}

#ifdef GHIP_DBG_ANI_PHASES   // timing experiment only: cycles per phase of wave 0 of pair 0
__device__ unsigned long long g_ani_phase[16];
#define PH(i) do { if (pair == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); g_ani_phase[i] += t_ - ph_t; ph_t = t_; } } while (0)
#else
#define PH(i) do {} while (0)
#endif
constexpr uint32_t ANI_PAIR_WAVES_WIDE = 8, ANI_PAIR_WAVES_TALL = 16;   // waves per pair: 8 for long pair lists, 16 for short ones
constexpr uint32_t ANI_STAGE = 192;  // seeds of one round's consecutive bins staged per wave and genome (rounds are sized for ~150)
constexpr uint32_t ANI_BAND_SHIFT = 12, ANI_MIN_COLINEAR = 3;   // 4 kb diagonal bands; >= 3 seeds of a chunk must agree (oracle: GO_ANI_*)
constexpr uint32_t ANI_VOTE_WORDS = 9;                          // per chunk: 16 bands x u16, then the repeat-seed counter

// location word of a seed: chunk << 16 | strand << 15 | offset in chunk
__device__ __forceinline__ uint32_t loc_pos(uint32_t loc, uint32_t chunk) { return (loc >> 16) * chunk + (loc & 0x7fffu); }

// band of the anchor (q seed, r seed): orientation and diagonal (oracle/galah_oracle_ani.c)
__device__ __forceinline__ uint32_t anchor_band(uint32_t qloc, uint32_t qpos, uint32_t rloc, uint32_t chunk) {
    const uint32_t o = ((qloc ^ rloc) >> 15) & 1u, rpos = loc_pos(rloc, chunk);
    // same orientation: bands of the odd number 2 (rpos - qpos) + 1 centred on 0, so that swapping q and r mirrors them
    return o ? (((rpos + qpos) >> ANI_BAND_SHIFT) & 7u) | 8u
             : ((2u * (rpos - qpos) + 1u + (1u << ANI_BAND_SHIFT)) >> (ANI_BAND_SHIFT + 1)) & 7u;
}

// The vote of one seed of chunk `c` whose anchors fall into the bands of `mask` (!= 0): its band's counter if they all
// agree, the chunk's repeat counter otherwise.  u16 counters, two per LDS word (a counter never exceeds the seeds of
// a chunk, <= 32768, so halves cannot carry into each other).
__device__ __forceinline__ void cast_vote(uint32_t *votes, uint32_t c, uint32_t mask) {
    const uint32_t band = (uint32_t)__builtin_ctz(mask);
    const bool single = (mask & (mask - 1)) == 0;
    atomicAdd(&votes[c * ANI_VOTE_WORDS + (single ? (band >> 1) : 8u)], single ? 1u << (16u * (band & 1u)) : 1u);
}

// The vote of an r seed is settled as its anchors arrive, by the lanes that find them: `old` = the seed's band mask before
// this anchor (the value the atomicOr returned, which orders the anchors of a seed).  First anchor: a vote for its band.
// An anchor in a DIFFERENT band while the mask held exactly one: that vote moves to the chunk's repeat counter.  Anything
// later changes nothing.  (u16 halves of a word: -1 on a half is +0xffff0000 / +0xffffffff on the word, which leaves the
// other half alone once the matching +1 has landed -- and word arithmetic mod 2^32 commutes, so the order the two
// atomics of different lanes land in does not matter.)  Same final counters as one cast_vote per seed on its full mask.
__device__ __forceinline__ void settle_r_vote(uint32_t *votes, uint32_t c, uint32_t old, uint32_t band) {
    if (old == 0) {
        atomicAdd(&votes[c * ANI_VOTE_WORDS + (band >> 1)], 1u << (16u * (band & 1u)));
    } else if ((old & (old - 1)) == 0 && old != (1u << band)) {
        const uint32_t b1 = (uint32_t)__builtin_ctz(old);
        atomicAdd(&votes[c * ANI_VOTE_WORDS + (b1 >> 1)], 0u - (1u << (16u * (b1 & 1u))));
        atomicAdd(&votes[c * ANI_VOTE_WORDS + 8u], 1u);
    }
}

// Appends (M_c << 32 | T_c) of every aligned chunk (M_c*10000 >= 510*T_c) to `list` and returns this
// thread's share of the aligned bases.  M_c = votes of the bands holding >= ANI_MIN_COLINEAR seeds, plus the repeat seeds
// of a chunk that has such a band, capped at T_c.
__device__ __forceinline__ unsigned long long collect_aligned(const uint32_t *votes, const uint32_t *tc, uint32_t nch, uint64_t L,
                                                              uint32_t chunk, bool listed, unsigned long long *list, uint32_t *n_al) {
    unsigned long long bases = 0;
    for (uint32_t c = threadIdx.x; c < nch; c += blockDim.x) {
        const unsigned long long t = tc[c];
        unsigned long long m = 0;
#pragma unroll
        for (uint32_t w = 0; w < 8; w++) {
            const uint32_t v = votes[c * ANI_VOTE_WORDS + w], lo = v & 0xffffu, hi = v >> 16;
            m += (lo >= ANI_MIN_COLINEAR ? lo : 0u) + (hi >= ANI_MIN_COLINEAR ? hi : 0u);
        }
        if (m) m += votes[c * ANI_VOTE_WORDS + 8];
        if (m > t) m = t;
        if (t >= 1 && m * 10000ull >= 510ull * t) {
            if (listed) list[atomicAdd(n_al, 1u)] = (m << 32) | t;
            const uint64_t lo = (uint64_t)c * chunk;
            uint64_t hi = lo + chunk;
            if (hi > L) hi = L;
            bases += hi - lo;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bases += __shfl_xor(bases, off, 64);
    return bases;
}

// The 64 bins a wave joins in one round are consecutive, so their seeds form ONE contiguous run
// per genome: the wave copies both runs into its private LDS stage with coalesced loads and every
// lane then joins its own seeds out of LDS.  (Walking the bins straight from global memory chained
// ~25 dependent L2 round trips per lane and bin.)  A round whose run exceeds the stage -- a
// pathologically repetitive or a very large genome -- falls back to the global-memory walk for that round only.
// WAVES wavefronts share a pair, each a slice of the bins.  Long pair lists run 8 (three workgroups per CU: throughput); a
// list the chip holds at once anyway runs 16 -- half the rounds per wave, i.e. about half a pair's latency, which is all
// a short launch costs (the lazy clusterer's rounds, a single calculate_ani).
// GENERAL = false: both genomes at the same seed density and their per-chunk state in LDS (every pair of genome-sized
// inputs).  GENERAL = true: the per-chunk state lives in a global scratch region of the pair (genomes of any length: a
// 100 Mb eukaryotic bin has 5 000 chunks) and the pair may be MIXED -- its genomes seeded at different densities
// (ghip_ani_density): it is evaluated at the sparser one, the denser genome's seeds are filtered by their selection hash
// and its per-chunk totals recounted.  `sel` (nullable) = the indices of the pairs this launch handles.
template <uint32_t ANI_PAIR_WAVES, bool GENERAL>
__global__ __launch_bounds__(ANI_PAIR_WAVES * 64) __attribute__((amdgpu_waves_per_eu(6))) void ani_pairs_kernel(
    const uint32_t *__restrict__ pairs, uint32_t n_pairs, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ seed_code,
    const uint32_t *__restrict__ seed_loc, const uint64_t *__restrict__ seed_start,
    const uint32_t *__restrict__ bin_start, const uint32_t *__restrict__ chunk_total,
    const uint64_t *__restrict__ chunk_start, const uint64_t *__restrict__ glen, const uint32_t *__restrict__ seed_thr,
    uint32_t seed_k, uint32_t chunk, uint32_t ro_cap, uint32_t *__restrict__ scratch, const uint64_t *__restrict__ scratch_off,
    uint64_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ unsigned long long red[6];
    __shared__ uint32_t n_al;
    __shared__ uint32_t st_code[ANI_PAIR_WAVES][2][ANI_STAGE + 8];   // + 8: the branch-free compares read up to 8 entries past a bin
    __shared__ uint32_t st_loc[ANI_PAIR_WAVES][2][ANI_STAGE];
    __shared__ uint32_t st_mask[ANI_PAIR_WAVES][ANI_STAGE / 2];  // band mask (16 bits) of every staged r seed, two per word

    // physical block b runs on XCD b % 8: give each XCD runs of 32 consecutive pairs (one precluster's
    // genomes are then re-read from that XCD's L2)
    const uint32_t xcd = blockIdx.x & 7u, slot_in_xcd = blockIdx.x >> 3;
    const uint32_t slot = ((slot_in_xcd >> 5) * 8u + xcd) * 32u + (slot_in_xcd & 31u);
    if (slot >= n_pairs) return;
    const uint32_t pair = sel ? sel[slot] : slot;
#ifdef GHIP_DBG_ANI_PHASES
    unsigned long long ph_t = __builtin_readcyclecounter();
#endif
    const uint32_t q = pairs[2 * pair], r = pairs[2 * pair + 1];
    const uint32_t nchq = (uint32_t)(chunk_start[q + 1] - chunk_start[q]);
    const uint32_t nchr = (uint32_t)(chunk_start[r + 1] - chunk_start[r]);
    // per-chunk state: votes of q, votes of r, [GENERAL: recounted totals of q, of r], the aligned-chunk list
    uint32_t *vq;
    unsigned long long *list;
    uint16_t *ro_base;
    if constexpr (GENERAL) {
        vq = scratch + scratch_off[slot];   // zeroed by the launcher
        ro_base = reinterpret_cast<uint16_t *>(smem_raw);
    } else {
        vq = reinterpret_cast<uint32_t *>(smem_raw);
    }
    uint32_t *vr = vq + (size_t)nchq * ANI_VOTE_WORDS;
    uint32_t *tq_re = vr + (size_t)nchr * ANI_VOTE_WORDS, *tr_re = tq_re + nchq;   // GENERAL only
    if constexpr (GENERAL) {
        list = reinterpret_cast<unsigned long long *>(scratch + ((scratch_off[slot] + (size_t)(nchq + nchr) * (ANI_VOTE_WORDS + 1) + 1) & ~(uint64_t)1));
    } else {
        list = reinterpret_cast<unsigned long long *>(smem_raw + (((size_t)(nchq + nchr) * ANI_VOTE_WORDS * 4 + 7) & ~(size_t)7));
        ro_base = reinterpret_cast<uint16_t *>(list + (nchq + nchr));
        for (uint32_t i = threadIdx.x; i < (nchq + nchr) * ANI_VOTE_WORDS; i += blockDim.x) vq[i] = 0;
    }
    if (threadIdx.x < 6) red[threadIdx.x] = threadIdx.x == 4 ? ~0ull : 0ull;   // red[4]: the minimum the rank selection's atomicMin starts from
    if (threadIdx.x == 0) n_al = 0;
    __syncthreads();

    // the pair's density is the sparser of its genomes': a genome seeded denser than that is filtered by the selection
    // hash of its codes (recomputed from the canonical code: fwd + rev is symmetric) and its totals are recounted
    const uint32_t thr_q = seed_thr[q], thr_r = seed_thr[r], thr_pair = min(thr_q, thr_r), smul = seed_mul(seed_k);
    const bool filt_q = GENERAL && thr_q != thr_pair, filt_r = GENERAL && thr_r != thr_pair;
    auto at_pair_density = [&](uint32_t code) { return (0u - 2u - code - revcomp_code(code, seed_k)) * smul < thr_pair; };
    const uint32_t *qb = bin_start + (uint64_t)q * (BIN_COUNT + 1), *rb = bin_start + (uint64_t)r * (BIN_COUNT + 1);
    const uint32_t *qc = seed_code + seed_start[q], *rc = seed_code + seed_start[r];
    const uint32_t *ql = seed_loc + seed_start[q], *rl = seed_loc + seed_start[r];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if constexpr (GENERAL) {
        if (filt_q) for (uint32_t i = threadIdx.x; i < qb[BIN_COUNT]; i += blockDim.x) if (at_pair_density(qc[i])) atomicAdd(&tq_re[ql[i] >> 16], 1u);
        if (filt_r) for (uint32_t i = threadIdx.x; i < rb[BIN_COUNT]; i += blockDim.x) if (at_pair_density(rc[i])) atomicAdd(&tr_re[rl[i] >> 16], 1u);
    }
    uint32_t *sq = st_code[wave][0], *sr = st_code[wave][1];
    uint32_t *lq = st_loc[wave][0], *lr = st_loc[wave][1];
    uint32_t *rm = st_mask[wave];
    // stage offsets of the round's r bins: dynamic LDS (behind the aligned-chunk list), ro_cap entries per wave
    uint16_t *ro = ro_base + (size_t)wave * ro_cap;
    for (uint32_t k = lane; k < ANI_STAGE / 2; k += 64) rm[k] = 0;
    // 2^32 - 1 is never a canonical code (its reverse complement, 0, is smaller): entries of the r stage that no round of
    // this workgroup has written yet can never compare equal to a seed
    for (uint32_t k = lane; k < ANI_STAGE + 8; k += 64) sr[k] = 0xffffffffu;
    // Bins joined per wave and round, sized so that a run holds ~150 seeds of the larger genome (the stage holds 192;
    // a longer run -- a pathologically repetitive stretch -- takes the global-memory walk for that round only): 64 for
    // genomes up to ~5 Mb at c = 125, 32 / 16 / 8 above, 256 / 1024 for small genomes and contigs, which would otherwise
    // pay the fixed cost of a round (bounds, barriers, three short loops) 32 times per wave for a handful of seeds.
    uint32_t rbins = 64;
    {
        const uint32_t tmax = max(qb[BIN_COUNT], rb[BIN_COUNT]);  // total seeds of the larger genome
        if (tmax > 41000) rbins = tmax > 1312000 ? 1u : tmax > 656000 ? 2u : tmax > 328000 ? 4u : tmax > 164000 ? 8u : (tmax > 82000 ? 16u : 32u);
        else if (ro_cap > 1024) rbins = tmax <= 3000 ? 1024u : (tmax <= 12000 ? 256u : 64u);
    }
    // Software pipeline over the rounds of a wave: the BOUNDS of round t + 2 and the SEEDS of round t + 1 (three per lane and
    // array, held in registers) are fetched while round t is joined out of LDS -- a round's ~8 dependent global-memory
    // round trips (bounds, then the runs) overlap the previous round's join instead of preceding its own.
    struct Bounds { uint32_t rs, q_lo, q_hi, r_hi; };
    constexpr uint32_t STAGE_REGS = ANI_STAGE / 64;
    const uint32_t stride = ANI_PAIR_WAVES * rbins;
    auto load_bounds = [&](uint32_t b) {
        Bounds o{0, 0, 0, 0};
        if (b < BIN_COUNT) { o.rs = rb[min(b + lane, BIN_COUNT)]; o.q_lo = qb[b]; o.q_hi = qb[b + rbins]; o.r_hi = rb[b + rbins]; }
        return o;
    };
    // a round is joined out of the stage when both runs are non-empty and fit it (else: skipped, or the global-memory walk)
    auto is_staged = [&](const Bounds &o) {
        const uint32_t r_lo = __shfl(o.rs, 0, 64);
        return o.q_lo != o.q_hi && r_lo != o.r_hi && o.q_hi - o.q_lo <= ANI_STAGE && o.r_hi - r_lo <= ANI_STAGE;
    };
    uint32_t pc1[STAGE_REGS], ph1[STAGE_REGS], pc2[STAGE_REGS], ph2[STAGE_REGS];
    auto fetch_runs = [&](const Bounds &o) {
        const uint32_t r_lo = __shfl(o.rs, 0, 64), nqs = o.q_hi - o.q_lo, nrs = o.r_hi - r_lo;
#pragma unroll
        for (uint32_t u = 0; u < STAGE_REGS; u++) {
            const uint32_t k = lane + 64 * u;
            pc1[u] = 0; ph1[u] = 0; pc2[u] = 0; ph2[u] = 0;
            if (k < nqs) { pc1[u] = qc[o.q_lo + k]; ph1[u] = ql[o.q_lo + k]; }
            if (k < nrs) { pc2[u] = rc[r_lo + k]; ph2[u] = rl[r_lo + k]; }
        }
    };
    Bounds cur = load_bounds(wave * rbins), nxt = load_bounds(wave * rbins + stride);
    if (is_staged(cur)) fetch_runs(cur);
    PH(0);   // prologue
    for (uint32_t b0 = wave * rbins; b0 < BIN_COUNT; b0 += stride) {
        const Bounds c = cur;
        const uint32_t rs = c.rs, q_lo = c.q_lo, q_hi = c.q_hi, r_hi = c.r_hi;
        const uint32_t r_lo = __shfl(rs, 0, 64);
        const bool staged = is_staged(c);
        cur = nxt;
        nxt = load_bounds(b0 + 2 * stride);
        PH(1);   // bounds
        if (staged) {
            {
                const uint32_t nqs = q_hi - q_lo, nrs = r_hi - r_lo;
#pragma unroll
                for (uint32_t u = 0; u < STAGE_REGS; u++) {
                    const uint32_t k = lane + 64 * u;
                    uint32_t c2 = pc2[u];
                    if constexpr (GENERAL) { if (filt_r && k < nrs && !at_pair_density(c2)) c2 = 0xffffffffu; }   // never a code: matches nothing
                    if (k < nqs) { sq[k] = pc1[u]; lq[k] = ph1[u]; }
                    if (k < nrs) { sr[k] = c2; lr[k] = ph2[u]; }
                }
            }
        }
        if (is_staged(cur)) fetch_runs(cur);   // the next round's runs: in flight during this round's join
        if (q_lo == q_hi || r_lo == r_hi) continue;  // wave-uniform: one genome has no seed in these bins
        if (staged) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            PH(2);   // staging
            // Flattened join: q seeds are dealt to the lanes one each (not one BIN per lane, whose cost is the
            // largest bin of the wave squared); a seed's bin, hence its short r range, follows from its code.
            // Every anchor sets its band in the q seed's mask (a register) and in the r seed's mask (LDS); the q seed votes
            // once its bin is compared, the r seed's vote is settled by the lanes that find its anchors (settle_r_vote).
            const uint32_t nq = q_hi - q_lo, nr = r_hi - r_lo;
            if (lane < rbins) ro[lane] = (uint16_t)(rs - r_lo);   // r range of bin b0 + x = [ro[x], ro[x + 1])
            for (uint32_t u = 64 + lane; u < rbins; u += 64) ro[u] = (uint16_t)(rb[b0 + u] - r_lo);
            if (lane == 63) ro[rbins] = (uint16_t)nr;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            PH(3);   // offsets
            for (uint32_t i = lane; i < nq; i += 64) {
                const uint32_t c = sq[i];
                if constexpr (GENERAL) { if (filt_q && !at_pair_density(c)) continue; }
                const uint32_t x = code_bin(c) - b0;
                const uint32_t j0 = ro[x], jz = ro[x + 1];
                // the first 8 seeds of the bin branch-free (a bin holds ~2.4 seeds, more than 8 with P ~ 5e-4): an
                // exec-masked loop over the bin costs ~6 scalar instructions per iteration and the kernel is as much
                // scalar- as vector-issue bound
                // no bound check: an entry past the bin is a seed of ANOTHER bin (this round's next bin, or a stale one of
                // an earlier round, or the never-a-code fill) -- a code determines its bin, so it cannot equal c
                uint32_t mm = 0;
#ifdef GHIP_DBG_ANI_NOCOMPARE   // timing experiment only: wrong results
                mm = (j0 ^ c) == 0x12345u ? 1u : 0u;
#else
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) mm |= (uint32_t)(sr[j0 + u] == c) << u;
#endif
                bool any = mm != 0;
                for (uint32_t j = j0 + 8; j < jz; j++) any |= (sr[j] == c);
                if (any) {   // ~half of the seeds of related genomes
                    const uint32_t qloc = lq[i], qpos = loc_pos(qloc, chunk);
                    uint32_t qmask = 0;
                    while (mm) {
                        const uint32_t j = j0 + (uint32_t)__builtin_ctz(mm);
                        mm &= mm - 1;
                        const uint32_t rloc = lr[j], band = anchor_band(qloc, qpos, rloc, chunk), sh = 16u * (j & 1u);
                        qmask |= 1u << band;
                        settle_r_vote(vr, rloc >> 16, (atomicOr(&rm[j >> 1], (1u << band) << sh) >> sh) & 0xffffu, band);
                    }
                    for (uint32_t j = j0 + 8; j < jz; j++)
                        if (sr[j] == c) {
                            const uint32_t rloc = lr[j], band = anchor_band(qloc, qpos, rloc, chunk), sh = 16u * (j & 1u);
                            qmask |= 1u << band;
                            settle_r_vote(vr, rloc >> 16, (atomicOr(&rm[j >> 1], (1u << band) << sh) >> sh) & 0xffffu, band);
                        }
                    cast_vote(vq, qloc >> 16, qmask);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            PH(4);   // join
            for (uint32_t w = lane; w < (nr + 1) / 2; w += 64) rm[w] = 0;   // the r votes were settled as the anchors arrived
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the stage is rewritten in the next round
            PH(5);   // r votes
        } else {  // a run longer than the stage: lane-per-bin walk in global memory
            for (uint32_t u = lane; u < rbins; u += 64) {
                const uint32_t qs = qb[b0 + u], qe = qb[b0 + u + 1], rs_ = rb[b0 + u], re = rb[b0 + u + 1];
                for (uint32_t i = qs; i < qe; i++) {
                    const uint32_t c = qc[i], qloc = ql[i], qpos = loc_pos(qloc, chunk);
                    uint32_t qmask = 0;
                    if constexpr (GENERAL) { if ((filt_q || filt_r) && !at_pair_density(c)) continue; }
                    for (uint32_t j = rs_; j < re; j++)
                        if (rc[j] == c) qmask |= 1u << anchor_band(qloc, qpos, rl[j], chunk);
                    if (qmask) cast_vote(vq, qloc >> 16, qmask);
                }
                for (uint32_t j = rs_; j < re; j++) {
                    const uint32_t c = rc[j], rloc = rl[j];
                    uint32_t rmask = 0;
                    if constexpr (GENERAL) { if ((filt_q || filt_r) && !at_pair_density(c)) continue; }
                    for (uint32_t i = qs; i < qe; i++)
                        if (qc[i] == c) { const uint32_t qloc = ql[i]; rmask |= 1u << anchor_band(qloc, loc_pos(qloc, chunk), rloc, chunk); }
                    if (rmask) cast_vote(vr, rloc >> 16, rmask);
                }
            }
        }
    }
    __syncthreads();
    PH(6);   // waiting for the other waves
    // (M_c, T_c) of the aligned chunks, then the LOWER MEDIAN containment M_c/T_c
    // by rank selection (exact: fractions compared by cross-multiplication)
    // the median is taken over the chunks of the SHORTER genome (both at equal length): a chunk of the longer one is diluted
    // wherever the shorter one ends inside it (oracle/galah_oracle_ani.c)
    const uint64_t Lq = glen[q], Lr = glen[r];
    const unsigned long long bq = collect_aligned(vq, filt_q ? tq_re : chunk_total + chunk_start[q], nchq, Lq, chunk, Lq <= Lr, list, &n_al);
    const unsigned long long br = collect_aligned(vr, filt_r ? tr_re : chunk_total + chunk_start[r], nchr, Lr, chunk, Lr <= Lq, list, &n_al);
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&red[2], bq); atomicAdd(&red[5], br); }
    __syncthreads();
    const uint32_t n = n_al;
    PH(7);   // collect
    // (red[4] = ~0 was written HERE by thread 0 until round 6, with no barrier between it and the other waves' atomicMin in the
    // cross-multiplication branch below -- a wave that got there first had its minimum overwritten; found by tests/emu/wavesan.cpp)
    if (threadIdx.x == 0) red[3] = n;
    const uint32_t target = n ? (n - 1) / 2 : 0;
    // Exact order of the fractions M_c/T_c through f64 keys: the quotient is correctly rounded, hence monotone, and
    // two different fractions with T < 2^26 differ by more than an ulp, so key order == cross-multiplied order and
    // equal keys <=> equal fractions.  The keys live in the (now idle) seed stage; longer lists compare by
    // cross-multiplication.  Every member of the median's tie group holds the same fraction: the smallest
    // (m, t) of the group is reported.
    constexpr uint32_t KEY_CAP = sizeof(st_code) / sizeof(double);
    if (n < GHIP_ANI_POOL_BELOW) {
        // too few chunks for a median (of two it is the minimum): the pooled count over them (oracle: GO_ANI_POOL_BELOW)
        if (threadIdx.x == 0 && n) {
            unsigned long long pm = 0, pt = 0;
            for (uint32_t e = 0; e < n; e++) { pm += list[e] >> 32; pt += list[e] & 0xffffffffull; }
            red[4] = (pm << 32) | pt;   // (at most eight chunks of at most 32 768 seeds each: the sums stay in their halves)
        }
    } else if (n <= KEY_CAP) {
        // (key, entry) pairs sorted in LDS by an ascending-only bitonic network (positions >= n act as +infinity without
        // being stored: pairs_join.hip has the same network) -- ~45 block-wide exchange steps for the ~500 aligned chunks
        // of two 5 Mb genomes, where counting every entry's rank against every other took 500 steps per thread and a
        // quarter of the kernel's instructions.  Ties: equal keys <=> equal fractions; the order inside a tie group is
        // by entry, so the group's first element is its smallest (m, t).
        double *keys = reinterpret_cast<double *>(&st_code[0][0][0]);
        for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
            const unsigned long long me = list[e];
            keys[e] = (double)(uint32_t)(me >> 32) / (double)(uint32_t)me;
        }
        uint32_t P = 2;
        while (P < n) P <<= 1;
        auto exchange = [&](uint32_t i, uint32_t x) {
            if (x >= n) return;
            const double ka = keys[i], kb = keys[x];
            if (ka < kb) return;
            const unsigned long long ma = list[i], mb = list[x];
            if (ka > kb || ma > mb) { keys[i] = kb; keys[x] = ka; list[i] = mb; list[x] = ma; }
        };
        // 64 consecutive threads of a step whose span (kk, or 2 j) is at most 128 stay inside their wavefront's own 128
        // entries: such a step only needs the wave's own writes (a wavefront fence) unless the step before it reached
        // further -- 5 block barriers instead of 45 for 512 entries
        uint32_t prev_span = ~0u;   // the keys were written block-wide
        auto step_sync = [&](uint32_t span) {
            if (span > 128 || prev_span > 128) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
            prev_span = span;
        };
        for (uint32_t kk = 2; kk <= P && n > 1; kk <<= 1) {
            step_sync(kk);
            const uint32_t half = kk >> 1;
            for (uint32_t t = threadIdx.x; t < P / 2; t += blockDim.x) {
                const uint32_t i = 2 * t - (t & (half - 1));   // = kk * (t / half) + t % half, half a power of two (no division)
                exchange(i, i ^ (kk - 1));
            }
            for (uint32_t j = half >> 1; j > 0; j >>= 1) {
                step_sync(2 * j);
                for (uint32_t t = threadIdx.x; t < P / 2; t += blockDim.x) {
                    const uint32_t i = 2 * t - (t & (j - 1));   // = 2 j * (t / j) + t % j, j a power of two
                    exchange(i, i + j);
                }
            }
        }
        __syncthreads();
        if (n) {
            const double km = keys[target];
            for (uint32_t e = threadIdx.x; e < n; e += blockDim.x)
                if (keys[e] == km && (e == 0 || keys[e - 1] != km)) red[4] = list[e];   // exactly one thread
        }
    } else {
        for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
            const unsigned long long me = list[e];
            const unsigned long long mm = me >> 32, mt = me & 0xffffffffull;
            uint32_t lt = 0, le = 0;
            for (uint32_t j = 0; j < n; j++) {
                const unsigned long long o = list[j];
                const unsigned long long l = (o >> 32) * mt, rr = mm * (o & 0xffffffffull);  // o < me  <=>  om*mt < mm*ot
                lt += l < rr ? 1u : 0u;
                le += l <= rr ? 1u : 0u;
            }
            if (lt <= target && target < le) atomicMin(&red[4], me);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && n) { red[0] = red[4] >> 32; red[1] = red[4] & 0xffffffffull; }
    if (threadIdx.x == 0) red[4] = 0;
    __syncthreads();
    if (threadIdx.x < 6) out[(uint64_t)pair * 6 + threadIdx.x] = red[threadIdx.x];
    PH(8);   // median + output
}

}  // namespace

void ghip_launch_ani_seeds(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, const uint32_t *d_seed_thr, uint32_t chunk,
                           uint32_t *d_seed_code, uint32_t *d_seed_loc, const uint64_t *d_seed_start,
                           uint32_t *d_seg_count, uint32_t *d_chunk_total, const uint64_t *d_chunk_start,
                           const ghip_sketch_work *d_work, size_t n_work) {
    if (n_work == 0) return;
    ghip_seed::SeedOut so{k, ghip_seed::seed_mul(k), chunk, d_seed_thr, ghip_seed::seed_chunk_magic(chunk), d_seed_code, d_seed_loc, d_seed_start, d_seg_count, d_chunk_total, d_chunk_start};
    ghip_prof_begin(ctx, "ani_seeds");
    for (size_t off = 0; off < n_work; off += GHIP_MAX_GRID)  // one AQL dispatch holds < 2^32 work-items
        hipLaunchKernelGGL(ani_seeds_kernel, dim3((unsigned)std::min<size_t>(n_work - off, GHIP_MAX_GRID)),
                           dim3(GHIP_SKETCH_THREADS), 0, ctx->stream, g->d_packed, g->d_valid, g->d_starts, g->d_lens, d_work + off, so);
    ghip_prof_end(ctx);
}

void ghip_launch_ani_bin(ghip_ctx *ctx, size_t n, const uint32_t *in_code, const uint32_t *in_loc, uint32_t *out_code,
                         uint32_t *out_loc, const uint64_t *d_seed_start, const uint32_t *d_seg_count,
                         uint32_t *d_bin_start, uint32_t *d_pos_tmp) {
    if (n == 0) return;
    ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(ani_bin_kernel), 96 * 1024);
    ghip_prof_begin(ctx, "ani_bin");
    for (size_t off = 0; off < n; off += GHIP_MAX_GRID / ghip_seed::SEGMENTS) {   // one block per genome and segment
        const size_t m = std::min<size_t>(n - off, GHIP_MAX_GRID / ghip_seed::SEGMENTS);
        hipLaunchKernelGGL(ani_bin_kernel, dim3((unsigned)(m * ghip_seed::SEGMENTS)), dim3(BIN_THREADS), BIN_LDS, ctx->stream,
                           in_code, in_loc, out_code, out_loc, d_seed_start + off, d_seg_count + off * ghip_seed::SEGMENTS,
                           d_bin_start + off * (size_t)(BIN_COUNT + 1), d_pos_tmp);
    }
    ghip_prof_end(ctx);
}

namespace {
// Index slices on the move (comm.cpp, ghip_exchange_ani_index): run x copies runs[3x + 2] 32-bit words from
// src + runs[3x] to dst + runs[3x + 1].  One launch packs everything a rank sends out of one array of the index.
__global__ void __launch_bounds__(256) copy_runs_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, const uint64_t *__restrict__ runs) {
    const uint64_t *r = runs + 3 * (size_t)blockIdx.x;
    const uint32_t *s = src + r[0];
    uint32_t *d = dst + r[1];
    const uint64_t n = r[2];
    for (uint64_t i = (uint64_t)blockIdx.y * 256 + threadIdx.x; i < n; i += 256ull * gridDim.y) d[i] = s[i];
}
}  // namespace

void ghip_launch_copy_runs(hipStream_t stream, const uint32_t *d_src, uint32_t *d_dst, const uint64_t *d_runs, size_t n_runs) {
    for (size_t off = 0; off < n_runs; off += 1u << 20) {
        const size_t m = std::min<size_t>(n_runs - off, 1u << 20);
        hipLaunchKernelGGL(copy_runs_kernel, dim3((unsigned)m, 8), dim3(256), 0, stream, d_src, d_dst, d_runs + 3 * off);
    }
}

namespace {
struct DeviceFree_ {   // pool blocks handed back at scope exit (the caller synchronises the stream before they are reused)
    ghip_ctx *ctx;
    std::vector<void *> ptrs;
    ~DeviceFree_() { for (void *p : ptrs) ghip_pool_free(ctx, p); }
};
constexpr size_t ANI_PER_CHUNK_LDS = ANI_VOTE_WORDS * sizeof(uint32_t) + sizeof(uint64_t);   // votes + aligned-chunk list entry
template <uint32_t WAVES>
constexpr size_t ani_static_lds() { return (size_t)WAVES * (2 * (ANI_STAGE + 8) + 2 * ANI_STAGE + ANI_STAGE / 2) * 4 + 256; }   // seed stage + band masks

// dynamic LDS of an LDS-form launch with WAVES waves per pair whose largest pair holds pair_chunks chunks (both genomes), 0
// if it does not fit next to the kernel's static arrays; ro_cap falls back to the narrow rounds when the wide rounds'
// offset table has no room
template <uint32_t WAVES>
size_t ani_pairs_lds(uint32_t pair_chunks, uint32_t &ro_cap) {
    const size_t fixed = (size_t)pair_chunks * ANI_PER_CHUNK_LDS + 24, room = 160 * 1024 - ani_static_lds<WAVES>();
    if (fixed + (size_t)WAVES * ro_cap * sizeof(uint16_t) > room) ro_cap = 66;
    const size_t lds = fixed + (size_t)WAVES * ro_cap * sizeof(uint16_t);   // + r-bin offsets of a round
    return lds > room ? 0 : lds;
}

template <uint32_t WAVES, bool GENERAL>
void ani_pairs_launch(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *d_pairs, const uint32_t *d_sel, size_t n_launch, size_t lds,
                      uint32_t ro_cap, uint32_t *d_scratch, const uint64_t *d_scratch_off, uint64_t *d_out) {
    if (lds > 48 * 1024) ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(ani_pairs_kernel<WAVES, GENERAL>), lds);   // (48 KiB is the default allowance)
    const uint32_t grid = ((uint32_t)n_launch + 255u) / 256u * 256u;  // whole runs of 32 pairs on each of 8 XCDs
    hipLaunchKernelGGL((ani_pairs_kernel<WAVES, GENERAL>), dim3(grid), dim3(WAVES * 64), lds, ctx->stream, d_pairs, (uint32_t)n_launch, d_sel,
                       idx->d_seed_code, idx->d_seed_loc, idx->d_seed_start, idx->d_bin_start, idx->d_chunk_total,
                       idx->d_chunk_start, idx->d_glen, idx->d_seed_thr, idx->k, idx->chunk, ro_cap, d_scratch, d_scratch_off, d_out);
}
}  // namespace

// Two kernel forms (see ani_pairs_kernel): a pair whose genomes share a seed density and whose per-chunk state fits the
// LDS -- every pair of genome-sized inputs -- takes the LDS form; a MIXED pair (densities differ: one genome short
// enough for a denser tier, ghip_ani_density) or a pair of very long genomes (above GHIP_ANI_LDS_PAIR_CHUNKS chunks together:
// 58 Mb at the default chunk) takes the general form, its per-chunk state in a zeroed global scratch region.
int ghip_launch_ani_pairs(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *pairs, const uint32_t *d_pairs, size_t n_pairs,
                          uint64_t *d_out) {
    if (n_pairs == 0) return GHIP_OK;
    auto nch_of = [&](uint32_t g) { return (uint32_t)(idx->chunk_start[g + 1] - idx->chunk_start[g]); };
    std::vector<uint32_t> sel_fast, sel_gen;
    uint32_t fast_chunks = 1;
    bool small_fast = false, small_gen = false;   // some genome small enough for the wide rounds (their offset table costs LDS)
    size_t n_gen = 0;
    const bool force_general = ctx->opt.ani_force_general != 0;   // measurement aid: every pair through the general form
    for (size_t p = 0; p < n_pairs; p++) {
        const uint32_t q = pairs[2 * p], r = pairs[2 * p + 1], nch = nch_of(q) + nch_of(r);
        n_gen += (force_general || idx->seed_thr[q] != idx->seed_thr[r] || nch > GHIP_ANI_LDS_PAIR_CHUNKS) ? 1 : 0;
    }
    for (size_t p = 0; p < n_pairs; p++) {
        const uint32_t q = pairs[2 * p], r = pairs[2 * p + 1], nch = nch_of(q) + nch_of(r);
        const bool small = idx->seed_count[q] <= 12000 || idx->seed_count[r] <= 12000;
        if (force_general || idx->seed_thr[q] != idx->seed_thr[r] || nch > GHIP_ANI_LDS_PAIR_CHUNKS) { sel_gen.push_back((uint32_t)p); small_gen |= small; }
        else { if (n_gen) sel_fast.push_back((uint32_t)p); fast_chunks = std::max(fast_chunks, nch); small_fast |= small; }
    }
    const size_t n_fast = n_pairs - n_gen;
    // 16 waves per pair while there are fewer pairs than CUs (GHIP_ANI_TALL_BELOW; 0 = never).  Measured, 8 -> 16 waves:
    // 8 pairs 0.133 -> 0.095 ms, 96 pairs 0.168 -> 0.122, but 384 pairs 0.192 -> 0.235 and 1 536 pairs 0.44 -> 0.64.
    const size_t tall_below = ctx->opt.ani_tall_below;   // (ghip_options: the tests switch it)
    DeviceFree_ tmp{ctx, {}};
    uint32_t *d_sel = nullptr;
    if (n_gen) {   // the two selections, one upload: [fast..., general...]
        std::vector<uint32_t> both(sel_fast);
        both.insert(both.end(), sel_gen.begin(), sel_gen.end());
        d_sel = (uint32_t *)ghip_pool_alloc(ctx, both.size() * sizeof(uint32_t));
        if (!d_sel) return GHIP_EHIP;
        tmp.ptrs.push_back(d_sel);
        if (hipMemcpyAsync(d_sel, both.data(), both.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return ghip_set_error(ctx, GHIP_EHIP, "ani_pairs: selection upload failed");
    }
    ghip_prof_begin(ctx, "ani_pairs");
    if (n_fast) {
        // wide rounds (256 / 1024 bins) only exist when some genome is small enough to use them: their offset table
        // costs 16 KiB of LDS per workgroup, which large-genome runs keep for a third resident workgroup per CU
        uint32_t ro_cap = small_fast ? 1026 : 66, ro_tall = ro_cap;
        const size_t lds_tall = n_fast < tall_below ? ani_pairs_lds<ANI_PAIR_WAVES_TALL>(fast_chunks, ro_tall) : 0;
        if (lds_tall) ani_pairs_launch<ANI_PAIR_WAVES_TALL, false>(ctx, idx, d_pairs, n_gen ? d_sel : nullptr, n_fast, lds_tall, ro_tall, nullptr, nullptr, d_out);
        else {
            const size_t lds = ani_pairs_lds<ANI_PAIR_WAVES_WIDE>(fast_chunks, ro_cap);   // (always fits: GHIP_ANI_LDS_PAIR_CHUNKS)
            ani_pairs_launch<ANI_PAIR_WAVES_WIDE, false>(ctx, idx, d_pairs, n_gen ? d_sel : nullptr, n_fast, lds, ro_cap, nullptr, nullptr, d_out);
        }
    }
    int rc = GHIP_OK;
    if (n_gen) {
        // scratch region of a pair, in 32-bit words: votes (9 per chunk) + recounted totals (1 per chunk), padded to an even
        // count, then the aligned-chunk list (one u64 per chunk); launches are cut so that a launch's scratch stays below 1 GiB
        const uint32_t ro_cap = small_gen ? 1026 : 66;
        const size_t lds = (size_t)ANI_PAIR_WAVES_WIDE * ro_cap * sizeof(uint16_t);
        constexpr uint64_t SCRATCH_WORDS_MAX = (1ull << 30) / 4;
        size_t at = 0;
        while (at < n_gen && rc == GHIP_OK) {
            std::vector<uint64_t> off;
            uint64_t words = 0;
            size_t m = 0;
            while (at + m < n_gen) {
                const uint32_t p = sel_gen[at + m], nch = nch_of(pairs[2 * p]) + nch_of(pairs[2 * p + 1]);
                const uint64_t w = (((uint64_t)nch * (ANI_VOTE_WORDS + 1) + 1) & ~1ull) + 2ull * nch;
                if (m && words + w > SCRATCH_WORDS_MAX) break;
                off.push_back(words);
                words += w;
                m++;
            }
            uint32_t *d_scratch = (uint32_t *)ghip_pool_alloc(ctx, std::max<uint64_t>(words, 2) * sizeof(uint32_t));
            uint64_t *d_off = (uint64_t *)ghip_pool_alloc(ctx, m * sizeof(uint64_t));
            if (!d_scratch || !d_off) { ghip_pool_free(ctx, d_scratch); ghip_pool_free(ctx, d_off); rc = GHIP_EHIP; break; }
            if (hipMemsetAsync(d_scratch, 0, words * sizeof(uint32_t), ctx->stream) != hipSuccess ||
                hipMemcpyAsync(d_off, off.data(), m * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = ghip_set_error(ctx, GHIP_EHIP, "ani_pairs: scratch set-up failed");
            else ani_pairs_launch<ANI_PAIR_WAVES_WIDE, true>(ctx, idx, d_pairs, d_sel + sel_fast.size() + at, m, lds, ro_cap, d_scratch, d_off, d_out);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "ani_pairs (general form) failed");   // `off` and the scratch are reused
            ghip_pool_free(ctx, d_scratch); ghip_pool_free(ctx, d_off);
            at += m;
        }
    }
    ghip_prof_end(ctx);
#ifdef GHIP_DBG_ANI_PHASES
    {
        unsigned long long h[16];
        hipStreamSynchronize(ctx->stream);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ani_phase), sizeof h);
        static const char *nm[9] = {"prologue", "bounds", "staging", "offsets", "join", "r-votes", "wait-waves", "collect", "median+out"};
        fprintf(stderr, "[ani_pairs phases, wave 0 of pair 0, cumulative cycles]");
        for (int i = 0; i < 9; i++) fprintf(stderr, " %s %llu", nm[i], h[i]);
        fprintf(stderr, "\n");
    }
#endif
    return rc;
}
