// Device code shared by the standalone seeding pass (ani.hip: ani_seeds) and the fused
// MinHash + seeding pass (sketch.hip: sketch_kmers<K, true>).
#pragma once
#include "ghip_internal.h"

namespace ghip_seed {

// Seed selection (oracle/galah_oracle_ani.c): kept iff ((0 - 2 - fwd - rev) * mul) mod 2^32 < 2^32 / c with
// mul = 0x85EBCA6B << (32 - 2k).  The key fwd + rev mod 4^k is strand-symmetric like the canonical code but needs no
// masked min: the fused pass holds the COMPLEMENTS of both codes in the low 2k bits of two registers (with junk above),
// their plain sum is 2(4^k - 1) - fwd - rev = -2 - fwd - rev mod 4^k, and the shifted multiplier drops the junk --
// add, multiply, compare per position (MurmurHash3's fmix32 of the canonical code, round 1: 12 instructions; a
// multiplicative mix of it, before: 6).  The multiplier differs from the bin hash's (ani.hip code_bin).
__host__ __device__ __forceinline__ uint32_t seed_mul(uint32_t k) { return 0x85EBCA6Bu << (32u - 2u * k); }
__device__ __forceinline__ bool seed_selected(uint32_t fwd, uint32_t rev, uint32_t mul, uint32_t thr) { return (0u - 2u - fwd - rev) * mul < thr; }

// Bin of a code in the per-genome join index (ani_bin / ani_pairs): the top bits of a second multiplicative hash.  The
// seeding pass already files every seed under the top SEG_BITS of its bin -- the genome's unordered list is SEGMENTS
// lists, each of capacity (genome capacity / SEGMENTS) -- so that ani_bin sorts a whole segment inside LDS in one
// read and one write (a genome's 40 000 seeds x 8 B do not fit; the one-list form re-read them once per output window).
__device__ __forceinline__ uint32_t code_bin(uint32_t code) { return (code * 0x9E3779B1u) >> (32 - GHIP_ANI_BIN_BITS); }
constexpr uint32_t SEG_BITS = 3, SEGMENTS = 1u << SEG_BITS;
static_assert(SEGMENTS == GHIP_ANI_SEGMENTS, "host and device agree on the segment count");
__device__ __forceinline__ uint32_t code_segment(uint32_t code) { return (code * 0x9E3779B1u) >> (32 - SEG_BITS); }

// A block-relative position `at` (< 2^16: at most one chunk length plus one block) divided by the chunk length, exactly,
// by one v_mul_hi: M = ceil(2^32 / chunk) overestimates 2^32 / chunk by e / chunk with e < chunk <= 2^15, so at * M / 2^32
// exceeds at / chunk by less than at * e / (chunk * 2^32) < 1 / chunk -- never enough to reach the next integer.  (The
// compiler's division by a run-time value is ~15 instructions, once per seed and pass of the flush.)
__host__ __device__ __forceinline__ uint32_t seed_chunk_magic(uint32_t chunk) { return chunk <= 1 ? 0u : (uint32_t)(((1ull << 32) + chunk - 1) / chunk); }
__device__ __forceinline__ uint32_t seed_chunk_of(uint32_t at, uint32_t magic) { return magic ? __umulhi(at, magic) : at; }

// packed location of a seed: chunk << 16 | strand << 15 | offset within the chunk (chunk length <= 32768)
__device__ __forceinline__ uint32_t seed_loc(uint32_t chunk_id, uint32_t strand, uint32_t off) { return (chunk_id << 16) | (strand << 15) | off; }

// A..T -> 0..3, anything else -> 4
__device__ __forceinline__ uint32_t base_code(uint32_t c) {
    uint32_t d = c - 0x41u;
    bool ok = d < 20u && ((0x80045u >> d) & 1u);
    return ok ? (((c >> 1) ^ (c >> 2)) & 3u) : 4u;
}

constexpr uint32_t SEED_LANE_CAP = 4;      // seeds buffered per lane (expected 64/c ~ 0.5 at c = 125)
constexpr uint32_t SEED_LDS_CHUNKS = 64;   // per-block chunk counters
constexpr uint32_t SEED_WAVES = GHIP_SKETCH_THREADS / 64;

struct SeedOut {  // where a block's seeds go (kernel argument, by value)
    uint32_t k, mul, chunk;        // mul = seed_mul(k)
    const uint32_t *seed_thr;      // [n] selection threshold of every genome, (2^32 - 1) / c_g: the density is per genome
                                   // (ghip_ani_density: short genomes are seeded 4x, 16x ... denser)
    uint32_t chunk_magic;          // seed_chunk_magic(chunk): positions -> chunks by one multiply
    uint32_t *seed_code;
    uint32_t *seed_loc;   // chunk << 16 | strand << 15 | offset in chunk
    const uint64_t *seed_start;   // [n + 1], capacity layout; a genome's capacity is a multiple of SEGMENTS
    uint32_t *seg_count;          // [n][SEGMENTS] seeds filed under each segment (may exceed the capacity: overflow)
    uint32_t *chunk_total;
    const uint64_t *chunk_start;
};

struct SeedLds {  // per-block LDS state; slot-major so that the 64 lanes of a wave write consecutive words
    uint32_t raw[SEED_LANE_CAP][GHIP_SKETCH_THREADS];   // forward code as the pass holds it (see seed_canon)
    uint32_t list[SEED_WAVES][SEED_LANE_CAP * 64];      // the flush's wave-compacted list of the rowed seeds (seed_block_flush)
    uint32_t ctot[SEED_LDS_CHUNKS];
    uint32_t seg_n[SEGMENTS], seg_base[SEGMENTS];
    uint32_t seg_over[SEGMENTS], seg_cur[SEGMENTS];   // seeds beyond a lane's rows (dense genomes): their number, their cursor
};

// Per-thread view of one block's seeding state.  A lane owns 64 consecutive positions; WHICH of them hold a seed is a
// 64-bit mask in registers (one v_or with a constant per append), so an append stores the code only -- one LDS write.
// The store address is clamped to the last row: a lane with more than SEED_LANE_CAP seeds (P ~ 1e-4 at c = 125) loses
// nothing, its mask still names every position, and the flush recomputes from the bases the codes that found no row.
struct SeedBlock {
    uint32_t g, ch_first, rem_first, scap;
    uint64_t sstart;
    uint32_t *ctot;
    uint32_t slot, slot_last;   // BYTE offsets into SeedLds::raw: the lane's next free row / its last row
    uint32_t m16;               // positions of the current 16-position group
    uint64_t mask;              // positions of the groups already closed
};

// reverse complement of a k-base big-endian 2-bit code (k <= 16)
__device__ __forceinline__ uint32_t revcomp_code(uint32_t f, uint32_t k) {
    uint32_t x = __brev(f);                                           // bases reversed, bits inside each base swapped
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    return ~x >> (32 - 2 * k);
}

// The position loop stores a seed's forward code only -- RAW: COMPLEMENTED ? its complement with junk above bit 2k (the
// fused pass reads it straight out of its reverse-complement window) : the code itself; the canonical code and the strand
// are worked out here, once per stored seed, not once per wave position.
template <bool COMPLEMENTED>
__device__ __forceinline__ void seed_canon(uint32_t raw, uint32_t k, uint32_t &canon, uint32_t &strand) {
    const uint32_t mask = (k < 16) ? ((1u << (2 * k)) - 1) : ~0u;
    const uint32_t f = (COMPLEMENTED ? ~raw : raw) & mask, r = revcomp_code(f, k);
    canon = min(f, r);
    strand = r < f ? 1u : 0u;
}

// call before the position loop (contains a __syncthreads)
__device__ __forceinline__ SeedBlock seed_block_begin(SeedLds &sl, const SeedOut &so, uint32_t g, uint64_t blk0) {
    SeedBlock sb;
    sb.g = g;
    sb.ch_first = (uint32_t)(blk0 / so.chunk);
    sb.rem_first = (uint32_t)(blk0 - (uint64_t)sb.ch_first * so.chunk);
    sb.sstart = so.seed_start[g];
    sb.scap = (uint32_t)(so.seed_start[g + 1] - sb.sstart) / SEGMENTS;   // capacity of ONE segment
    sb.ctot = so.chunk_total + so.chunk_start[g];
    sb.slot = 4 * threadIdx.x;
    sb.slot_last = 4 * ((SEED_LANE_CAP - 1) * GHIP_SKETCH_THREADS + threadIdx.x);
    sb.m16 = 0;
    sb.mask = 0;
    if (threadIdx.x < SEED_LDS_CHUNKS) sl.ctot[threadIdx.x] = 0;
    if (threadIdx.x < SEGMENTS) { sl.seg_n[threadIdx.x] = 0; sl.seg_over[threadIdx.x] = 0; sl.seg_cur[threadIdx.x] = 0; }
    __syncthreads();
    return sb;
}

// Lane-private append, inside the divergent `if (selected)`: min, store, add, or.  bit16 = the position's index in
// its group of 16 (a constant in the unrolled fused loop).
__device__ __forceinline__ void seed_mark(SeedLds &sl, SeedBlock &sb, uint32_t raw, uint32_t bit16) {
    // an (empty) inline-asm statement makes the compiler keep the s_cbranch_execz around this block: without it the
    // five instructions are if-converted and issue, EXEC = 0, on the 60 % of wave positions where no lane selected a seed
#ifndef GHIP_DBG_IFCVT   // timing experiment only
    asm volatile("");
#endif
    *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(&sl.raw[0][0]) + min(sb.slot, sb.slot_last)) = raw;
    sb.slot += 4 * GHIP_SKETCH_THREADS;
    sb.m16 |= 1u << bit16;
}
// the same with a run-time position p in [0, 64) (standalone pass)
__device__ __forceinline__ void seed_mark_at(SeedLds &sl, SeedBlock &sb, uint32_t raw, uint32_t p) {
    *reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(&sl.raw[0][0]) + min(sb.slot, sb.slot_last)) = raw;
    sb.slot += 4 * GHIP_SKETCH_THREADS;
    sb.mask |= 1ull << p;
}
// after every 16 positions: group v of the lane's four
__device__ __forceinline__ void seed_group_end(SeedBlock &sb, uint32_t v) {
    sb.mask |= (uint64_t)sb.m16 << (16u * v);
    sb.m16 = 0;
}

// A seed outside the lane's 64 positions (the fused pass: the few seeds of a genome that end before byte 20) goes
// straight to the global list.
__device__ __forceinline__ void seed_emit_global(const SeedOut &so, SeedBlock &sb, uint32_t canon, uint32_t strand, uint32_t rel) {
    const uint32_t at = sb.rem_first + rel, ch = seed_chunk_of(at, so.chunk_magic), seg = code_segment(canon);
    const uint32_t idx = atomicAdd(&so.seg_count[(uint64_t)sb.g * SEGMENTS + seg], 1u);
    if (idx < sb.scap) {
        const uint64_t o = sb.sstart + (uint64_t)seg * sb.scap + idx;
        so.seed_code[o] = canon;
        so.seed_loc[o] = seed_loc(sb.ch_first + ch, strand, at - ch * so.chunk);
    }
    atomicAdd(&sb.ctot[sb.ch_first + ch], 1u);
}

// call after the position loop by every thread of the block (contains __syncthreads).  rel0 = block-relative position
// of the lane's first seed position (bit 0 of its mask); block_packed = the packed word holding the block's first base
// (a block starts at a multiple of 16 bases).
template <bool COMPLEMENTED>
__device__ __forceinline__ void seed_block_flush(SeedLds &sl, const SeedOut &so, SeedBlock &sb, uint32_t rel0, const uint32_t *block_packed) {
    const uint32_t mine = (uint32_t)__popcll(sb.mask);
    // seeds with a row of their own in SeedLds::raw (the clamped stores of a lane with more than SEED_LANE_CAP seeds
    // overwrote its last row): they are filed through the block's segment runs; the others, rare, one by one
    const uint32_t rowed = mine <= SEED_LANE_CAP ? mine : SEED_LANE_CAP - 1;
    // A lane holds 64 / c ~ 0.5 seeds: filing them lane by lane ran the ~55 instructions per seed below three times per
    // wave (some lane nearly always holds three) for the ~32 seeds of the wave.  The wave first COMPACTS its rowed seeds
    // into a list -- entry = byte offset of the code in SeedLds::raw << 6 | position bit in the lane's mask -- and then
    // files one seed per lane: one round for up to 64 seeds.
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *list = sl.list[threadIdx.x >> 6];
    uint64_t over = sb.mask;   // ends as the positions beyond the rowed ones
    uint32_t n_wave = 0;       // wave-uniform: seeds in the list
#pragma unroll
    for (uint32_t i = 0; i < SEED_LANE_CAP; i++) {
        const bool has = i < rowed;
        const uint64_t b = __builtin_amdgcn_ballot_w64(has);
        if (has) {
            const uint32_t at = n_wave + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            list[at] = ((i * GHIP_SKETCH_THREADS + threadIdx.x) << 8) | (uint32_t)__builtin_ctzll(over);
            over &= over - 1;
        }
        n_wave += (uint32_t)__popcll(b);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pass 1: canonical code, strand, segment, and the rank in the block's run of that segment (one LDS counter per segment)
    const uint32_t rel_base = rel0 - threadIdx.x * GHIP_SKETCH_POS_PER_THREAD;   // uniform: a lane's positions start at 64 * thread + this
    uint32_t canon_r[SEED_LANE_CAP], place_r[SEED_LANE_CAP], rel_r[SEED_LANE_CAP];   // place = segment << 28 | strand << 27 | rank
#pragma unroll
    for (uint32_t r = 0; r < SEED_LANE_CAP; r++) {
        canon_r[r] = 0; place_r[r] = 0; rel_r[r] = 0;
        if (64u * r + lane < n_wave) {
            const uint32_t entry = list[64u * r + lane], off = entry >> 6;   // off = 4 * (row * threads + thread)
            uint32_t canon, strand;
            seed_canon<COMPLEMENTED>(*reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(&sl.raw[0][0]) + off), so.k, canon, strand);
            const uint32_t seg = code_segment(canon);
            canon_r[r] = canon;
            place_r[r] = (seg << 28) | (strand << 27) | atomicAdd(&sl.seg_n[seg], 1u);
            rel_r[r] = rel_base + ((off & (4u * GHIP_SKETCH_THREADS - 1u)) << 4) + (entry & 63u);   // 64 * thread + bit
        }
    }
    // The seeds that found no row (a lane with more than SEED_LANE_CAP of them: rare at c = 125, the RULE for genomes seeded
    // densely -- short contigs take every 15-mer, 64 per lane) are filed through the same segment runs, behind the rowed
    // ones: counted here, written below with a cursor per segment, their code read back from the packed bases both times
    // (two words and a funnel shift; the forward big-endian code is the little-endian field with its bases reversed, the
    // reverse complement's is simply its complement).  Emitting them one by one with two global atomics each, as this
    // did, made the 100 000-contig workload's pass 85x slower per base than a genome's.
    const uint32_t kmask = so.k < 16 ? (1u << (2 * so.k)) - 1u : ~0u;
    auto code_at = [&](uint32_t rel, uint32_t &canon, uint32_t &strand) {
        const uint32_t w0 = block_packed[rel >> 4], w1 = block_packed[(rel >> 4) + 1];
        const uint32_t le = __builtin_amdgcn_alignbit(w1, w0, 2u * (rel & 15u)) & kmask;
        uint32_t x = __brev(le);
        x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
        const uint32_t f = x >> (32 - 2 * so.k), r = ~le & kmask;
        canon = min(f, r);
        strand = r < f ? 1u : 0u;
    };
    for (uint64_t m = over; m; m &= m - 1) {
        uint32_t canon, strand;
        code_at(rel0 + (uint32_t)__builtin_ctzll(m), canon, strand);
        atomicAdd(&sl.seg_over[code_segment(canon)], 1u);
    }
    __syncthreads();
    // one global atomic per segment and block reserves the block's run in the genome's segment
    if (threadIdx.x < SEGMENTS) {
        const uint32_t c = sl.seg_n[threadIdx.x] + sl.seg_over[threadIdx.x];
        sl.seg_base[threadIdx.x] = c ? atomicAdd(&so.seg_count[(uint64_t)sb.g * SEGMENTS + threadIdx.x], c) : 0u;
    }
    __syncthreads();
    // pass 2: the lane's list entries again, from registers
#pragma unroll
    for (uint32_t r = 0; r < SEED_LANE_CAP; r++) {
        if (64u * r + lane < n_wave) {
            const uint32_t at = sb.rem_first + rel_r[r], chrel = seed_chunk_of(at, so.chunk_magic);
            const uint32_t seg = place_r[r] >> 28, idx = sl.seg_base[seg] + (place_r[r] & 0x07ffffffu);
            if (idx < sb.scap) {
                const uint64_t o = sb.sstart + (uint64_t)seg * sb.scap + idx;
                so.seed_code[o] = canon_r[r];
                so.seed_loc[o] = seed_loc(sb.ch_first + chrel, (place_r[r] >> 27) & 1u, at - chrel * so.chunk);
            }
            if (chrel < SEED_LDS_CHUNKS) atomicAdd(&sl.ctot[chrel], 1u);
            else atomicAdd(&sb.ctot[sb.ch_first + chrel], 1u);
        }
    }
    for (uint64_t m = over; m; m &= m - 1) {   // the seeds without a row: behind the rowed ones of their segment
        const uint32_t rel = rel0 + (uint32_t)__builtin_ctzll(m);
        uint32_t canon, strand;
        code_at(rel, canon, strand);
        const uint32_t at = sb.rem_first + rel, chrel = seed_chunk_of(at, so.chunk_magic), seg = code_segment(canon);
        const uint32_t idx = sl.seg_base[seg] + sl.seg_n[seg] + atomicAdd(&sl.seg_cur[seg], 1u);
        if (idx < sb.scap) {
            const uint64_t o = sb.sstart + (uint64_t)seg * sb.scap + idx;
            so.seed_code[o] = canon;
            so.seed_loc[o] = seed_loc(sb.ch_first + chrel, strand, at - chrel * so.chunk);
        }
        if (chrel < SEED_LDS_CHUNKS) atomicAdd(&sl.ctot[chrel], 1u);
        else atomicAdd(&sb.ctot[sb.ch_first + chrel], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SEED_LDS_CHUNKS && sl.ctot[threadIdx.x]) atomicAdd(&sb.ctot[sb.ch_first + threadIdx.x], sl.ctot[threadIdx.x]);
}

}  // namespace ghip_seed
