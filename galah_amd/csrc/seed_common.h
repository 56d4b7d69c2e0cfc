// Device code shared by the standalone seeding pass (ani.hip: ani_seeds) and the fused
// MinHash + seeding pass (sketch.hip: sketch_kmers<K, true>).
#pragma once
#include "ghip_internal.h"

namespace ghip_seed {

// Seed selection: a bijective multiplicative mix of the 32-bit canonical code, kept iff mix(code) < 2^32 / c (see
// oracle/galah_oracle_ani.c).  ONE multiply per position -- every position of every genome pays it (MurmurHash3's
// fmix32, used before, cost 6 more instructions per base: 0.6 ms per 5 Gbase).  The multiplier differs from the bin
// hash's (ani.hip code_bin), or all selected seeds would share their top bits and land in the first bins.
__device__ __forceinline__ uint32_t seed_mix(uint32_t code) { return code * 0x85EBCA6Bu; }

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
    uint32_t k, thr, chunk;
    uint32_t *seed_code;
    uint32_t *seed_loc;   // chunk << 16 | strand << 15 | offset in chunk
    const uint64_t *seed_start;
    uint32_t *seed_count;
    uint32_t *chunk_total;
    const uint64_t *chunk_start;
};

struct SeedLds {  // per-block LDS state; slot-major so that the 64 lanes of a wave write consecutive words
    uint32_t code[SEED_LANE_CAP][GHIP_SKETCH_THREADS];
    uint16_t pos[SEED_LANE_CAP][GHIP_SKETCH_THREADS];   // position relative to the block start (< 2^15) | strand << 15
    uint32_t ctot[SEED_LDS_CHUNKS];
    uint32_t wave_n[SEED_WAVES], wave_base[SEED_WAVES];
};

struct SeedBlock {  // per-thread view of one block's seeding state
    uint32_t g, ch_first, rem_first, scap, n_lane;
    uint64_t sstart;
    uint32_t *ctot;
};

// call before the position loop (contains a __syncthreads)
__device__ __forceinline__ SeedBlock seed_block_begin(SeedLds &sl, const SeedOut &so, uint32_t g, uint64_t blk0) {
    SeedBlock sb;
    sb.g = g;
    sb.ch_first = (uint32_t)(blk0 / so.chunk);
    sb.rem_first = (uint32_t)(blk0 - (uint64_t)sb.ch_first * so.chunk);
    sb.sstart = so.seed_start[g];
    sb.scap = (uint32_t)(so.seed_start[g + 1] - sb.sstart);
    sb.ctot = so.chunk_total + so.chunk_start[g];
    sb.n_lane = 0;
    if (threadIdx.x < SEED_LDS_CHUNKS) sl.ctot[threadIdx.x] = 0;
    __syncthreads();
    return sb;
}

// Lane-private append: no cross-lane ranking in the position loop (a ballot/mbcnt rank cost ~10 instructions
// on the 40 % of positions where some lane of the wave holds a seed).  rel = position - block start.
__device__ __forceinline__ void seed_append(SeedLds &sl, const SeedOut &so, SeedBlock &sb, bool pass, uint32_t canon, uint32_t rel, uint32_t strand) {
    if (pass) {
        if (sb.n_lane < SEED_LANE_CAP) { sl.code[sb.n_lane][threadIdx.x] = canon; sl.pos[sb.n_lane][threadIdx.x] = (uint16_t)(rel | (strand << 15)); }
        else {  // lane buffer full (P ~ 1e-4 per lane at c = 125; always at c = 1): straight to the global list
            const uint32_t at = sb.rem_first + rel, ch = at / so.chunk;
            uint32_t idx = atomicAdd(&so.seed_count[sb.g], 1u);
            if (idx < sb.scap) { so.seed_code[sb.sstart + idx] = canon; so.seed_loc[sb.sstart + idx] = seed_loc(sb.ch_first + ch, strand, at - ch * so.chunk); }
            atomicAdd(&sb.ctot[sb.ch_first + ch], 1u);
        }
        sb.n_lane++;
    }
}

// call after the position loop by every thread of the block (contains __syncthreads)
__device__ __forceinline__ void seed_block_flush(SeedLds &sl, const SeedOut &so, SeedBlock &sb) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t mine = min(sb.n_lane, SEED_LANE_CAP);
    uint32_t incl = mine;  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += v;
    }
    if (lane == 63) sl.wave_n[w] = incl;
    __syncthreads();
    // one global atomic per block reserves room for all waves' seeds
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t i = 0; i < SEED_WAVES; i++) { sl.wave_base[i] = tot; tot += sl.wave_n[i]; }
        const uint32_t base = tot ? atomicAdd(&so.seed_count[sb.g], tot) : 0u;
        for (uint32_t i = 0; i < SEED_WAVES; i++) sl.wave_base[i] += base;
    }
    __syncthreads();
    uint32_t idx = sl.wave_base[w] + incl - mine;
    for (uint32_t i = 0; i < mine; i++, idx++) {
        const uint32_t pv = sl.pos[i][threadIdx.x], at = sb.rem_first + (pv & 0x7fffu), chrel = at / so.chunk;
        if (idx < sb.scap) {
            so.seed_code[sb.sstart + idx] = sl.code[i][threadIdx.x];
            so.seed_loc[sb.sstart + idx] = seed_loc(sb.ch_first + chrel, pv >> 15, at - chrel * so.chunk);
        }
        if (chrel < SEED_LDS_CHUNKS) atomicAdd(&sl.ctot[chrel], 1u);
        else atomicAdd(&sb.ctot[sb.ch_first + chrel], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SEED_LDS_CHUNKS && sl.ctot[threadIdx.x]) atomicAdd(&sb.ctot[sb.ch_first + threadIdx.x], sl.ctot[threadIdx.x]);
}

}  // namespace ghip_seed
