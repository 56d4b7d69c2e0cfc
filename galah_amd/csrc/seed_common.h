// Device code shared by the standalone seeding pass (ani.hip: ani_seeds) and the fused
// MinHash + seeding pass (sketch.hip: sketch_kmers<K, true>).
#pragma once
#include "ghip_internal.h"

namespace ghip_seed {

// Seed-selection hash: MurmurHash3 fmix32, a bijection on the 32-bit canonical code (see
// oracle/galah_oracle_ani.c).  8 instructions per position; every position of every genome pays it.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// A..T -> 0..3, anything else -> 4
__device__ __forceinline__ uint32_t base_code(uint32_t c) {
    uint32_t d = c - 0x41u;
    bool ok = d < 20u && ((0x80045u >> d) & 1u);
    return ok ? (((c >> 1) ^ (c >> 2)) & 3u) : 4u;
}

constexpr uint32_t SEED_LDS_CAP = 1024;   // seeds buffered per block (expected 16384/c ~ 131)
constexpr uint32_t SEED_LDS_CHUNKS = 64;  // per-block chunk counters
constexpr uint32_t SEED_WAVES = GHIP_SKETCH_THREADS / 64;
constexpr uint32_t SEED_WAVE_CAP = SEED_LDS_CAP / SEED_WAVES;

struct SeedOut {  // where a block's seeds go (kernel argument, by value)
    uint32_t k, thr, chunk;
    uint32_t *seed_code;
    uint16_t *seed_chunk;
    const uint64_t *seed_start;
    uint32_t *seed_count;
    uint32_t *chunk_total;
    const uint64_t *chunk_start;
};

struct SeedLds {  // per-block LDS state; declare as `__shared__ ghip_seed::SeedLds`
    uint32_t code[SEED_LDS_CAP];
    uint16_t pos[SEED_LDS_CAP];
    uint32_t ctot[SEED_LDS_CHUNKS];
    uint32_t wave_n[SEED_WAVES], wave_base[SEED_WAVES];
};

struct SeedBlock {  // per-thread view of one block's seeding state
    uint32_t g, ch_first, rem_first, scap, wave_base, wave_n;
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
    sb.wave_base = (threadIdx.x >> 6) * SEED_WAVE_CAP;
    sb.wave_n = 0;
    if (threadIdx.x < SEED_LDS_CHUNKS) sl.ctot[threadIdx.x] = 0;
    __syncthreads();
    return sb;
}

// Wave-private append: rank among the passing lanes by ballot/mbcnt, wave-uniform count in a scalar --
// no LDS atomic in the position loop.  Must be reached by all 64 lanes.  rel = position - block start.
__device__ __forceinline__ void seed_append(SeedLds &sl, const SeedOut &so, SeedBlock &sb, bool pass, uint32_t canon, uint32_t rel) {
    const unsigned long long m = __ballot(pass);
    if (m) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t slot = sb.wave_n + rank;
        if (pass) {
            if (slot < SEED_WAVE_CAP) { sl.code[sb.wave_base + slot] = canon; sl.pos[sb.wave_base + slot] = (uint16_t)rel; }
            else {  // wave buffer full (never at c=125): straight to the global list
                const uint32_t ch = sb.ch_first + (sb.rem_first + rel) / so.chunk;
                uint32_t idx = atomicAdd(&so.seed_count[sb.g], 1u);
                if (idx < sb.scap) { so.seed_code[sb.sstart + idx] = canon; so.seed_chunk[sb.sstart + idx] = (uint16_t)ch; }
                atomicAdd(&sb.ctot[ch], 1u);
            }
        }
        sb.wave_n += (uint32_t)__popcll(m);
    }
}

// call after the position loop by every thread of the block (contains __syncthreads)
__device__ __forceinline__ void seed_block_flush(SeedLds &sl, const SeedOut &so, SeedBlock &sb) {
    if ((threadIdx.x & 63u) == 0) sl.wave_n[threadIdx.x >> 6] = min(sb.wave_n, SEED_WAVE_CAP);
    __syncthreads();
    // one global atomic per block reserves room for all waves' seeds; then every wave flushes its own
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < SEED_WAVES; w++) { sl.wave_base[w] = tot; tot += sl.wave_n[w]; }
        const uint32_t base = tot ? atomicAdd(&so.seed_count[sb.g], tot) : 0u;
        for (uint32_t w = 0; w < SEED_WAVES; w++) sl.wave_base[w] += base;
    }
    __syncthreads();
    {
        const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        const uint32_t nw = sl.wave_n[w], gbase = sl.wave_base[w];
        for (uint32_t i = lane; i < nw; i += 64) {
            const uint32_t chrel = (sb.rem_first + sl.pos[sb.wave_base + i]) / so.chunk;
            const uint32_t idx = gbase + i;
            if (idx < sb.scap) { so.seed_code[sb.sstart + idx] = sl.code[sb.wave_base + i]; so.seed_chunk[sb.sstart + idx] = (uint16_t)(sb.ch_first + chrel); }
            if (chrel < SEED_LDS_CHUNKS) atomicAdd(&sl.ctot[chrel], 1u);
            else atomicAdd(&sb.ctot[sb.ch_first + chrel], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < SEED_LDS_CHUNKS && sl.ctot[threadIdx.x]) atomicAdd(&sb.ctot[sb.ch_first + threadIdx.x], sl.ctot[threadIdx.x]);
}

}  // namespace ghip_seed
