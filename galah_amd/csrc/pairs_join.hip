// Pair stage, join form (large N): the same result as the dense all-vs-all kernels (pairs_probe.hip, pairs.hip) from
// an inverted index instead of N^2/2 sketch-pair probes.
//
// raw_distance (reference src/finch.rs:79 -> finch::distance::raw_distance) needs |A n B| and two ranks per pair, and a
// pair can only reach the precluster threshold if it shares hashes (common >= cmin[total] >= 1 for every total >= 1
// whenever min_ani > 0).  So instead of probing 5x10^7 sketch pairs at N = 10 000 (5x10^10 set probes), all N*s
// (hash, genome) elements are partitioned by 16-20 bits of the hash (two MSD passes, 8 + 8..12 bits, the bucket count
// sized to ~256 elements per bucket: LDS histograms, one
// global atomic per digit and block, unordered scatter -- grouping is all that is needed, not order), equal hashes
// inside a bucket yield one record (i, j) per sharing genome pair, the records are partitioned the same
// way by a hash of (i, j), and a per-bucket LDS table counts them: that count IS common.  Ranks i = #{a <= m},
// j = #{b <= m} come from binary searches in the sorted rows, the integer filter common >= cmin[total] and the host's
// exact f64 recheck are those of the dense path.  Work ~ N*s + #records instead of N^2*s.
//
// Two inputs that used to make the join decline are handled around it instead.  Empty sketches (by the reference's
// NaN semantics their ANI with every other sketch is 1.0) take no part: the caller passes their indices (`empties`)
// and lists their N - 1 pairs itself.  Large families (many genomes sharing each hash: a hash run beyond J_RUN_MAX
// genomes, or a bucket beyond the LDS stage) go hybrid: a block per oversized bucket (join_elem_pairs_big_kernel)
// marks the genomes of the long runs in `big`, the join drops the pairs whose two genomes are both marked, and the
// caller counts exactly those pairs with a dense kernel over the marked rows (api_pairs.cpp, precluster_impl).
//
// The join form still declines (the caller then runs a dense kernel over everything) when it would not pay or cannot
// be exact in its fixed-size buffers: min_ani <= 0 (pairs without a common hash qualify), a bucket beyond J_BIG_CAP
// elements or more than J_BIG_LIST oversized buckets, a record table overflow, more records than a dense pass would cost,
// N*s >= 2^32.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "ghip_internal.h"

namespace {

constexpr uint32_t J_THREADS = 256, J_PER = 16, J_TILE = J_THREADS * J_PER;  // elements per partition block
constexpr uint32_t J_BITS2_MIN = 8, J_BITS2_MAX = 12;   // buckets = 256 << bits2: 65 536 .. 1 048 576, sized to the input
constexpr uint32_t J_BITS2_MIN_ELEM = 6;   // the element buckets may go down to 16 384: a wave per bucket wants ~100+ elements
                                           // (2 000 genomes: 1.34 -> 0.9 ms); the record buckets keep 65 536 for table headroom
constexpr uint32_t J_D2_MAX = 1u << J_BITS2_MAX;
constexpr uint32_t J_ELEM_CAP = 1024;   // elements of one hash bucket staged per wave (expected N*s/65536)
constexpr uint32_t J_TAB = 256;         // distinct genome pairs counted per record bucket
constexpr uint32_t J_RUN_MAX = 256;     // a hash shared by more genomes than this is left to the caller's dense pass (`big`)
constexpr uint32_t J_BIG_CAP = 8192;    // elements of a bucket too large for a wave's stage, sorted by a whole block instead
constexpr uint32_t J_BIG_LIST = 65536;  // such buckets per launch
constexpr uint32_t J_WAVES = J_THREADS / 64;
constexpr uint64_t J_EMPTY = ~0ull;

struct ElemSrc {  // the packed sketch matrix as (hash, genome) elements; padded slots are skipped
    const uint64_t *hashes;
    const uint32_t *lens;
    uint32_t s;
    uint32_t total;  // n * s < 2^32
    uint32_t hrank, hworld;   // hash-sharded multi-rank form: only the hashes whose first-level digit d has d % hworld == hrank
    __device__ bool get(uint32_t t, uint64_t &key, uint32_t &val) const {
        if (t >= total) return false;
        const uint32_t g = t / s, r = t - g * s;
        if (r >= lens[g]) return false;
        key = hashes[t];
        val = g;
        return hworld <= 1 || (mix(key) >> 12) % hworld == hrank;
    }
    __device__ static uint32_t mix(uint64_t key) { return (uint32_t)key & 0xfffffu; }  // 20 MurmurHash3 low bits: uniform
};

struct JoinEntry { uint64_t key; uint32_t count, pad; };   // partial common of one genome pair (hash-sharded form); 16 bytes
static_assert(sizeof(JoinEntry) == GHIP_JOIN_ENTRY_BYTES, "entry size is part of the exchange layout");

struct EntrySrc {  // the gathered partial counts of all ranks as weighted records; only the pairs this rank owns
    const JoinEntry *e;
    uint32_t total, rank, world;
    __device__ bool get(uint32_t t, uint64_t &key, uint32_t &val) const {
        if (t >= total) return false;
        key = e[t].key;
        val = e[t].count;
        return key != ~0ull && ((uint32_t)(key >> 32) + (uint32_t)key) % world == rank;   // (padding entries hold 2^64 - 1)
    }
    __device__ static uint32_t mix(uint64_t key) {
        return (((uint32_t)(key >> 32) * 0x9E3779B1u) ^ ((uint32_t)key * 0x85EBCA77u)) >> 12;  // 20 bits (as RecSrc)
    }
};

struct RecSrc {  // genome-pair records (i << 32 | j), i < j
    const uint64_t *rec;
    uint32_t total;
    __device__ bool get(uint32_t t, uint64_t &key, uint32_t &val) const {
        if (t >= total) return false;
        key = rec[t];
        val = 0;
        return true;
    }
    __device__ static uint32_t mix(uint64_t key) {
        return (((uint32_t)(key >> 32) * 0x9E3779B1u) ^ ((uint32_t)key * 0x85EBCA77u)) >> 12;  // 20 bits
    }
};

// mix() yields 20 bits: the first-level digit is bits 19..12, the second-level digit the next bits2 bits below.
__device__ __forceinline__ uint32_t digit1(uint32_t m) { return m >> 12; }
__device__ __forceinline__ uint32_t digit2(uint32_t m, uint32_t bits2) { return (m >> (12 - bits2)) & ((1u << bits2) - 1); }

// ---- MSD pass 1 -------------------------------------------------------------------------------------------------
template <typename Src>
__global__ __launch_bounds__(J_THREADS) void join_hist1_kernel(Src src, uint32_t *__restrict__ hist1) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * J_TILE;
#pragma unroll 4
    for (uint32_t u = 0; u < J_PER; u++) {
        uint64_t key; uint32_t val;
        if (src.get(base + u * J_THREADS + threadIdx.x, key, val)) atomicAdd(&h[digit1(Src::mix(key))], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist1[threadIdx.x], h[threadIdx.x]);
}

// exclusive scan of 256 counters -> start[257]
__global__ __launch_bounds__(256) void join_scan256_kernel(const uint32_t *__restrict__ hist, uint32_t *__restrict__ start) {
    __shared__ uint32_t v[256];
    v[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; i++) { const uint32_t c = v[i]; start[i] = run; run += c; }
        start[256] = run;
    }
}

// start1 == nullptr: the CAPACITY form (fused partition) -- first-level bucket d owns slots [d * cap1, (d + 1) * cap1) and
// the per-digit cursors ARE the counts afterwards: no histogram pass, no scan, no host round trip before the second level.
// An element beyond its bucket's capacity is dropped and flags bit 3 raised: the caller repeats the join in the exact form.
template <typename Src>
__global__ __launch_bounds__(J_THREADS) void join_scatter1_kernel(Src src, const uint32_t *__restrict__ start1,
                                                                  uint32_t *__restrict__ cursor1, uint32_t cap1, uint32_t *__restrict__ flags,
                                                                  uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_vals) {
    __shared__ uint32_t h[256], base[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t t0 = blockIdx.x * J_TILE;
    uint64_t keys[J_PER];
    uint32_t vals[J_PER];
    uint32_t okmask = 0;
#pragma unroll
    for (uint32_t u = 0; u < J_PER; u++) {
        if (src.get(t0 + u * J_THREADS + threadIdx.x, keys[u], vals[u])) {
            okmask |= 1u << u;
            atomicAdd(&h[digit1(Src::mix(keys[u]))], 1u);
        }
    }
    __syncthreads();
    __shared__ uint32_t lim[256];   // capacity form: one past the last slot of the bucket (exact form: no limit)
    {   // one global atomic per digit and block reserves the block's range in that bucket
        const uint32_t c = h[threadIdx.x];
        uint32_t b = 0, l = 0xffffffffu;
        if (c) {
            const uint32_t at = atomicAdd(&cursor1[threadIdx.x], c);
            if (start1) b = start1[threadIdx.x] + at;
            else {
                b = threadIdx.x * cap1 + at;
                l = (threadIdx.x + 1) * cap1;
                if (at + c > cap1) atomicOr(flags, 8u);
            }
        }
        base[threadIdx.x] = b; lim[threadIdx.x] = l;
    }
    __syncthreads();
    h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < J_PER; u++) {
        if (okmask & (1u << u)) {
            const uint32_t d = digit1(Src::mix(keys[u]));
            const uint32_t pos = base[d] + atomicAdd(&h[d], 1u);
            if (pos < lim[d]) {
                out_keys[pos] = keys[u];
                if (out_vals) out_vals[pos] = vals[u];
            }
        }
    }
}

// first-level bucket d1 of the intermediate array: [start1[d1], start1[d1 + 1]) in the exact form, the first
// min(count1[d1], cap1) slots of [d1 * cap1, ...) in the capacity form (count1 != nullptr)
__device__ __forceinline__ void join_bucket1(const uint32_t *start1, const uint32_t *count1, uint32_t cap1, uint32_t d1, uint32_t &lo, uint32_t &end) {
    if (count1) { lo = d1 * cap1; end = lo + min(count1[d1], cap1); }
    else { lo = start1[d1]; end = start1[d1 + 1]; }
}

// ---- MSD pass 2: inside first-level bucket blockIdx.y, digit = the next bits2 bits --------------------------------
template <typename Src>
__global__ __launch_bounds__(J_THREADS) void join_hist2_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ start1,
                                                               const uint32_t *__restrict__ count1, uint32_t cap1,
                                                               uint32_t bits2, uint32_t *__restrict__ hist2) {
    const uint32_t d1 = blockIdx.y, nd2 = 1u << bits2;
    uint32_t lo, end;
    join_bucket1(start1, count1, cap1, d1, lo, end);
    lo += blockIdx.x * J_TILE;
    if (lo >= end) return;
    const uint32_t hi = min(end, lo + J_TILE);
    __shared__ uint32_t h[J_D2_MAX];
    for (uint32_t d = threadIdx.x; d < nd2; d += J_THREADS) h[d] = 0;
    __syncthreads();
    for (uint32_t t = lo + threadIdx.x; t < hi; t += J_THREADS) atomicAdd(&h[digit2(Src::mix(keys[t]), bits2)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nd2; d += J_THREADS)
        if (h[d]) atomicAdd(&hist2[(d1 << bits2) + d], h[d]);
}

// Exclusive scan of nb counters (nb a multiple of J_SCAN_TILE, at most 256 tiles) -> start[nb + 1], and the largest
// counter.  Two small launches: per-tile sums, then every tile scans the (<= 256) tile sums in LDS for its own offset
// and writes its part -- coalesced 16-byte accesses throughout.  (One 1024-thread block walking the array with a
// 256-byte stride per lane took 97 us for 65 536 counters, three times per join call.)
constexpr uint32_t J_SCAN_PER = 16, J_SCAN_TILE = J_THREADS * J_SCAN_PER;

__global__ __launch_bounds__(J_THREADS) void join_scan_sums_kernel(const uint32_t *__restrict__ hist, uint32_t *__restrict__ tile_sum,
                                                                   uint32_t *__restrict__ tile_max) {
    __shared__ uint32_t ws[J_WAVES], wm[J_WAVES];
    const uint4 *src = reinterpret_cast<const uint4 *>(hist + (size_t)blockIdx.x * J_SCAN_TILE) + threadIdx.x * (J_SCAN_PER / 4);
    uint32_t sum = 0, m = 0;
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER / 4; q++) {
        const uint4 v = src[q];
        sum += v.x + v.y + v.z + v.w;
        m = max(max(m, max(v.x, v.y)), max(v.z, v.w));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off, 64); m = max(m, (uint32_t)__shfl_xor(m, off, 64)); }
    if ((threadIdx.x & 63u) == 0) { ws[threadIdx.x >> 6] = sum; wm[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, x = 0;
        for (uint32_t w = 0; w < J_WAVES; w++) { t += ws[w]; x = max(x, wm[w]); }
        tile_sum[blockIdx.x] = t;
        tile_max[blockIdx.x] = x;
    }
}

__global__ __launch_bounds__(J_THREADS) void join_scan_write_kernel(const uint32_t *__restrict__ hist, const uint32_t *__restrict__ tile_sum,
                                                                    const uint32_t *__restrict__ tile_max, uint32_t n_tiles,
                                                                    uint32_t *__restrict__ start, uint32_t *__restrict__ max_out) {
    __shared__ uint32_t ts[256], tm[256], ws[J_WAVES];
    // offset of this tile = sum of the tiles before it (n_tiles <= 256: one element per thread)
    ts[threadIdx.x] = threadIdx.x < n_tiles ? tile_sum[threadIdx.x] : 0u;
    tm[threadIdx.x] = threadIdx.x < n_tiles ? tile_max[threadIdx.x] : 0u;
    __syncthreads();
    for (uint32_t off = 128; off > 0; off >>= 1) {   // largest counter of all tiles -> tm[0]
        __syncthreads();
        if (threadIdx.x < off) tm[threadIdx.x] = max(tm[threadIdx.x], tm[threadIdx.x + off]);
    }
    __syncthreads();
    uint32_t before = 0, total = 0;
    {   // every thread adds a strided share of the tile sums, then a wave + block reduction
        uint32_t mine_before = 0, mine_all = 0;
        for (uint32_t t = threadIdx.x; t < n_tiles; t += J_THREADS) { mine_all += ts[t]; if (t < blockIdx.x) mine_before += ts[t]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mine_before += __shfl_xor(mine_before, off, 64); mine_all += __shfl_xor(mine_all, off, 64); }
        __shared__ uint32_t rb[J_WAVES], ra[J_WAVES];
        if ((threadIdx.x & 63u) == 0) { rb[threadIdx.x >> 6] = mine_before; ra[threadIdx.x >> 6] = mine_all; }
        __syncthreads();
        for (uint32_t w = 0; w < J_WAVES; w++) { before += rb[w]; total += ra[w]; }
    }
    const size_t base = (size_t)blockIdx.x * J_SCAN_TILE + threadIdx.x * J_SCAN_PER;
    uint32_t v[J_SCAN_PER];
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER / 4; q++) {
        const uint4 x = reinterpret_cast<const uint4 *>(hist + base)[q];
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
    }
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER; q++) sum += v[q];
    uint32_t incl = sum;   // inclusive scan of the thread sums over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63u) >= (uint32_t)off) incl += u;
    }
    if ((threadIdx.x & 63u) == 63u) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = before + incl - sum;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += ws[w];
    uint32_t o[J_SCAN_PER];
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER; q++) { o[q] = run; run += v[q]; }
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER / 4; q++)
        reinterpret_cast<uint4 *>(start + base)[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { start[(size_t)n_tiles * J_SCAN_TILE] = total; if (max_out) *max_out = tm[0]; }
    // 32-bit offsets: the host bounds the inputs (n*s < 2^32 elements, < 2^31 records from a 64-bit total)
}

constexpr uint32_t J_BARRIER_VOID = 0x80000000u;  // in the arrival word of a fused scan: a waiter has given up, the scan is void (the count lives in the bits below)
constexpr uint32_t J_BARRIER_POLLS = 1u << 18;   // polls of the grid barrier before a waiter gives up (an agent-scope load + s_sleep 2: ~1 us each)

// The same scan in ONE launch (fused form): per-tile sums, a grid barrier, the write.  The grid is n_tiles <= 256 blocks of
// 256 threads with no dynamic LDS -- a fraction of what the chip holds at once, so every block is (or becomes, as soon as a
// co-running kernel's blocks retire) resident and the barrier is not expected to starve (and gives up if it does: J_BARRIER_POLLS); the tile sums cross the XCDs through
// agent-scope atomics.  `arrive` is zeroed by the caller (one word per scan of a join call).  hist_b (nullable): a second
// counter array that is only summed; totals (nullable) = {sum of hist_b, sum of hist} as 64-bit numbers -- what
// join_totals_kernel delivered in a launch of its own.  The counters are read ONCE (they wait in registers across the barrier).
__global__ __launch_bounds__(J_THREADS) void join_scan_fused_kernel(const uint32_t *__restrict__ hist, const uint32_t *__restrict__ hist_b,
                                                                    uint32_t *__restrict__ tile_sum, uint32_t *__restrict__ tile_max,
                                                                    unsigned long long *__restrict__ tile_sum_b, uint32_t *__restrict__ arrive,
                                                                    uint32_t n_tiles, uint32_t *__restrict__ start, uint32_t *__restrict__ max_out,
                                                                    unsigned long long *__restrict__ totals, uint32_t *__restrict__ flags) {
    __shared__ uint32_t ws[J_WAVES], wm[J_WAVES], ts[256], tm[256];
    __shared__ unsigned long long wb[J_WAVES];
    __shared__ uint32_t void_scan;
    const size_t base = (size_t)blockIdx.x * J_SCAN_TILE + threadIdx.x * J_SCAN_PER;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t v[J_SCAN_PER];
    uint32_t sum = 0, m = 0;
    unsigned long long sum_b = 0;
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER / 4; q++) {
        const uint4 x = reinterpret_cast<const uint4 *>(hist + base)[q];
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        sum += x.x + x.y + x.z + x.w;
        m = max(max(m, max(x.x, x.y)), max(x.z, x.w));
        if (hist_b) { const uint4 y = reinterpret_cast<const uint4 *>(hist_b + base)[q]; sum_b += (unsigned long long)y.x + y.y + y.z + y.w; }
    }
    uint32_t incl = sum;   // inclusive scan of the thread sums over the wave (kept for the write phase)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m = max(m, (uint32_t)__shfl_xor(m, off, 64)); sum_b += __shfl_xor(sum_b, off, 64); }
    if (lane == 63u) ws[wave] = incl;
    if (lane == 0u) { wm[wave] = m; wb[wave] = sum_b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, x = 0;
        unsigned long long b = 0;
        for (uint32_t w = 0; w < J_WAVES; w++) { t += ws[w]; x = max(x, wm[w]); b += wb[w]; }
        __hip_atomic_store(&tile_sum[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&tile_max[blockIdx.x], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&tile_sum_b[blockIdx.x], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // the three stores above are ordered before it
        // A BOUNDED wait: should a block of this grid not become resident while the others spin -- a co-running
        // kernel that holds the chip for longer than any of this library's does -- a waiter gives up after ~0.3 s of polls.
        // Giving up VOIDS the scan for every block, those that have not started included: the waiter marks the arrival word
        // (J_BARRIER_VOID, by compare-and-swap against the count it saw: a count that completes meanwhile wins and nobody
        // gives up), no block passes a marked word, every block writes zeros for its offsets -- all buckets empty, so the
        // kernels queued behind this one find nothing to do instead of offsets made of partial sums -- and flags bit 3 (what an
        // outgrown capacity raises) has the host discard this form's output and repeat the join in the exact form.
        uint32_t polls = 0, seen = __hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        while (!(seen & J_BARRIER_VOID) && seen < n_tiles) {
            if (++polls > J_BARRIER_POLLS) {
                uint32_t expect = seen;
                if (__hip_atomic_compare_exchange_strong(arrive, &expect, seen | J_BARRIER_VOID, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
                    atomicOr(flags, 8u);
                    seen |= J_BARRIER_VOID;
                    break;
                }
                seen = expect;   // the count moved on: look again
                continue;
            }
            __builtin_amdgcn_s_sleep(2);
            seen = __hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        }
        void_scan = (seen & J_BARRIER_VOID) ? 1u : 0u;
    }
    __syncthreads();
    if (void_scan) {   // (block-uniform: thread 0 wrote it in front of the barrier)
#pragma unroll
        for (uint32_t q = 0; q < J_SCAN_PER / 4; q++) reinterpret_cast<uint4 *>(start + base)[q] = make_uint4(0, 0, 0, 0);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            start[(size_t)n_tiles * J_SCAN_TILE] = 0;
            if (max_out) *max_out = 0;
            if (totals) { totals[0] = 0; totals[1] = 0; }
        }
        return;
    }
    ts[threadIdx.x] = threadIdx.x < n_tiles ? __hip_atomic_load(&tile_sum[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    tm[threadIdx.x] = threadIdx.x < n_tiles ? __hip_atomic_load(&tile_max[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t t = 0; t < blockIdx.x; t++) before += ts[t];   // <= 255 LDS broadcasts
    uint32_t run = before + incl - sum;
    for (uint32_t w = 0; w < wave; w++) run += ws[w];
    uint32_t o[J_SCAN_PER];
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER; q++) { o[q] = run; run += v[q]; }
#pragma unroll
    for (uint32_t q = 0; q < J_SCAN_PER / 4; q++)
        reinterpret_cast<uint4 *>(start + base)[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long total = 0, total_b = 0;
        uint32_t x = 0;
        for (uint32_t t = 0; t < n_tiles; t++) {
            total += ts[t]; x = max(x, tm[t]);
            total_b += __hip_atomic_load(&tile_sum_b[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        start[(size_t)n_tiles * J_SCAN_TILE] = (uint32_t)total;   // (32-bit offsets: the host bounds the inputs and declines on a 64-bit total beyond them)
        if (max_out) *max_out = x;
        if (totals) { totals[0] = total_b; totals[1] = total; }
    }
}

template <typename Src>
__global__ __launch_bounds__(J_THREADS) void join_scatter2_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                                  const uint32_t *__restrict__ start1, const uint32_t *__restrict__ count1, uint32_t cap1,
                                                                  uint32_t bits2,
                                                                  const uint32_t *__restrict__ start2, uint32_t *__restrict__ cursor2,
                                                                  uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_vals) {
    const uint32_t d1 = blockIdx.y, nd2 = 1u << bits2;
    uint32_t lo, end;
    join_bucket1(start1, count1, cap1, d1, lo, end);
    lo += blockIdx.x * J_TILE;
    if (lo >= end) return;
    const uint32_t hi = min(end, lo + J_TILE);
    __shared__ uint32_t h[J_D2_MAX], base[J_D2_MAX];
    for (uint32_t d = threadIdx.x; d < nd2; d += J_THREADS) h[d] = 0;
    __syncthreads();
    for (uint32_t t = lo + threadIdx.x; t < hi; t += J_THREADS) atomicAdd(&h[digit2(Src::mix(keys[t]), bits2)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nd2; d += J_THREADS) {
        const uint32_t c = h[d], b = (d1 << bits2) + d;
        base[d] = c ? start2[b] + atomicAdd(&cursor2[b], c) : 0u;
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nd2; d += J_THREADS) h[d] = 0;
    __syncthreads();
    for (uint32_t t = lo + threadIdx.x; t < hi; t += J_THREADS) {
        const uint64_t k = keys[t];
        const uint32_t d = digit2(Src::mix(k), bits2);
        const uint32_t pos = base[d] + atomicAdd(&h[d], 1u);
        out_keys[pos] = k;
        if (out_vals) out_vals[pos] = vals[t];
    }
}

// ---- equal hashes inside a bucket -> one record per sharing genome pair ------------------------------------------
// One wavefront per bucket.  The counting pass (EMIT = false) sorts the bucket by hash in LDS -- a bitonic
// network over the next power of two, 64 lanes -- writes it back sorted and counts, for every element, the later
// elements of its run of equal hashes; the emitting pass (EMIT = true) reads the sorted bucket and writes those pairs
// at rec_start[bucket].  (Comparing every element with every later one cost cnt^2 / 2 dependent LDS reads per bucket:
// 0.85 + 0.56 ms of the 2.6 ms join at 8 000 genomes.)  flags bit 0: a bucket exceeds the LDS stage.
// Does element a of the sorted stage k[0..cnt) belong to a run of more than J_RUN_MAX equal hashes?  Two reads screen
// out everything shorter than half of that (a longer run reaches J_RUN_MAX / 2 places one way or the other from any of
// its members); the exact length, by two binary searches, is only taken behind the screen.
__device__ __forceinline__ bool join_in_long_run(const uint64_t *k, uint32_t cnt, uint32_t a, uint64_t ka) {
    constexpr uint32_t H = J_RUN_MAX / 2;
    const bool fwd = a + H < cnt && k[a + H] == ka, bwd = a >= H && k[a - H] == ka;
    if (!fwd && !bwd) return false;
    uint32_t lo = 0, hi = a;            // first index holding ka
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (k[mid] < ka) lo = mid + 1; else hi = mid; }
    uint32_t lo2 = a + 1, hi2 = cnt;    // one past the last
    while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2) >> 1; if (k[mid] <= ka) lo2 = mid + 1; else hi2 = mid; }
    return lo2 - lo > J_RUN_MAX;
}

// The buckets no wave's stage holds (up to J_BIG_CAP elements: a hash shared by a thousand genomes, next to the bucket's
// ordinary ones), one per BLOCK: the same sort, the same run scan, the same long-run rule as join_elem_pairs_kernel.
template <bool EMIT>
__global__ __launch_bounds__(J_THREADS) void join_elem_pairs_big_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                                        const uint32_t *__restrict__ start2, uint32_t *__restrict__ rec_count,
                                                                        const uint32_t *__restrict__ rec_start, uint64_t *__restrict__ rec,
                                                                        uint32_t *__restrict__ flags, uint32_t *__restrict__ all_count, uint32_t rank,
                                                                        uint32_t world, uint32_t row_lo, uint8_t *__restrict__ big,
                                                                        const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ big_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char jraw[];
    uint64_t *k = reinterpret_cast<uint64_t *>(jraw);
    uint32_t *v = reinterpret_cast<uint32_t *>(jraw + (size_t)J_BIG_CAP * sizeof(uint64_t));
    __shared__ uint32_t n_found, n_all, n_marked;
    const uint32_t listed = min(*big_n, J_BIG_LIST);
    for (uint32_t bi = blockIdx.x; bi < listed; bi += gridDim.x) {
        const uint32_t bucket = big_list[bi];
        const uint32_t lo = start2[bucket], cnt = start2[bucket + 1] - lo;
        __syncthreads();   // the previous bucket's stage is no longer read
        for (uint32_t e = threadIdx.x; e < cnt; e += J_THREADS) { k[e] = keys[lo + e]; v[e] = vals[lo + e]; }
        if (threadIdx.x == 0) { n_found = 0; n_all = 0; n_marked = 0; }
        if (!EMIT) {
            uint32_t P = 64;
            while (P < cnt) P <<= 1;
            auto exchange = [&](uint32_t i, uint32_t x) {
                if (x >= cnt) return;
                const uint64_t ka = k[i], kb = k[x];
                if (ka > kb) { const uint32_t va = v[i], vb = v[x]; k[i] = kb; k[x] = ka; v[i] = vb; v[x] = va; }
            };
            for (uint32_t kk = 2; kk <= P; kk <<= 1) {
                __syncthreads();
                const uint32_t half = kk >> 1;
                for (uint32_t t = threadIdx.x; t < P / 2; t += J_THREADS) { const uint32_t i = 2 * t - (t & (half - 1)); exchange(i, i ^ (kk - 1)); }
                for (uint32_t j = half >> 1; j > 0; j >>= 1) {
                    __syncthreads();
                    for (uint32_t t = threadIdx.x; t < P / 2; t += J_THREADS) { const uint32_t i = 2 * t - (t & (j - 1)); exchange(i, i + j); }
                }
            }
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < cnt; e += J_THREADS) { keys[lo + e] = k[e]; vals[lo + e] = v[e]; }   // the emitting pass reads it sorted
        }
        __syncthreads();
        const uint32_t out0 = EMIT ? rec_start[bucket] : 0u;
        uint32_t found = 0, found_all = 0;
        bool marked = false;
        for (uint32_t a = threadIdx.x; a < cnt; a += J_THREADS) {
            const uint64_t ka = k[a];
            const uint32_t ga = v[a];
            if (join_in_long_run(k, cnt, a, ka)) {
                if (!EMIT) { __hip_atomic_store(&big[ga], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); marked = true; }   // (several workgroups may mark one genome)
                continue;
            }
            for (uint32_t b = a + 1; b < cnt && k[b] == ka; b++) {
                const uint32_t gb = v[b];
                if (gb == ga) continue;
                found_all++;
                if (world > 1 && (ga + gb) % world != rank) continue;
                if (max(ga, gb) < row_lo) continue;
                if (EMIT) rec[out0 + atomicAdd(&n_found, 1u)] = ((uint64_t)min(ga, gb) << 32) | max(ga, gb);
                else found++;
            }
        }
        if (!EMIT) {
            if (found) atomicAdd(&n_found, found);
            if (found_all) atomicAdd(&n_all, found_all);
            if (marked) atomicOr(&n_marked, 1u);
            __syncthreads();
            if (threadIdx.x == 0) {
                rec_count[bucket] = n_found; all_count[bucket] = n_all;
                if (n_marked) atomicOr(flags, 4u);
            }
        }
    }
}

template <bool EMIT>
__global__ __launch_bounds__(J_THREADS) void join_elem_pairs_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                                    const uint32_t *__restrict__ start2,
                                                                    uint32_t *__restrict__ rec_count, const uint32_t *__restrict__ rec_start,
                                                                    uint64_t *__restrict__ rec, uint32_t *__restrict__ flags,
                                                                    uint32_t *__restrict__ all_count, uint32_t rank, uint32_t world, uint32_t row_lo,
                                                                    uint32_t elem_cap, uint8_t *__restrict__ big, uint32_t *__restrict__ big_list,
                                                                    uint32_t *__restrict__ big_n) {
    // elem_cap elements per wave: keys, then genome ids (dynamic LDS: a launch sized for the buckets it expects keeps eight
    // blocks on a CU; with room for J_ELEM_CAP elements per wave three fit, and the sort is bound by its steps' latency)
    extern __shared__ __attribute__((aligned(16))) unsigned char jraw[];
    __shared__ uint32_t lcnt[J_WAVES];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t bucket = blockIdx.x * J_WAVES + wave;
    const uint32_t lo = start2[bucket], cnt = start2[bucket + 1] - lo;
    if (cnt > elem_cap) {
        // no stage of a wave holds the bucket.  With `big` (the last attempt of a caller that can finish the pairs of very
        // large families densely) a bucket of up to J_BIG_CAP elements goes to join_elem_pairs_big_kernel, a block per
        // bucket; anything else raises flags bit 0: the caller retries with a larger stage or declines.
        if (big && cnt <= J_BIG_CAP) {
            if (!EMIT && lane == 0) {
                const uint32_t at = atomicAdd(big_n, 1u);
                if (at < J_BIG_LIST) big_list[at] = bucket; else atomicOr(flags, 1u);
            }
        } else if (lane == 0) { atomicOr(flags, 1u); if (!EMIT) { rec_count[bucket] = 0; all_count[bucket] = 0; } }
        return;
    }
    uint64_t *k = reinterpret_cast<uint64_t *>(jraw) + (size_t)wave * elem_cap;
    uint32_t *v = reinterpret_cast<uint32_t *>(jraw + (size_t)J_WAVES * elem_cap * sizeof(uint64_t)) + (size_t)wave * elem_cap;
    for (uint32_t e = lane; e < cnt; e += 64) { k[e] = keys[lo + e]; v[e] = vals[lo + e]; }
    if (lane == 0) lcnt[wave] = 0;
    if (!EMIT && cnt > 1) {
        // Ascending-only bitonic network over the next power of two P: the first step of every merge pairs i with its
        // mirror i ^ (kk - 1), the later ones with i + j, and every exchange moves the smaller hash down -- so
        // positions >= cnt act as +infinity without being stored, and a step whose partner lies there is skipped.
        // Only equal hashes have to end up adjacent: no tie-break, the genome ids are read only when a swap happens.
        uint32_t P = 64;
        while (P < cnt) P <<= 1;
        auto exchange = [&](uint32_t i, uint32_t x) {
            if (x >= cnt) return;
            const uint64_t ka = k[i], kb = k[x];
            if (ka > kb) {
                const uint32_t va = v[i], vb = v[x];
                k[i] = kb; k[x] = ka; v[i] = vb; v[x] = va;
            }
        };
        for (uint32_t kk = 2; kk <= P; kk <<= 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t half = kk >> 1;
            for (uint32_t t = lane; t < P / 2; t += 64) {
                const uint32_t i = 2 * t - (t & (half - 1));   // = kk * (t / half) + t % half, half a power of two (no division)
                exchange(i, i ^ (kk - 1));
            }
            for (uint32_t j = half >> 1; j > 0; j >>= 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t t = lane; t < P / 2; t += 64) {
                    const uint32_t i = 2 * t - (t & (j - 1));   // = 2 j * (t / j) + t % j, j a power of two
                    exchange(i, i + j);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e = lane; e < cnt; e += 64) { keys[lo + e] = k[e]; vals[lo + e] = v[e]; }   // the emitting pass reads it sorted
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Multi-GPU: a rank only materialises the records of ITS genome pairs, (i + j) mod world == rank -- every record
    // of a pair then lives on one rank, so the per-pair counts of the reduce stage are complete there.  The element
    // stage above is the same on every rank; the decisions to decline use rec_total, which counts the records of ALL
    // ranks, so every rank decides alike.
    const uint32_t out0 = EMIT ? rec_start[bucket] : 0u;
    uint32_t found = 0, found_all = 0;
    bool marked = false;
    for (uint32_t a = lane; a < cnt; a += 64) {
        const uint64_t ka = k[a];
        const uint32_t ga = v[a];
        if (big && join_in_long_run(k, cnt, a, ka)) {   // a hash of a very large family: its genomes are marked, it emits nothing
            if (!EMIT) { __hip_atomic_store(&big[ga], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); marked = true; }   // (several workgroups may mark one genome)
            continue;
        }
        for (uint32_t b = a + 1; b < cnt && k[b] == ka; b++) {
            const uint32_t gb = v[b];
            if (gb == ga) continue;  // a sketch row is distinct; guards caller-supplied matrices
            found_all++;
            if (world > 1 && (ga + gb) % world != rank) continue;
            if (max(ga, gb) < row_lo) continue;   // incremental run: a pair of two old genomes is not asked for
            if (EMIT) rec[out0 + atomicAdd(&lcnt[wave], 1u)] = ((uint64_t)min(ga, gb) << 32) | max(ga, gb);
            else found++;
        }
    }
    if (!EMIT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { found += __shfl_xor(found, off, 64); found_all += __shfl_xor(found_all, off, 64); }
        // (the totals are summed from these two arrays by join_totals_kernel: one atomic pair per wave on the same two
        // addresses serialised 131 072 L2 atomics and WAS the kernel's duration, 1.6 ms at 10 000 genomes)
        if (lane == 0) { rec_count[bucket] = found; all_count[bucket] = found_all; }
        if (__builtin_amdgcn_ballot_w64(marked) != 0 && lane == 0) atomicOr(flags, 4u);
    }
}

// ---- records of one bucket -> common per genome pair -> ranks, integer filter, candidate list --------------------
// flags bit 1: more distinct genome pairs in a record bucket than the LDS table holds.
// WEIGHTED: record t counts weight[t] times (the gathered partial counts of the hash-sharded form).  PARTIAL != 0: no ranks,
// no filter -- every (pair, count) of the table is this rank's partial common of that pair: PARTIAL = 1 counts the
// bucket's pairs into ent_count[bucket], PARTIAL = 2 writes them at ent_start[bucket] (count -> scan -> emit, as for the
// records: one returning atomic per entry on a shared counter cost 0.46 ms for 45 000 entries).
template <bool WEIGHTED, int PARTIAL>
__global__ __launch_bounds__(J_THREADS) void join_reduce_kernel(const uint64_t *__restrict__ rec, const uint32_t *__restrict__ weight,
                                                                JoinEntry *__restrict__ entries, uint32_t *__restrict__ ent_count, const uint32_t *__restrict__ ent_start,
                                                                const uint32_t *__restrict__ start2,
                                                                const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens,
                                                                uint32_t s, const uint16_t *__restrict__ cmin, uint32_t cmin_floor,
                                                                uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *__restrict__ out, unsigned long long *__restrict__ out_count,
                                                                uint64_t cap, uint32_t *__restrict__ flags, const uint8_t *__restrict__ big) {
    __shared__ unsigned long long tk[J_WAVES][J_TAB];
    __shared__ uint32_t tc[J_WAVES][J_TAB];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t bucket = blockIdx.x * J_WAVES + wave;
    const uint32_t lo = start2[bucket], hi = start2[bucket + 1];
    if (PARTIAL != 0 && lo == hi) { if (PARTIAL == 1 && lane == 0) ent_count[bucket] = 0; return; }   // (PARTIAL == 0 goes on: block barriers below)
    for (uint32_t e = lane; e < J_TAB; e += 64) { tk[wave][e] = J_EMPTY; tc[wave][e] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lo + lane; t < hi; t += 64) {
        const unsigned long long key = rec[t];
        uint32_t slot = ((uint32_t)(key >> 32) * 0x632BE5ABu + (uint32_t)key * 0x2545F491u) >> 24;  // 8 bits; independent of the bucket hash
        bool done = false;
        for (uint32_t it = 0; it < J_TAB && !done; it++) {
            const unsigned long long prev = atomicCAS(&tk[wave][slot], (unsigned long long)J_EMPTY, key);
            if (prev == J_EMPTY || prev == key) { atomicAdd(&tc[wave][slot], WEIGHTED ? weight[t] : 1u); done = true; }
            else slot = (slot + 1) & (J_TAB - 1);
        }
        if (!done) atomicOr(flags, 2u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (PARTIAL != 0) {
        uint32_t run = 0;   // wave-uniform: entries of this bucket so far
        const uint32_t out0 = PARTIAL == 2 ? ent_start[bucket] : 0u;
        for (uint32_t e = lane; e < J_TAB; e += 64) {
            const unsigned long long key = tk[wave][e];
            const unsigned long long m = __ballot(key != J_EMPTY);
            if (PARTIAL == 2 && key != J_EMPTY) {
                const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                entries[out0 + run + before] = JoinEntry{key, tc[wave][e], 0u};
            }
            run += (uint32_t)__popcll(m);
        }
        if (PARTIAL == 1 && lane == 0) ent_count[bucket] = run;
        return;
    }
    // The pairs that pass the filter are collected per lane (a lane looks at J_TAB / 64 table slots), counted per block
    // in LDS, and the block reserves its run of the output list with ONE global atomic: a returning atomic per listed pair
    // on the one list counter serialised 45 000 L2 atomics, which was the kernel's duration (0.5 ms at 10 000 genomes).
    constexpr uint32_t PER_LANE = J_TAB / 64;
    __shared__ uint32_t blk_n;
    __shared__ unsigned long long blk_base;
    if (threadIdx.x == 0) blk_n = 0;
    __syncthreads();
    ghip_pair hit[PER_LANE];
    uint32_t nh = 0;
#pragma unroll
    for (uint32_t it = 0; it < PER_LANE; it++) {
        const uint32_t e = lane + 64u * it;
        const unsigned long long key = tk[wave][e];
        const uint32_t common = tc[wave][e];
        bool take = key != J_EMPTY && common >= cmin_floor;
        const uint32_t gi = (uint32_t)(key >> 32), gj = (uint32_t)key;
        // Multi-GPU: every rank runs the whole (cheap) join, so all ranks take the same accept/decline decisions, and
        // reports the pairs with (i + j) mod world == rank.
        if (take && world > 1 && (gi + gj) % world != rank) take = false;
        if (take && gj < row_lo) take = false;
        if (take && big && big[gi] && big[gj]) take = false;   // two genomes of an oversized bucket: their count here is short; the dense pass has them
        uint32_t total = 0;
        if (take) {
            const uint32_t na = lens[gi], nb = lens[gj];  // both > 0: they share a hash
            const uint64_t *ra = hashes + (uint64_t)gi * s, *rb = hashes + (uint64_t)gj * s;
            const uint64_t maxa = ra[na - 1], maxb = rb[nb - 1];
            // i = #{a <= m}, j = #{b <= m}, m = min(max A, max B)  (closed form of the reference's merge loop)
            uint32_t icnt = na, jcnt = nb;
            if (maxa > maxb) {  // upper bound of maxb in row A
                uint32_t l = 0, h = na;
                while (l < h) { const uint32_t mid = (l + h) >> 1; if (ra[mid] <= maxb) l = mid + 1; else h = mid; }
                icnt = l;
            } else if (maxb > maxa) {
                uint32_t l = 0, h = nb;
                while (l < h) { const uint32_t mid = (l + h) >> 1; if (rb[mid] <= maxa) l = mid + 1; else h = mid; }
                jcnt = l;
            }
            total = icnt + jcnt - common;
            take = common >= (uint32_t)cmin[total];
        }
        hit[it].i = gi; hit[it].j = gj; hit[it].common = common; hit[it].total = total; hit[it].ani = take ? 1.0f : 0.0f;   // ani: the mark, until written
        nh += take ? 1u : 0u;
    }
    // lane's offset inside the wave's hits, the wave's inside the block's
    uint32_t incl = nh;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += v; }
    const uint32_t wave_total = __shfl(incl, 63, 64);
    uint32_t wave_base = 0;
    if (lane == 0 && wave_total) wave_base = atomicAdd(&blk_n, wave_total);
    wave_base = __shfl(wave_base, 0, 64);
    __syncthreads();
    if (threadIdx.x == 0 && blk_n) blk_base = atomicAdd(out_count, (unsigned long long)blk_n);
    __syncthreads();
    if (nh) {
        unsigned long long idx = blk_base + wave_base + (incl - nh);
#pragma unroll
        for (uint32_t it = 0; it < PER_LANE; it++)
            if (hit[it].ani != 0.0f) {
                if (idx < cap) { ghip_pair r = hit[it]; r.ani = 0.0f; out[idx] = r; }
                idx++;
            }
    }
}

template <typename T>
T *jalloc(ghip_ctx *ctx, std::vector<void *> &owned, size_t count) {
    T *p = (T *)ghip_pool_alloc(ctx, std::max<size_t>(count, 1) * sizeof(T));
    if (p) owned.push_back(p);
    return p;
}

// start[nb + 1] = exclusive scan of hist[nb] (nb a multiple of J_SCAN_TILE, nb <= 256 tiles); *max_out = largest counter
int scan_counters(ghip_ctx *ctx, std::vector<void *> &owned, const uint32_t *d_hist, uint32_t nb, uint32_t *d_start, uint32_t *d_max) {
    const uint32_t n_tiles = nb / J_SCAN_TILE;
    if (nb % J_SCAN_TILE || n_tiles == 0 || n_tiles > 256) return ghip_set_error(ctx, GHIP_EINVAL, "join: counter array not scannable");
    uint32_t *d_tile = jalloc<uint32_t>(ctx, owned, 512);   // tile sums | tile maxima
    if (!d_tile) return GHIP_EHIP;
    hipLaunchKernelGGL(join_scan_sums_kernel, dim3(n_tiles), dim3(J_THREADS), 0, ctx->stream, d_hist, d_tile, d_tile + 256);
    hipLaunchKernelGGL(join_scan_write_kernel, dim3(n_tiles), dim3(J_THREADS), 0, ctx->stream, d_hist, d_tile, d_tile + 256, n_tiles, d_start, d_max);
    return GHIP_OK;
}

// The fused form of a join call (ghip_options.join_fused): what it needs zeroed, in ONE block cleared by one memset per
// stage -- the first-level cursors, the second-level histogram and cursors, the arrival words of the single-launch scans.
constexpr uint32_t J_ARRIVE = 8;
struct JoinFused {
    bool on = false;
    uint32_t *d_arrive = nullptr;   // [J_ARRIVE] one word per fused scan of the call (zeroed with the call's flag words)
    uint32_t n_arrive = 0;
    uint32_t *d_flags = nullptr;    // bit 3: a first-level bucket outgrew its capacity
};

// the same in one launch (join_scan_fused_kernel); d_hist_b / d_totals nullable (the record totals of the counting pass)
int scan_counters_fused(ghip_ctx *ctx, std::vector<void *> &owned, JoinFused &jf, const uint32_t *d_hist, const uint32_t *d_hist_b, uint32_t nb,
                        uint32_t *d_start, uint32_t *d_max, unsigned long long *d_totals) {
    const uint32_t n_tiles = nb / J_SCAN_TILE;
    if (nb % J_SCAN_TILE || n_tiles == 0 || n_tiles > 256 || jf.n_arrive >= J_ARRIVE) return ghip_set_error(ctx, GHIP_EINVAL, "join: counter array not scannable");
    uint32_t *d_tile = jalloc<uint32_t>(ctx, owned, 512 + 2 * 256);   // tile sums | tile maxima | 64-bit sums of the second array
    if (!d_tile) return GHIP_EHIP;
    hipLaunchKernelGGL(join_scan_fused_kernel, dim3(n_tiles), dim3(J_THREADS), 0, ctx->stream, d_hist, d_hist_b, d_tile, d_tile + 256,
                       reinterpret_cast<unsigned long long *>(d_tile + 512), jf.d_arrive + jf.n_arrive, n_tiles, d_start, d_max, d_totals, jf.d_flags);
    jf.n_arrive++;
    return GHIP_OK;
}

// capacity of a first-level bucket in the fused form: the mean load plus slack for what the digit does to it.  Elements
// (uniform hash bits, but EQUAL hashes travel together: the ~6 members of a species that keep an ancestral k-mer put six
// elements into one bucket -- the load's variance is ~4x a Poisson's on the bench's genomes): twelve Poisson standard
// deviations, i.e. six of the real ones.  Records: all records of a genome pair share one digit -- a bucket's
// load is a sum of ~pairs/256 lumps of up to s records each -- half the mean on top, and room for a few whole lumps.
uint32_t fused_cap1(uint64_t n_valid, bool lumpy, uint32_t lump) {
    const double mean = (double)n_valid / 256.0;
    // (records at 10 000 genomes: ~176 pairs of ~350 +- 200 records per bucket -> sigma ~ 5 400 on a mean of 61 600: this is 7 sigma)
    const double cap = lumpy ? 1.5 * mean + 8.0 * (double)lump + 2048.0 : mean + 12.0 * std::sqrt(mean) + 1024.0;
    return (uint32_t)std::min<double>(((uint64_t)cap + 3) / 4 * 4, (double)(0xffffffffu / 256u));
}

// Partition `src` (total_t candidate slots, at most n_valid_bound of them valid) into nb = 256 << bits2 buckets.
// On return keys_out/vals_out hold the elements bucket by bucket and d_start2[nb + 1] the bucket offsets.
// Exact form: histogram, scan, scatter at both levels (7 launches, one host round trip for the second level's grid).
// Fused form (jf.on): the first level scatters into fixed-capacity buckets -- no histogram, no scan, no round trip --
// and the second level's scan is one launch: 4 launches.  Overflow of a capacity raises jf.d_flags bit 3.
template <typename Src>
int partition(ghip_ctx *ctx, std::vector<void *> &owned, const Src &src, uint32_t total_t, uint32_t n_valid_bound, uint32_t bits2,
              bool with_vals, uint64_t **keys_out, uint32_t **vals_out, uint32_t **d_start2_out, uint32_t *d_max /* nullable */,
              JoinFused *jf = nullptr, bool lumpy = false, uint32_t lump = 0) {
    const uint32_t nb = 256u << bits2;
    const bool dbg = ghip_dbg(ctx->opt, GHIP_DEBUG_COMM);
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "    [partition] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t; } };
    const bool fused = jf && jf->on;
    const uint32_t cap1 = fused ? fused_cap1(n_valid_bound, lumpy, lump) : 0u;
    const size_t n_mid = fused ? (size_t)256 * cap1 : (size_t)n_valid_bound;   // slots of the intermediate (first-level) arrays
    if (fused && n_mid >= (1ull << 32)) return ghip_set_error(ctx, GHIP_EINVAL, "join: fused partition beyond 2^32 slots");
    // hist1[256] | start1[257 -> 260] | cursor1[256] | hist2[nb] | cursor2[nb]: one block, one memset.  (start1 is padded to 260
    // words so that hist2 starts on a 16-byte boundary: the scan kernels read it as uint4)
    constexpr size_t HEAD = 256 + 260 + 256;
    static_assert(HEAD % 4 == 0, "hist2 must be 16-byte aligned");
    uint32_t *d_hist1 = jalloc<uint32_t>(ctx, owned, HEAD + 2 * (size_t)nb);
    uint32_t *d_start2 = jalloc<uint32_t>(ctx, owned, nb + 1);
    uint64_t *k1 = jalloc<uint64_t>(ctx, owned, n_mid), *k2 = jalloc<uint64_t>(ctx, owned, n_valid_bound);
    uint32_t *v1 = with_vals ? jalloc<uint32_t>(ctx, owned, n_mid) : nullptr;
    uint32_t *v2 = with_vals ? jalloc<uint32_t>(ctx, owned, n_valid_bound) : nullptr;
    if (!d_hist1 || !d_start2 || !k1 || !k2 || (with_vals && (!v1 || !v2))) return GHIP_EHIP;
    lap("alloc");
    uint32_t *d_start1 = d_hist1 + 256, *d_cursor1 = d_hist1 + 256 + 260, *d_hist2 = d_hist1 + HEAD, *d_cursor2 = d_hist2 + nb;
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_hist1, 0, (HEAD + 2 * (size_t)nb) * sizeof(uint32_t), ctx->stream));
    const unsigned tiles = (total_t + J_TILE - 1) / J_TILE;
    unsigned tiles2 = 0;
    if (fused) {
        if (tiles) hipLaunchKernelGGL((join_scatter1_kernel<Src>), dim3(tiles), dim3(J_THREADS), 0, ctx->stream, src, (const uint32_t *)nullptr, d_cursor1,
                                      cap1, jf->d_flags, k1, v1);
        tiles2 = (cap1 + J_TILE - 1) / J_TILE;   // (blocks past a bucket's count leave at once)
        if (tiles && tiles2) hipLaunchKernelGGL((join_hist2_kernel<Src>), dim3(tiles2, 256), dim3(J_THREADS), 0, ctx->stream, k1, (const uint32_t *)nullptr,
                                                (const uint32_t *)d_cursor1, cap1, bits2, d_hist2);
        { const int src_ = scan_counters_fused(ctx, owned, *jf, d_hist2, nullptr, nb, d_start2, d_max, nullptr); if (src_) return src_; }
        if (tiles && tiles2) hipLaunchKernelGGL((join_scatter2_kernel<Src>), dim3(tiles2, 256), dim3(J_THREADS), 0, ctx->stream, k1, v1, (const uint32_t *)nullptr,
                                                (const uint32_t *)d_cursor1, cap1, bits2, d_start2, d_cursor2, k2, v2);
        lap("fused: 4 launches");
    } else {
        if (tiles) hipLaunchKernelGGL((join_hist1_kernel<Src>), dim3(tiles), dim3(J_THREADS), 0, ctx->stream, src, d_hist1);
        hipLaunchKernelGGL(join_scan256_kernel, dim3(1), dim3(256), 0, ctx->stream, d_hist1, d_start1);
        if (tiles) hipLaunchKernelGGL((join_scatter1_kernel<Src>), dim3(tiles), dim3(J_THREADS), 0, ctx->stream, src, (const uint32_t *)d_start1, d_cursor1, 0u,
                                      (uint32_t *)nullptr, k1, v1);
        // the second pass runs per first-level bucket: its grid needs the largest bucket
        uint32_t hist1[256];
        GHIP_HIP_CHECK(ctx, hipMemcpyAsync(hist1, d_hist1, sizeof(hist1), hipMemcpyDeviceToHost, ctx->stream));
        GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        lap("pass 1 + sync");
        const uint32_t big = *std::max_element(hist1, hist1 + 256);
        tiles2 = (big + J_TILE - 1) / J_TILE;
        if (tiles2) hipLaunchKernelGGL((join_hist2_kernel<Src>), dim3(tiles2, 256), dim3(J_THREADS), 0, ctx->stream, k1, (const uint32_t *)d_start1,
                                       (const uint32_t *)nullptr, 0u, bits2, d_hist2);
        { const int src_ = scan_counters(ctx, owned, d_hist2, nb, d_start2, d_max); if (src_) return src_; }
        if (tiles2) hipLaunchKernelGGL((join_scatter2_kernel<Src>), dim3(tiles2, 256), dim3(J_THREADS), 0, ctx->stream, k1, v1, (const uint32_t *)d_start1,
                                       (const uint32_t *)nullptr, 0u, bits2, d_start2, d_cursor2, k2, v2);
        lap("pass 2 launches");
    }
    *keys_out = k2;
    if (vals_out) *vals_out = v2;
    *d_start2_out = d_start2;
    return GHIP_OK;
}

// buckets sized so that the average bucket holds <= 256 items
uint32_t bits2_for(uint64_t items, uint32_t min_bits = J_BITS2_MIN) {
    uint32_t b = min_bits;
    while (b < J_BITS2_MAX && (items >> (8 + b)) > 256) b++;
    return b;
}

}  // namespace

// Smallest common that can pass the integer filter for any total >= 1 (0xffff entries = impossible totals).
uint32_t ghip_cmin_floor(const std::vector<uint16_t> &cmin) {
    uint32_t f = 0xffffu;
    for (size_t t = 1; t < cmin.size(); t++) f = std::min<uint32_t>(f, cmin[t]);
    return f;
}

// totals[0] = records of all ranks, totals[1] = records this rank keeps: 64-bit sums of the per-bucket counts (the 32-bit
// offsets of the scan may wrap; the caller declines then)
__global__ __launch_bounds__(1024) void join_totals_kernel(const uint32_t *__restrict__ all_count, const uint32_t *__restrict__ rec_count,
                                                           uint32_t nb, unsigned long long *__restrict__ totals) {
    __shared__ unsigned long long part[2][16];
    unsigned long long a = 0, b = 0;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) { a += all_count[i]; b += rec_count[i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if ((threadIdx.x & 63u) == 0) { part[0][threadIdx.x >> 6] = a; part[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < blockDim.x / 64; w++) t += part[threadIdx.x][w];
        totals[threadIdx.x] = t;
    }
}

// The counting pass of the element stage (sort every bucket, count its records), optimistically with LDS for J_ELEM_SMALL
// elements per wave -- eight blocks per CU instead of three; a bucket beyond that raises flag bit 0 and the pass is repeated
// once with room for J_ELEM_CAP (the limit of the join form: beyond it the caller declines, as before).  *cap_used = the
// capacity the emitting pass has to be launched with.  flags / totals = what the host needs next (read back here).
constexpr uint32_t J_ELEM_SMALL = 384;
static size_t elem_pairs_lds(uint32_t cap) { return (size_t)J_WAVES * cap * (sizeof(uint64_t) + sizeof(uint32_t)); }
constexpr size_t J_BIG_LDS = (size_t)J_BIG_CAP * (sizeof(uint64_t) + sizeof(uint32_t));
struct JoinBig {   // a caller's means to finish the pairs of very large families densely (ghip_pairs_join's d_big)
    uint8_t *d_big = nullptr;      // [n] marks
    uint32_t *d_list = nullptr;    // buckets beyond a wave's stage, for join_elem_pairs_big_kernel
    uint32_t *d_n = nullptr;
    bool used = false;             // the counting pass that succeeded ran with the marks on: the emitting pass must too
};
static int elem_pairs_count(ghip_ctx *ctx, std::vector<void *> &owned, uint32_t enb, uint64_t *ek, uint32_t *ev, const uint32_t *d_estart,
                            uint32_t *d_rcount, uint32_t *d_rstart, uint32_t *d_flags, unsigned long long *d_total, uint32_t rank,
                            uint32_t world, uint32_t row_lo, uint32_t flags[2], unsigned long long totals[2], uint32_t *cap_used,
                            JoinBig *jb = nullptr, JoinFused *jf = nullptr) {
    ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(join_elem_pairs_kernel<false>), elem_pairs_lds(J_ELEM_CAP));
    ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(join_elem_pairs_kernel<true>), elem_pairs_lds(J_ELEM_CAP));
    uint32_t *d_all = jalloc<uint32_t>(ctx, owned, enb);
    if (!d_all) return GHIP_EHIP;
    if (jb && jb->d_big) {
        jb->d_list = jalloc<uint32_t>(ctx, owned, J_BIG_LIST);
        jb->d_n = jalloc<uint32_t>(ctx, owned, 1);
        if (!jb->d_list || !jb->d_n) return GHIP_EHIP;
        ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(join_elem_pairs_big_kernel<false>), J_BIG_LDS);
        ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(join_elem_pairs_big_kernel<true>), J_BIG_LDS);
    }
    uint32_t sticky = 0;   // flag bits that must survive the reset between the two attempts: bit 3, a fused partition's overflow
    for (uint32_t cap : {J_ELEM_SMALL, J_ELEM_CAP}) {
        *cap_used = cap;
        // only the last attempt may hand buckets to the block kernel and long runs to the caller's dense pass
        const bool with_big = jb && jb->d_big && cap == J_ELEM_CAP;
        if (jb) jb->used = with_big;
        if (with_big) GHIP_HIP_CHECK(ctx, hipMemsetAsync(jb->d_n, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL((join_elem_pairs_kernel<false>), dim3(enb / J_WAVES), dim3(J_THREADS), elem_pairs_lds(cap), ctx->stream, ek, ev,
                           d_estart, d_rcount, (const uint32_t *)nullptr, (uint64_t *)nullptr, d_flags, d_all, rank, world, row_lo, cap,
                           with_big ? jb->d_big : (uint8_t *)nullptr, with_big ? jb->d_list : (uint32_t *)nullptr, with_big ? jb->d_n : (uint32_t *)nullptr);
        if (with_big)
            hipLaunchKernelGGL((join_elem_pairs_big_kernel<false>), dim3(256), dim3(J_THREADS), J_BIG_LDS, ctx->stream, ek, ev, d_estart, d_rcount,
                               (const uint32_t *)nullptr, (uint64_t *)nullptr, d_flags, d_all, rank, world, row_lo, jb->d_big, jb->d_list, jb->d_n);
        int rc;
        if (jf && jf->on) rc = scan_counters_fused(ctx, owned, *jf, d_rcount, d_all, enb, d_rstart, nullptr, d_total);   // offsets and both totals in one launch
        else {
            hipLaunchKernelGGL(join_totals_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_all, d_rcount, enb, d_total);
            rc = scan_counters(ctx, owned, d_rcount, enb, d_rstart, nullptr);
        }
        if (rc) return rc;
        totals[0] = totals[1] = 0;
        GHIP_HIP_CHECK(ctx, hipMemcpyAsync(flags, d_flags, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        GHIP_HIP_CHECK(ctx, hipMemcpyAsync(totals, d_total, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        sticky |= flags[0] & 8u;
        flags[0] |= sticky;
        if (sticky || !(flags[0] & 1u) || cap == J_ELEM_CAP) break;   // (an overflow ends the call: the caller repeats it in the exact form)
        // a bucket did not fit the small stage (flags[1] = the largest one): once more, with the full stage
        if (flags[1] > J_ELEM_CAP && !(jb && jb->d_big)) break;   // no stage holds it: the caller declines
        GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_flags, 0, sizeof(uint32_t), ctx->stream));
    }
    return GHIP_OK;
}

// rows idx[0..m) of a packed sketch matrix -> a compact matrix of m rows (the dense pass over the genomes of oversized buckets)
__global__ __launch_bounds__(256) void join_gather_rows_kernel(const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens,
                                                               const uint32_t *__restrict__ idx, uint32_t s, uint64_t *__restrict__ out,
                                                               uint32_t *__restrict__ out_lens) {
    const uint32_t g = idx[blockIdx.x];
    for (uint32_t e = threadIdx.x; e < s; e += blockDim.x) out[(uint64_t)blockIdx.x * s + e] = hashes[(uint64_t)g * s + e];
    if (threadIdx.x == 0) out_lens[blockIdx.x] = lens[g];
}
void ghip_launch_gather_rows(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, const uint32_t *d_idx, size_t m, uint32_t s,
                             uint64_t *d_out, uint32_t *d_out_lens) {
    if (m) hipLaunchKernelGGL(join_gather_rows_kernel, dim3((unsigned)m), dim3(256), 0, ctx->stream, d_hashes, d_lens, d_idx, s, d_out, d_out_lens);
}

// *used = false: the join form declined (see the file header) and nothing was written; run a dense kernel instead.
// fused: the partitions in their fused form (ghip_options.join_fused); *overflow = a first-level capacity was exceeded -- nothing
// usable was written, the caller repeats the call in the exact form.
static int pairs_join_impl(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s,
                           const uint16_t *d_cmin, uint32_t cmin_floor, uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *d_out,
                           unsigned long long *d_count, uint64_t cap, uint64_t *pairs_compared, bool *used, bool *late_decline,
                           std::vector<uint32_t> *empties, uint8_t *d_big, bool *has_big, bool fused, bool *overflow) {
    *used = false;
    *overflow = false;
    if (has_big) *has_big = false;
    if (late_decline) *late_decline = false;
    if (cmin_floor == 0 || cmin_floor == 0xffffu || n < 2 || (uint64_t)n * s >= (1ull << 32)) return GHIP_OK;
    std::vector<uint32_t> lens(n);
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(lens.data(), d_lens, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t n_elem = 0;
    size_t n_empty = 0;
    for (uint32_t l : lens) { n_elem += l; n_empty += (l == 0); }
    // An empty sketch ends the reference's merge loop at once: common = total = 0, Jaccard 0/0 = NaN, and Rust's
    // NaN-dropping f64::max/min turn that into ANI 1.0 -- it pairs with EVERY other sketch without sharing a hash.
    // The join cannot find those pairs; a caller that takes the list of empty sketches (`empties`) adds them itself
    // (N - 1 pairs per empty sketch, no kernel needed), for any other caller the join declines and a dense form runs.
    if (n_empty >= 1) {
        if (!empties) return GHIP_OK;
        for (size_t g = 0; g < n; g++) if (lens[g] == 0) empties->push_back((uint32_t)g);
        if (n_elem == 0) { *used = true; if (pairs_compared) *pairs_compared = (uint64_t)n * (n - 1) / 2; return GHIP_OK; }   // nothing but empty sketches
    }
    struct Owned { ghip_ctx *c; std::vector<void *> p; ~Owned() { for (void *x : p) ghip_pool_free(c, x); } } own{ctx, {}};
    uint32_t *d_flags = jalloc<uint32_t>(ctx, own.p, 6 + J_ARRIVE);  // [0] flags, [1] largest element bucket, [2..3] u64 records of all ranks, [4..5] of this rank, then the fused scans' arrival words
    if (!d_flags) return GHIP_EHIP;
    unsigned long long *d_total = reinterpret_cast<unsigned long long *>(d_flags + 2);
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_flags, 0, (6 + J_ARRIVE) * sizeof(uint32_t), ctx->stream));
    JoinFused jf;
    jf.on = fused; jf.d_arrive = d_flags + 6; jf.d_flags = d_flags;

    ghip_prof_begin(ctx, "pair_join");
    // 1. elements -> 65536 hash buckets
    uint64_t *ek = nullptr; uint32_t *ev = nullptr, *d_estart = nullptr;
    ElemSrc es{d_hashes, d_lens, s, (uint32_t)(n * s), 0u, 1u};
    const uint32_t ebits = bits2_for(n_elem, J_BITS2_MIN_ELEM), enb = 256u << ebits;
    int rc = partition(ctx, own.p, es, es.total, (uint32_t)n_elem, ebits, true, &ek, &ev, &d_estart, d_flags + 1, &jf);
    if (rc) { ghip_prof_end(ctx); return rc; }
    // 2. records per bucket, their offsets, their number
    uint32_t *d_rcount = jalloc<uint32_t>(ctx, own.p, enb), *d_rstart = jalloc<uint32_t>(ctx, own.p, enb + 1);
    if (!d_rcount || !d_rstart) { ghip_prof_end(ctx); return GHIP_EHIP; }
    uint32_t flags[2], ecap = 0;
    unsigned long long totals[2] = {0, 0};   // records of all ranks (what every rank decides on), records of this rank
    JoinBig jb;
    jb.d_big = d_big;
    if ((rc = elem_pairs_count(ctx, own.p, enb, ek, ev, d_estart, d_rcount, d_rstart, d_flags, d_total, rank, world, row_lo, flags, totals, &ecap, &jb, &jf))) { ghip_prof_end(ctx); return rc; }
    if (flags[0] & 8u) { *overflow = true; ghip_prof_end(ctx); return GHIP_OK; }   // (fused form) an element bucket of the first level outgrew its capacity
    if (has_big) *has_big = jb.used && (flags[0] & 4u) != 0;
    uint8_t *const big_used = jb.used ? d_big : nullptr;
    const unsigned long long total_rec = totals[0];
    const uint64_t P = (uint64_t)n * (n - 1) / 2;
    // a dense pass costs ~1 ns per pair, a record ~0.3 ns: beyond 4 records per pair the dense kernel is the better tool
    if ((flags[0] & 1u) || total_rec > 4 * P + (1u << 20) || total_rec >= (1ull << 31)) { ghip_prof_end(ctx); return GHIP_OK; }
    const uint32_t n_rec = (uint32_t)totals[1];
    uint64_t *d_rec = jalloc<uint64_t>(ctx, own.p, n_rec);
    if (!d_rec) { ghip_prof_end(ctx); return GHIP_EHIP; }
    hipLaunchKernelGGL((join_elem_pairs_kernel<true>), dim3(enb / J_WAVES), dim3(J_THREADS), elem_pairs_lds(ecap), ctx->stream, ek, ev, d_estart,
                       (uint32_t *)nullptr, d_rstart, d_rec, d_flags, (uint32_t *)nullptr, rank, world, row_lo, ecap, big_used, jb.d_list, jb.d_n);
    if (big_used)
        hipLaunchKernelGGL((join_elem_pairs_big_kernel<true>), dim3(256), dim3(J_THREADS), J_BIG_LDS, ctx->stream, ek, ev, d_estart, (uint32_t *)nullptr,
                           d_rstart, d_rec, d_flags, (uint32_t *)nullptr, rank, world, row_lo, big_used, jb.d_list, jb.d_n);
    // 3. records -> 65536 pair buckets -> common per pair -> candidates
    uint64_t *rk = nullptr; uint32_t *d_pstart = nullptr;
    RecSrc rs{d_rec, n_rec};
    const uint32_t rbits = bits2_for(n_rec), rnb = 256u << rbits;  // ~350 records per sharing pair: few distinct pairs per bucket
    rc = partition(ctx, own.p, rs, n_rec, n_rec, rbits, false, &rk, nullptr, &d_pstart, nullptr, &jf, true, s);
    if (rc) { ghip_prof_end(ctx); return rc; }
    hipLaunchKernelGGL((join_reduce_kernel<false, 0>), dim3(rnb / J_WAVES), dim3(J_THREADS), 0, ctx->stream, rk, (const uint32_t *)nullptr,
                       (JoinEntry *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, d_pstart, d_hashes, d_lens, s,
                       d_cmin, cmin_floor, rank, world, row_lo, d_out, d_count, cap, d_flags, big_used);
    ghip_prof_end(ctx);
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(flags, d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (flags[0] & 8u) {  // (fused form) a record bucket of the first level outgrew its capacity: records were dropped
        GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
        *overflow = true;
        return GHIP_OK;
    }
    if (flags[0] & 2u) {  // a record bucket overflowed its table: discard what was written, let a dense kernel run
        GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
        // (with world > 1 this decision is this rank's alone -- its own records filled the table -- so the caller must
        // produce exactly this rank's (i + j) mod world share by other means: *late_decline)
        if (late_decline) *late_decline = true;
        return GHIP_OK;
    }
    if (pairs_compared) {  // pairs this rank is responsible for: (i + j) mod world == rank
        if (world == 1) *pairs_compared = P;
        else {
            uint64_t c = 0;
            for (uint64_t i = 0; i + 1 < n; i++) {  // j in (i, n) with (i + j) % world == rank
                const uint64_t first = i + 1 + ((rank + 2 * (uint64_t)world - (2 * i + 1) % world) % world);
                if (first < n) c += (n - 1 - first) / world + 1;
            }
            *pairs_compared = c;
        }
    }
    *used = true;
    return GHIP_OK;
}

int ghip_pairs_join(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s,
                    const uint16_t *d_cmin, uint32_t cmin_floor, uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *d_out,
                    unsigned long long *d_count, uint64_t cap, uint64_t *pairs_compared, bool *used, bool *late_decline,
                    std::vector<uint32_t> *empties, uint8_t *d_big, bool *has_big) {
    bool overflow = false;
    if (ctx->opt.join_fused) {
        std::vector<uint32_t> emp;   // (a repeated call must not list the empty sketches twice)
        const int rc = pairs_join_impl(ctx, d_hashes, d_lens, n, s, d_cmin, cmin_floor, rank, world, row_lo, d_out, d_count, cap, pairs_compared, used,
                                       late_decline, empties ? &emp : nullptr, d_big, has_big, true, &overflow);
        if (rc || !overflow) { if (empties) empties->insert(empties->end(), emp.begin(), emp.end()); return rc; }
        if (d_big) GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_big, 0, n, ctx->stream));   // the marks of the abandoned attempt
    }
    return pairs_join_impl(ctx, d_hashes, d_lens, n, s, d_cmin, cmin_floor, rank, world, row_lo, d_out, d_count, cap, pairs_compared, used, late_decline,
                           empties, d_big, has_big, false, &overflow);
}

// ---------------------------------------------------------------------------------------------------------------------
// Hash-sharded form for several ranks (comm.cpp: precluster_join_sharded).  The one-rank join's element stage -- one pass
// over all N*s hashes, three quarters of its time at 10 000 genomes -- does not shrink when the RECORDS are dealt over
// the ranks; it does when the HASHES are: rank r partitions only the hashes whose first-level digit d has
// d % world == r, finds the records of ALL genome pairs among them and reduces them to per-pair partial counts
// (stage 1: ghip_pairs_join_partials).  The partial counts of all ranks are gathered (16 bytes per sharing pair and
// rank: ~0.7 MB per rank at 10 000 genomes), and every rank sums those of the pairs it owns, (i + j) % world == rank,
// through the same partition + LDS-table machinery with the counts as weights -- the sum IS common; ranks, integer
// filter and candidate list as in the one-rank form (stage 2: ghip_pairs_join_finish).
// *status bit 0: this rank declines (an element bucket beyond the LDS stage / a table overflow / more records than fit).

static int join_partials_impl(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s, uint32_t hrank,
                              uint32_t hworld, void **d_entries_out, uint32_t *n_entries_out, uint32_t *status, unsigned long long *rec_total,
                              bool fused, bool *overflow) {
    *d_entries_out = nullptr; *n_entries_out = 0; *status = 1; *rec_total = 0;
    *overflow = false;
    if (n < 2 || (uint64_t)n * s >= (1ull << 32)) return GHIP_OK;
    std::vector<uint32_t> lens(n);
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(lens.data(), d_lens, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t n_elem = 0;
    for (uint32_t l : lens) { n_elem += l; if (l == 0) return GHIP_OK; }   // an empty sketch pairs with everything (NaN quirk): dense forms only
    struct Owned { ghip_ctx *c; std::vector<void *> p; ~Owned() { for (void *x : p) ghip_pool_free(c, x); } } own{ctx, {}};
    uint32_t *d_flags = jalloc<uint32_t>(ctx, own.p, 8 + J_ARRIVE);  // [0] flags, [1] largest element bucket, [2..3] u64 records found, [4..5] records kept, [8..] arrival words of the fused scans
    if (!d_flags) return GHIP_EHIP;
    unsigned long long *d_total = reinterpret_cast<unsigned long long *>(d_flags + 2);
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_flags, 0, (8 + J_ARRIVE) * sizeof(uint32_t), ctx->stream));
    JoinFused jf;
    jf.on = fused; jf.d_arrive = d_flags + 8; jf.d_flags = d_flags;
    ghip_prof_begin(ctx, "pair_join");
    auto done = [&](int rc) { ghip_prof_end(ctx); return rc; };
    const bool dbg = ghip_dbg(ctx->opt, GHIP_DEBUG_COMM);
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) { hipStreamSynchronize(ctx->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  [join_partials] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t; } };
    uint64_t *ek = nullptr; uint32_t *ev = nullptr, *d_estart = nullptr;
    ElemSrc es{d_hashes, d_lens, s, (uint32_t)(n * s), hrank, hworld};
    const uint64_t mine_bound = n_elem;   // (an upper bound: the share is ~1/hworld of it, but a skewed input may put more here)
    // (bucket count as in the one-rank form: a rank's hashes fill 1/hworld of the first-level digits, the buckets below
    // each digit hold what they hold there)
    const uint32_t ebits = bits2_for(n_elem, J_BITS2_MIN_ELEM), enb = 256u << ebits;
    // (fused form: a rank's hashes fill 1/hworld of the first-level buckets, each as full as in the one-rank form -- the
    // capacity follows from ALL the elements)
    int rc = partition(ctx, own.p, es, es.total, (uint32_t)mine_bound, ebits, true, &ek, &ev, &d_estart, d_flags + 1, &jf);
    if (rc) return done(rc);
    lap("element partition");
    uint32_t *d_rcount = jalloc<uint32_t>(ctx, own.p, enb), *d_rstart = jalloc<uint32_t>(ctx, own.p, enb + 1);
    if (!d_rcount || !d_rstart) return done(GHIP_EHIP);
    uint32_t flags[2], ecap = 0;
    unsigned long long totals[2] = {0, 0};
    if ((rc = elem_pairs_count(ctx, own.p, enb, ek, ev, d_estart, d_rcount, d_rstart, d_flags, d_total, 0u, 1u, 0u, flags, totals, &ecap, nullptr, &jf))) return done(rc);
    lap("count records");
    if (flags[0] & 8u) { *overflow = true; return done(GHIP_OK); }
    *rec_total = totals[0];
    if ((flags[0] & 1u) || totals[0] >= (1ull << 31)) return done(GHIP_OK);   // declined (status bit 0 stays set)
    const uint32_t n_rec = (uint32_t)totals[1];
    uint64_t *d_rec = jalloc<uint64_t>(ctx, own.p, n_rec);
    if (!d_rec) return done(GHIP_EHIP);
    hipLaunchKernelGGL((join_elem_pairs_kernel<true>), dim3(enb / J_WAVES), dim3(J_THREADS), elem_pairs_lds(ecap), ctx->stream, ek, ev, d_estart,
                       (uint32_t *)nullptr, d_rstart, d_rec, d_flags, (uint32_t *)nullptr, 0u, 1u, 0u, ecap, (uint8_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    lap("emit records");
    uint64_t *rk = nullptr; uint32_t *d_pstart = nullptr;
    RecSrc rs{d_rec, n_rec};
    const uint32_t rbits = bits2_for(n_rec), rnb = 256u << rbits;
    if ((rc = partition(ctx, own.p, rs, n_rec, n_rec, rbits, false, &rk, nullptr, &d_pstart, nullptr, &jf, true, s))) return done(rc);
    // pairs per bucket -> their offsets -> the entries, compact and in bucket order (no atomics)
    uint32_t *d_ecount = jalloc<uint32_t>(ctx, own.p, rnb), *d_estart2 = jalloc<uint32_t>(ctx, own.p, rnb + 1);
    if (!d_ecount || !d_estart2) return done(GHIP_EHIP);
    hipLaunchKernelGGL((join_reduce_kernel<false, 1>), dim3(rnb / J_WAVES), dim3(J_THREADS), 0, ctx->stream, rk, (const uint32_t *)nullptr,
                       (JoinEntry *)nullptr, d_ecount, (const uint32_t *)nullptr, d_pstart, d_hashes, d_lens, s, (const uint16_t *)nullptr, 0u, 0u, 1u, 0u,
                       (ghip_pair *)nullptr, (unsigned long long *)nullptr, (uint64_t)0, d_flags, (const uint8_t *)nullptr);
    if ((rc = fused ? scan_counters_fused(ctx, own.p, jf, d_ecount, nullptr, rnb, d_estart2, nullptr, nullptr) : scan_counters(ctx, own.p, d_ecount, rnb, d_estart2, nullptr))) return done(rc);
    uint32_t f0 = 0, n_ent = 0;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(&f0, d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(&n_ent, d_estart2 + rnb, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    lap("count entries");
    if (f0 & 8u) { *overflow = true; return done(GHIP_OK); }   // (fused form) a first-level record bucket outgrew its capacity
    if (f0 & 2u) return done(GHIP_OK);   // a record table overflowed: declined
    JoinEntry *d_ent = (JoinEntry *)ghip_pool_alloc(ctx, (size_t)std::max<uint32_t>(n_ent, 1) * sizeof(JoinEntry));
    if (!d_ent) return done(GHIP_EHIP);
    hipLaunchKernelGGL((join_reduce_kernel<false, 2>), dim3(rnb / J_WAVES), dim3(J_THREADS), 0, ctx->stream, rk, (const uint32_t *)nullptr,
                       d_ent, (uint32_t *)nullptr, d_estart2, d_pstart, d_hashes, d_lens, s, (const uint16_t *)nullptr, 0u, 0u, 1u, 0u,
                       (ghip_pair *)nullptr, (unsigned long long *)nullptr, (uint64_t)0, d_flags, (const uint8_t *)nullptr);
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // (the scratch blocks go back to the pool on return)
    lap("emit entries");
    *d_entries_out = d_ent; *n_entries_out = n_ent; *status = 0;
    return done(GHIP_OK);
}

int ghip_pairs_join_partials(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s, uint32_t hrank,
                             uint32_t hworld, void **d_entries_out, uint32_t *n_entries_out, uint32_t *status, unsigned long long *rec_total) {
    bool overflow = false;
    if (ctx->opt.join_fused) {
        const int rc = join_partials_impl(ctx, d_hashes, d_lens, n, s, hrank, hworld, d_entries_out, n_entries_out, status, rec_total, true, &overflow);
        if (rc || !overflow) return rc;
    }
    return join_partials_impl(ctx, d_hashes, d_lens, n, s, hrank, hworld, d_entries_out, n_entries_out, status, rec_total, false, &overflow);
}

// d_all: the entries of every rank, blocks padded with key = 2^64 - 1.  *ok = false: this rank's tables overflowed (huge
// families); the caller owes its (i + j) % world share by a dense pass.
int ghip_pairs_join_finish(ghip_ctx *ctx, const void *d_all, uint32_t n_all, uint32_t n_mine_bound, const uint64_t *d_hashes, const uint32_t *d_lens,
                           uint32_t s, const uint16_t *d_cmin, uint32_t cmin_floor, uint32_t rank, uint32_t world, ghip_pair *d_out,
                           unsigned long long *d_count, uint64_t cap, bool *ok) {
    *ok = false;
    const bool dbg = ghip_dbg(ctx->opt, GHIP_DEBUG_COMM);
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  [join_finish] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t; } };
    struct Owned { ghip_ctx *c; std::vector<void *> p; ~Owned() { for (void *x : p) ghip_pool_free(c, x); } } own{ctx, {}};
    uint32_t *d_flags = jalloc<uint32_t>(ctx, own.p, 2);
    if (!d_flags) return GHIP_EHIP;
    GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_flags, 0, 2 * sizeof(uint32_t), ctx->stream));
    ghip_prof_begin(ctx, "pair_join");
    auto done = [&](int rc) { ghip_prof_end(ctx); return rc; };
    uint64_t *rk = nullptr; uint32_t *rw = nullptr, *d_pstart = nullptr;
    EntrySrc src{reinterpret_cast<const JoinEntry *>(d_all), n_all, rank, world};
    const uint32_t bound = std::max<uint32_t>(std::min(n_all, n_mine_bound), 1u);
    const uint32_t rbits = bits2_for((uint64_t)bound * 64), rnb = 256u << rbits;   // ~world entries per pair: tables of 256 pairs want few pairs per bucket
    lap("set-up");
    int rc = partition(ctx, own.p, src, n_all, bound, rbits, true, &rk, &rw, &d_pstart, nullptr);
    if (rc) return done(rc);
    lap("partition");
    hipLaunchKernelGGL((join_reduce_kernel<true, 0>), dim3(rnb / J_WAVES), dim3(J_THREADS), 0, ctx->stream, rk, rw, (JoinEntry *)nullptr,
                       (uint32_t *)nullptr, (const uint32_t *)nullptr, d_pstart, d_hashes, d_lens, s, d_cmin, cmin_floor, 0u, 1u, 0u, d_out, d_count, cap, d_flags, (const uint8_t *)nullptr);
    uint32_t flags = 0;
    GHIP_HIP_CHECK(ctx, hipMemcpyAsync(&flags, d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GHIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    lap("reduce + sync");
    if (flags & 2u) { GHIP_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream)); return done(GHIP_OK); }
    *ok = true;
    return done(GHIP_OK);
}
