// Pair stage, hash-probe form (the default for s <= 1024): replaces the serial i<j loop of
// finch::distances (reference src/finch.rs:74-96) / finch::distance::raw_distance.
//
// raw_distance only needs |A n B| and two ranks, so sortedness is not required
// for the intersection itself.  Every sketch is turned once into a cuckoo set (2 hash positions x
// 2-slot 16-byte buckets, 2*next_pow2(s) slots, load <= 0.5) by pair_table_build.  Then
//
//   pair_intersect_tile (probe form): one workgroup = 8 A-sketches as cuckoo sets in LDS x 64 B-sketches.  One
//   sketch pair per wavefront: the wave holds its B-sketch in registers (lane l owns elements l, l+64, ...; coalesced
//   8-byte loads of the packed matrix row) and probes the A-set -- no sorted merge, no data-dependent loop.  A B-sketch
//   is loaded once per 8 pairs and an A-set once per 64.
//       common = #hits;  m = min(max A, max B)
//       max A <= max B :  i = |A|, j = #{b <= max A}          (counted while probing)
//       max A >  max B :  j = |B|, i = #{a <= max B}          (64 LDS samples + one 16-element
//                                                              row segment of the sorted matrix)
//   The kernel is bound by LDS reads at random addresses (bank conflicts), so the sets in LDS hold 31-bit TAGS of
//   the hashes (bits no bucket index uses) instead of the hashes: a probe is two ds_read_b64 (16 bytes) where the full
//   keys took two ds_read_b128 (32 bytes), the tag and the two bucket addresses of a B element are computed once per B row
//   and reused for its 8 A-sets, and a set is 8 KiB, so two workgroups share a CU.  A tag can match where the key does
//   not (never the reverse): the tag pass counts common' >= common, total' <= total, and since cmin[] is non-decreasing
//   its list is a SUPERSET of the exact one; pair_verify then recounts every listed pair against the full 64-bit cuckoo
//   set in global memory (a few thousand pairs out of N^2/2), so what leaves the stage is exact.  (Recounting inside the
//   probe kernel, by the wave that found the pair, cost the probe loop its registers: 0.41 -> 0.59 ms.)
//
// Integer work only; a pair is emitted iff common >= cmin[total] (host table from the f64
// formula) and the host recomputes the exact f32.  Exactness guards: a sketch that holds the
// value 2^64-1 (the empty-slot marker) or whose cuckoo insertion fails makes the host fall back
// to the merge-path kernel (pairs.hip) for the whole call.
#include "ghip_internal.h"
#include "probe_common.h"

namespace {

constexpr int PROBE_TA = 8;        // A-sketches (cuckoo sets) per workgroup
constexpr int PROBE_CB_MAX = 64;   // B-sketches per workgroup (16 / 32 / 64, chosen so the grid fills the chip)
constexpr int PROBE_WAVES = 8;
constexpr int PROBE_THREADS = PROBE_WAVES * 64;
constexpr uint64_t EMPTY = ~0ull;

using namespace ghip_probe;

// ---------------------------------------------------------------------------------------------
// pair_table_build: one 256-thread block per sketch, cuckoo insertion with 64-bit LDS atomics.
// flags: bit 0 = some sketch contains 2^64-1, bit 1 = an insertion did not converge.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_table_build_kernel(const uint64_t *__restrict__ hashes,
                                                               const uint32_t *__restrict__ lens, uint32_t s,
                                                               uint32_t buckets, uint32_t cbits, uint64_t *__restrict__ tables,
                                                               uint32_t *__restrict__ tags, uint32_t *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *tab = reinterpret_cast<unsigned long long *>(smem_raw);
    const uint32_t g = blockIdx.x;
    const uint32_t mask = buckets - 1, slots = 2 * buckets;
    for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) tab[i] = EMPTY;
    __syncthreads();
    const uint32_t n = lens[g];
    const uint64_t *row = hashes + (uint64_t)g * s;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        unsigned long long x = row[i];
        if (x == EMPTY) { atomicOr(flags, 1u); continue; }
        uint32_t b = bucket1(x, mask);
        bool placed = false;
        uint32_t rnd = (uint32_t)(x >> 40) ^ (uint32_t)x ^ (i * 0x9E3779B9u);
        for (int it = 0; it < 4000 && !placed; it++) {
            if (atomicCAS(&tab[2 * b], (unsigned long long)EMPTY, x) == EMPTY) { placed = true; break; }
            if (atomicCAS(&tab[2 * b + 1], (unsigned long long)EMPTY, x) == EMPTY) { placed = true; break; }
            // random-walk cuckoo: evict a pseudo-randomly chosen resident and carry it to its other
            // bucket (a fixed slot pattern can ping-pong between two full buckets forever)
            rnd = rnd * 1664525u + 1013904223u;
            const unsigned long long y = atomicExch(&tab[2 * b + (rnd >> 31)], x);
            x = y;
            if (x == EMPTY) { placed = true; break; }  // a concurrent eviction freed the slot
            const uint32_t b1 = bucket1(x, mask), b2 = bucket2(x, mask, cbits);
            b = (b == b1) ? b2 : b1;
        }
        if (!placed) atomicOr(flags, 2u);
    }
    __syncthreads();
    uint64_t *dst = tables + (uint64_t)g * slots;
    uint32_t *tdst = tags + (uint64_t)g * slots;
    for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) { dst[i] = tab[i]; tdst[i] = tab[i] == EMPTY ? 0u : tag_of(tab[i]); }
}

// Exact (common, total) of one pair, by the calling wavefront: the B row probes the A-sketch's full 64-bit cuckoo set in
// global memory (L2) -- the arithmetic of raw_distance (src/finch.rs:74-96 via finch::distance), nothing approximate left.
__device__ __forceinline__ uint2 exact_pair(const uint64_t *__restrict__ arow, uint32_t na, const uint64_t *__restrict__ brow,
                                            uint32_t nb, const uint64_t *__restrict__ table, uint32_t mask, uint32_t cbits, uint32_t lane) {
    const ulonglong2 *set = reinterpret_cast<const ulonglong2 *>(table);
    const uint64_t maxa = na ? arow[na - 1] : 0ull, maxb = nb ? brow[nb - 1] : 0ull;
    uint32_t common = 0, b_le = 0, a_le = 0;
    // eight elements per lane and step: their loads, then their sixteen bucket reads, are independent (a step is three
    // memory round trips, not twenty-four)
    for (uint32_t e0 = lane; e0 < nb; e0 += 512) {
        uint64_t x[8];
        ulonglong2 v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = e0 + 64u * u < nb ? brow[e0 + 64u * u] : 0ull;
#pragma unroll
        for (int u = 0; u < 8; u++) { v1[u] = set[bucket1(x[u], mask)]; v2[u] = set[bucket2(x[u], mask, cbits)]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool have = e0 + 64u * u < nb;
            common += (have && (v1[u].x == x[u] || v1[u].y == x[u] || v2[u].x == x[u] || v2[u].y == x[u])) ? 1u : 0u;
            b_le += (have && x[u] <= maxa) ? 1u : 0u;
        }
    }
    if (maxa > maxb)   // (wave-uniform) the other rank is only needed then
        for (uint32_t e = lane; e < na; e += 64) a_le += arow[e] <= maxb ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        common += __shfl_xor(common, off, 64); b_le += __shfl_xor(b_le, off, 64); a_le += __shfl_xor(a_le, off, 64);
    }
    uint32_t icnt = 0, jcnt = 0;
    if (na > 0 && nb > 0) {
        if (maxa <= maxb) { icnt = na; jcnt = b_le; } else { jcnt = nb; icnt = a_le; }
    }
    return make_uint2(common, icnt + jcnt - common);
}

// ---------------------------------------------------------------------------------------------
// probe kernel; NT = 64-element register slices of a B-sketch (s <= 64*NT)
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(PROBE_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void pair_probe_tile_kernel(
    const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens, const uint32_t *__restrict__ tags,
    uint32_t n, uint32_t s, uint32_t buckets, uint32_t cb, const uint64_t *__restrict__ row_start, uint32_t nta,
    uint64_t n_work, uint32_t rank, uint32_t world, uint32_t row_lo, const uint16_t *__restrict__ cmin,
    ghip_pair *__restrict__ out, unsigned long long *__restrict__ out_count, uint64_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // [TA] tag sets at a compile-time stride (the largest set of this NT), the two tags of a bucket side by side
    constexpr uint32_t SET_BYTES = (NT > 4 ? 1024u : 256u) * 8u;
    uint64_t *samp = reinterpret_cast<uint64_t *>(smem_raw + PROBE_TA * SET_BYTES);     // [TA][64]
    uint64_t *a_max = samp + PROBE_TA * 64;                                             // [TA]
    uint32_t *a_len = reinterpret_cast<uint32_t *>(a_max + PROBE_TA);                   // [TA]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t mask = buckets - 1;

    for (uint64_t blk = blockIdx.x;; blk += gridDim.x) {  // grid-stride: a dispatch holds < 2^32 work-items
        const uint64_t w = blk * world + rank;
        if (w >= n_work) return;
        // work item -> (A-tile ti, chunk c): largest ti with row_start[ti] <= w
        uint32_t lo = 0, hi = nta;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (row_start[mid] <= w) lo = mid; else hi = mid;
        }
        const uint32_t ti = lo;
        const uint32_t j0 = ti * PROBE_TA + (uint32_t)(w - row_start[ti]) * cb;
        if ((uint64_t)j0 + cb <= row_lo) continue;   // the (new x all) rectangle of an incremental run: every B-sketch of this item is an old genome
        __syncthreads();  // previous work item's LDS image is no longer read
        // ---- stage the 8 tag sets (linear 16-byte copies), rank samples and lengths ----
        for (uint32_t q = 0; q < PROBE_TA; q++) {
            const uint32_t g = ti * PROBE_TA + q;
            const uint4 *src = reinterpret_cast<const uint4 *>(tags + (uint64_t)g * 2 * buckets);
            uint4 *dst = reinterpret_cast<uint4 *>(smem_raw + q * SET_BYTES);
            for (uint32_t e = threadIdx.x; e < buckets / 2; e += PROBE_THREADS)
                dst[e] = (g < n) ? src[e] : make_uint4(0u, 0u, 0u, 0u);
        }
        if (threadIdx.x < PROBE_TA * 64) {
            const uint32_t q = threadIdx.x >> 6, l = threadIdx.x & 63u;
            const uint32_t g = ti * PROBE_TA + q;
            const uint32_t na = (g < n) ? lens[g] : 0u;
            const uint32_t idx = 16 * l + 15;
            samp[q * 64 + l] = (idx < na) ? hashes[(uint64_t)g * s + idx] : EMPTY;
            if (l == 0) { a_len[q] = na; a_max[q] = na ? hashes[(uint64_t)g * s + na - 1] : 0ull; }
        }
        __syncthreads();

        uint64_t amax[PROBE_TA];   // wave-uniform: scalar registers
#pragma unroll
        for (uint32_t q = 0; q < PROBE_TA; q++) {
            const uint64_t v = a_max[q];
            amax[q] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        }
        for (uint32_t jj = wave; jj < cb; jj += PROBE_WAVES) {
            const uint32_t gj = j0 + jj;
            if (gj >= n || gj <= ti * PROBE_TA || gj < row_lo) continue;  // no A of this tile has a smaller index / an old genome
            const uint32_t nb = lens[gj];
            const uint64_t *brow = hashes + (uint64_t)gj * s;
            // per element of the B row, once for all 8 A-sets: its tag (1: no element -- matches no slot, an empty one
            // included), the byte offsets of its two buckets inside a set, and its contribution to #{b <= max A} of every
            // A of the tile; the hash itself is not kept (registers: the probe loop below wants its reads in flight)
            uint32_t tg[NT], o1[NT], o2[NT];
            uint32_t le[PROBE_TA];
#pragma unroll
            for (uint32_t q = 0; q < PROBE_TA; q++) le[q] = 0;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint32_t e = lane + 64u * t;
                const uint64_t x = (e < nb) ? brow[e] : EMPTY;            // (EMPTY is > every max A: never counted)
                tg[t] = (e < nb) ? tag_of(x) : 1u;
                o1[t] = bucket1(x, mask) * 8u;
                o2[t] = bucket2(x, mask, 0u) * 8u;
#pragma unroll
                for (uint32_t q = 0; q < PROBE_TA; q++) le[q] += (x <= amax[q]) ? 0x10000u : 0u;
            }
            const uint64_t maxb = nb ? brow[nb - 1] : 0ull;
#pragma unroll
            for (uint32_t q = 0; q < PROBE_TA; q++) {   // unrolled: a set's base (q * SET_BYTES) is the immediate offset of its LDS reads
                const uint32_t gi = ti * PROBE_TA + q;
                if (gi >= gj) break;
                const uint32_t na = a_len[q];
                const uint64_t maxa = amax[q];
                const unsigned char *set = smem_raw + q * SET_BYTES;
                uint32_t packed = le[q];  // low 16 bits: hits, high 16 bits: #{b <= max A}
                // four elements at a time: their eight bucket reads are issued back to back (16 VGPRs in flight), then
                // compared -- the scheduler is fenced between the two halves, or it serialises read -> wait -> compare
                // per element and the wave idles on LDS latency (both buckets are always read: no data-dependent branch)
                constexpr int GROUP = 4;
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += GROUP) {
                    uint2 v1[GROUP], v2[GROUP];
#pragma unroll
                    for (int u = 0; u < GROUP; u++) {
                        v1[u] = *reinterpret_cast<const uint2 *>(set + o1[t0 + u]);
                        v2[u] = *reinterpret_cast<const uint2 *>(set + o2[t0 + u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < GROUP; u++) {
                        const uint32_t x = tg[t0 + u];
                        packed += (uint32_t)(v1[u].x == x) | (uint32_t)(v1[u].y == x) | (uint32_t)(v2[u].x == x) | (uint32_t)(v2[u].y == x);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) packed += __shfl_xor(packed, off, 64);
                const uint32_t common = packed & 0xffffu;
                uint32_t icnt = 0, jcnt = 0;
                if (na > 0 && nb > 0) {
                    if (maxa <= maxb) {
                        icnt = na; jcnt = packed >> 16;
                    } else {  // #{a <= max B}: whole 16-blocks from the LDS samples, the partial block from the row
                        jcnt = nb;
                        const uint32_t full = (uint32_t)__popcll(__ballot(samp[q * 64 + lane] <= maxb));
                        const uint32_t e = 16 * full + (lane & 15u);
                        const bool in = lane < 16 && e < na && hashes[(uint64_t)gi * s + e] <= maxb;
                        icnt = 16 * full + (uint32_t)__popcll(__ballot(in));
                    }
                }
                // (tag hits can exceed the exact count: keep the index inside cmin[0 .. 2 s + 1])
                const uint32_t total = min(icnt + jcnt - min(common, icnt + jcnt), 2u * s + 1u);
                if (lane == 0 && common >= (uint32_t)cmin[total]) {
                    unsigned long long idx = atomicAdd(out_count, 1ull);
                    if (idx < cap) {
                        ghip_pair r;
                        r.i = gi; r.j = gj; r.common = common; r.total = total; r.ani = 0.0f;
                        out[idx] = r;   // provisional: pair_verify_kernel recounts it against the full keys
                    }
                }
            }
        }
    }
}

// The ARRANGED form of the same kernel (ghip_options.probe_arranged).  B rows come from `arranged` (pair_arrange_kernel:
// NS x 64 slots per sketch, slot t * 64 + lane, 2^64 - 1 = hole) and the second bucket is the constrained one, so the 32
// lanes of a ds_read_b64 group read 32 different bank pairs.  The loops are turned inside out with respect to the kernel
// above: a GROUP of four row slots is fetched (the next group travels meanwhile), its tags and bucket offsets are worked
// out once, and all 8 A-sets are probed with them -- 12 registers of row state instead of 3 NS: 83 / 65 VGPRs and no spills (the
// row-major order of the kernel above holds 128 with 7 spilled at 16 slots per lane).
template <int NS>
__global__ __launch_bounds__(PROBE_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void pair_probe_arranged_kernel(
    const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens, const uint32_t *__restrict__ tags,
    const uint64_t *__restrict__ arranged, uint32_t cbits, uint32_t n, uint32_t s, uint32_t buckets, uint32_t cb, const uint64_t *__restrict__ row_start,
    uint32_t nta, uint64_t n_work, uint32_t rank, uint32_t world, uint32_t row_lo, const uint16_t *__restrict__ cmin,
    ghip_pair *__restrict__ out, unsigned long long *__restrict__ out_count, uint64_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr uint32_t SET_BYTES = (NS > 4 ? 1024u : 256u) * 8u;
    // (four slots per lane are two groups of two: as ONE group of four the group loop has a single turn, the compiler drops it, hoists
    // every set's LDS reads to the top and spills 176 bytes per lane at 128 VGPRs -- two turns: 65 VGPRs, no scratch; round 6, static)
    constexpr int GROUP = NS > 4 ? 4 : 2;
    static_assert(NS % GROUP == 0, "slots per lane");
    uint64_t *samp = reinterpret_cast<uint64_t *>(smem_raw + PROBE_TA * SET_BYTES);     // [TA][64]
    uint64_t *a_max = samp + PROBE_TA * 64;                                             // [TA]
    uint32_t *a_len = reinterpret_cast<uint32_t *>(a_max + PROBE_TA);                   // [TA]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t mask = buckets - 1;
    const uint32_t hole_off = (lane & 31u & mask) * 8u;   // a hole reads the bucket of its own lane's residue: no conflict with its group

    for (uint64_t blk = blockIdx.x;; blk += gridDim.x) {
        const uint64_t w = blk * world + rank;
        if (w >= n_work) return;
        uint32_t lo = 0, hi = nta;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (row_start[mid] <= w) lo = mid; else hi = mid;
        }
        const uint32_t ti = lo;
        const uint32_t j0 = ti * PROBE_TA + (uint32_t)(w - row_start[ti]) * cb;
        if ((uint64_t)j0 + cb <= row_lo) continue;
        __syncthreads();
        for (uint32_t q = 0; q < PROBE_TA; q++) {
            const uint32_t g = ti * PROBE_TA + q;
            const uint4 *src = reinterpret_cast<const uint4 *>(tags + (uint64_t)g * 2 * buckets);
            uint4 *dst = reinterpret_cast<uint4 *>(smem_raw + q * SET_BYTES);
            for (uint32_t e = threadIdx.x; e < buckets / 2; e += PROBE_THREADS)
                dst[e] = (g < n) ? src[e] : make_uint4(0u, 0u, 0u, 0u);
        }
        if (threadIdx.x < PROBE_TA * 64) {
            const uint32_t q = threadIdx.x >> 6, l = threadIdx.x & 63u;
            const uint32_t g = ti * PROBE_TA + q;
            const uint32_t na = (g < n) ? lens[g] : 0u;
            const uint32_t idx = 16 * l + 15;
            samp[q * 64 + l] = (idx < na) ? hashes[(uint64_t)g * s + idx] : EMPTY;
            if (l == 0) { a_len[q] = na; a_max[q] = na ? hashes[(uint64_t)g * s + na - 1] : 0ull; }
        }
        __syncthreads();

        uint64_t amax[PROBE_TA];   // wave-uniform: scalar registers
#pragma unroll
        for (uint32_t q = 0; q < PROBE_TA; q++) {
            const uint64_t v = a_max[q];
            amax[q] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        }
        for (uint32_t jj = wave; jj < cb; jj += PROBE_WAVES) {
            const uint32_t gj = j0 + jj;
            if (gj >= n || gj <= ti * PROBE_TA || gj < row_lo) continue;
            const uint32_t nb = lens[gj];
            const uint64_t *arow = arranged + (uint64_t)gj * (NS * 64) + lane;
            uint32_t packed[PROBE_TA];   // per A-set: low 16 bits hits, high 16 bits #{b <= max A}
#pragma unroll
            for (uint32_t q = 0; q < PROBE_TA; q++) packed[q] = 0;
            constexpr int NG = (NS + GROUP - 1) / GROUP;
            uint64_t xn[GROUP];
#pragma unroll
            for (int u = 0; u < GROUP; u++) xn[u] = arow[64 * u];
#pragma unroll 1   // (a real loop: unrolled, the compiler hoists every group's loads and offsets to the top and spills 600 registers)
            for (int gidx = 0; gidx < NG; gidx++) {
                uint32_t tg[GROUP], o1[GROUP], o2[GROUP];
                uint64_t x[GROUP];
#pragma unroll
                for (int u = 0; u < GROUP; u++) x[u] = xn[u];
                if (gidx + 1 < NG) {   // the next group's slots are in flight while this one is probed
#pragma unroll
                    for (int u = 0; u < GROUP; u++) xn[u] = ((gidx + 1) * GROUP + u < NS) ? arow[64 * ((gidx + 1) * GROUP + u)] : EMPTY;
                }
#pragma unroll
                for (int u = 0; u < GROUP; u++) {
                    const bool have = x[u] != EMPTY && gidx * GROUP + u < NS;   // (a sketch holding 2^64 - 1 never gets here: probe_flags)
                    tg[u] = have ? tag_of(x[u]) : 1u;
                    o1[u] = have ? bucket1(x[u], mask) * 8u : hole_off;
                    o2[u] = have ? bucket2(x[u], mask, cbits) * 8u : hole_off;
#pragma unroll
                    for (uint32_t q = 0; q < PROBE_TA; q++) packed[q] += (have && x[u] <= amax[q]) ? 0x10000u : 0u;
                }
#pragma unroll
                for (uint32_t q = 0; q < PROBE_TA; q++) {   // unrolled: a set's base is the immediate offset of its LDS reads
                    const unsigned char *set = smem_raw + q * SET_BYTES;
                    uint2 v1[GROUP], v2[GROUP];
#pragma unroll
                    for (int u = 0; u < GROUP; u++) {
                        v1[u] = *reinterpret_cast<const uint2 *>(set + o1[u]);
                        v2[u] = *reinterpret_cast<const uint2 *>(set + o2[u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < GROUP; u++) {
                        const uint32_t t = tg[u];
                        packed[q] += (uint32_t)(v1[u].x == t) | (uint32_t)(v1[u].y == t) | (uint32_t)(v2[u].x == t) | (uint32_t)(v2[u].y == t);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const uint64_t maxb = nb ? hashes[(uint64_t)gj * s + nb - 1] : 0ull;
#pragma unroll
            for (uint32_t q = 0; q < PROBE_TA; q++) {
                const uint32_t gi = ti * PROBE_TA + q;
                uint32_t pk = packed[q];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) pk += __shfl_xor(pk, off, 64);
                if (gi >= gj) continue;   // (wave-uniform; the sets past the diagonal were probed for nothing -- diagonal tiles only)
                const uint32_t na = a_len[q];
                const uint64_t maxa = amax[q];
                const uint32_t common = pk & 0xffffu;
                uint32_t icnt = 0, jcnt = 0;
                if (na > 0 && nb > 0) {
                    if (maxa <= maxb) {
                        icnt = na; jcnt = pk >> 16;
                    } else {
                        jcnt = nb;
                        const uint32_t full = (uint32_t)__popcll(__ballot(samp[q * 64 + lane] <= maxb));
                        const uint32_t e = 16 * full + (lane & 15u);
                        const bool in = lane < 16 && e < na && hashes[(uint64_t)gi * s + e] <= maxb;
                        icnt = 16 * full + (uint32_t)__popcll(__ballot(in));
                    }
                }
                const uint32_t total = min(icnt + jcnt - min(common, icnt + jcnt), 2u * s + 1u);
                if (lane == 0 && common >= (uint32_t)cmin[total]) {
                    unsigned long long idx = atomicAdd(out_count, 1ull);
                    if (idx < cap) {
                        ghip_pair r;
                        r.i = gi; r.j = gj; r.common = common; r.total = total; r.ani = 0.0f;
                        out[idx] = r;   // provisional: pair_verify_kernel recounts it against the full keys
                    }
                }
            }
        }
    }
}

// Exact (common, total) of every listed pair, one wavefront per pair (exact_pair).  The host's f64 recheck
// (ghip_pairs_finalize) then drops what only the tags let through.
__global__ __launch_bounds__(256) void pair_verify_kernel(const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens,
                                                          const uint64_t *__restrict__ tables, uint32_t s, uint32_t buckets, uint32_t cbits,
                                                          ghip_pair *__restrict__ out, const unsigned long long *__restrict__ out_count,
                                                          uint64_t cap) {
    const uint64_t listed = min((uint64_t)*out_count, cap);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < listed; c += (uint64_t)gridDim.x * 4) {
        const uint32_t gi = out[c].i, gj = out[c].j;
        const uint2 ex = exact_pair(hashes + (uint64_t)gi * s, lens[gi], hashes + (uint64_t)gj * s, lens[gj],
                                    tables + (uint64_t)gi * 2 * buckets, buckets - 1, cbits, lane);
        if (lane == 0) { out[c].common = ex.x; out[c].total = ex.y; }
    }
}

// One 256-thread block per sketch: the row's hashes dealt to the NT x 64 slots of the arranged form.  A hash whose first
// bucket has residue r mod 32 goes to lane r of the first or the second 32-lane group (alternating), at the next free
// step t: every (t, group) -- one LDS cycle of a ds_read_b64 when conflict-free -- then reads 32 different bank pairs.
// A class holds s / 32 hashes on average and 2 NT slots (NT = 16 for s <= 1024: 32 slots for 31 +- 5.6); the hashes of
// an overfull class take any hole (they conflict with that step's rightful lane: time, not correctness).  The order of the
// atomics is not deterministic; the counts the probe takes do not depend on it.
template <int NT>
__global__ __launch_bounds__(256) void pair_arrange_kernel(const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens, uint32_t s,
                                                           uint32_t buckets, uint64_t *__restrict__ arranged) {
    __shared__ unsigned long long slot[NT * 64];
    __shared__ uint32_t cnt[32], n_over;
    __shared__ unsigned long long over[NT * 64];
    const uint32_t g = blockIdx.x, mask = buckets - 1;
    for (uint32_t i = threadIdx.x; i < NT * 64; i += 256) slot[i] = EMPTY;
    if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) n_over = 0;
    __syncthreads();
    const uint32_t n = min(lens[g], (uint32_t)(NT * 64));
    const uint64_t *row = hashes + (uint64_t)g * s;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const unsigned long long x = row[i];
        const uint32_t r = bucket1(x, mask) & 31u;
        const uint32_t k = atomicAdd(&cnt[r], 1u);
        if (k < 2u * NT) slot[(k >> 1) * 64 + (k & 1u) * 32 + r] = x;
        else over[atomicAdd(&n_over, 1u)] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // (rare and short: the hashes of overfull classes into whatever is free)
        uint32_t at = 0;
        for (uint32_t o = 0; o < n_over; o++) {
            while (at < NT * 64 && slot[at] != EMPTY) at++;
            if (at < NT * 64) slot[at++] = over[o];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NT * 64; i += 256) arranged[(uint64_t)g * (NT * 64) + i] = slot[i];
}

}  // namespace

// slots per sketch of the arranged form: NS x 64 with NS = 4 (s <= 256) or 16 (s <= 1024) -- no more than the free form's
// row.  A residue class (s / 32 hashes on average) then overflows its 2 NS slots about as often as not and ~60 of 1 000
// hashes end up in a lane of another residue; the CPU simulation of the LDS groups (scripts/sim/probe_bank_cycles.py) says
// that costs less than the extra wave-instructions of a roomier row: 149 LDS cycles per row and A-set at NS = 16, 149 at 18,
// 153 at 20 (3 constrained bits; 218 for the free form's rows)
size_t ghip_probe_arranged_slots(uint32_t s) { return (s <= 256 ? 4 : 16) * 64; }

size_t ghip_probe_table_slots(uint32_t s) {
    uint32_t b = 1;
    while (b < s) b <<= 1;
    return 2 * (size_t)b;
}

void ghip_launch_pair_tables(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s,
                             uint64_t *d_tables, uint32_t *d_tags, uint32_t *d_flags, uint64_t *d_arranged /* nullable: the arranged form */,
                             uint32_t cbits /* bucket bits the second cuckoo choice shares with the first (0: free) */) {
    if (n == 0) return;
    const uint32_t buckets = (uint32_t)(ghip_probe_table_slots(s) / 2);
    ghip_prof_begin(ctx, "pair_table_build");
    hipLaunchKernelGGL(pair_table_build_kernel, dim3((unsigned)n), dim3(256), 2 * (size_t)buckets * sizeof(uint64_t),
                       ctx->stream, d_hashes, d_lens, s, buckets, cbits, d_tables, d_tags, d_flags);
    if (d_arranged) {
        if (s <= 256) hipLaunchKernelGGL(pair_arrange_kernel<4>, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_hashes, d_lens, s, buckets, d_arranged);
        else hipLaunchKernelGGL(pair_arrange_kernel<16>, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_hashes, d_lens, s, buckets, d_arranged);
    }
    ghip_prof_end(ctx);
}

// bucket bits the arranged form constrains (ghip_options.probe_arranged: 0 off, 1 by table size, 2..4 that many bits)
uint32_t ghip_probe_constrained_bits(uint32_t option, uint32_t s) {
    if (option == 0) return 0;
    if (option >= 2) return std::min(option, 4u);
    return ghip_probe_table_slots(s) / 2 >= 1024 ? 3u : 2u;
}

// work items: A-tile ti (8 sketches) x chunk of cb B-sketches starting at 8*ti.  cb shrinks for small
// N so that there are several work items per CU (each stages 128 KiB, amortised over 8*cb pairs).
uint64_t ghip_probe_work_rows(size_t n, int num_cus, uint32_t *cb_out, std::vector<uint64_t> &row_start) {
    const size_t nta = (n + PROBE_TA - 1) / PROBE_TA;
    uint32_t cb = PROBE_CB_MAX;
    while (cb > 16 && nta * ((n / 2 + cb - 1) / cb) < (size_t)num_cus * 12) cb >>= 1;
    row_start.assign(nta + 1, 0);
    for (size_t ti = 0; ti < nta; ti++) {
        const size_t j0 = ti * PROBE_TA;
        row_start[ti + 1] = row_start[ti] + (n - j0 + cb - 1) / cb;
    }
    *cb_out = cb;
    return row_start[nta];
}

uint64_t ghip_probe_pairs_of_rank(size_t n, uint32_t PROBE_CB, const std::vector<uint64_t> &row_start, uint32_t rank, uint32_t world) {
    if (world == 1) return (uint64_t)n * (n - 1) / 2;
    uint64_t cnt = 0;
    const size_t nta = row_start.size() - 1;
    for (size_t ti = 0; ti < nta; ti++)
        for (uint64_t w = row_start[ti]; w < row_start[ti + 1]; w++) {
            if (w % world != rank) continue;
            const uint64_t j0 = ti * PROBE_TA + (w - row_start[ti]) * PROBE_CB;
            for (uint64_t gi = ti * PROBE_TA; gi < std::min<uint64_t>((ti + 1) * PROBE_TA, n); gi++) {
                const uint64_t lo = std::max<uint64_t>(j0, gi + 1), hi = std::min<uint64_t>(j0 + PROBE_CB, n);
                if (hi > lo) cnt += hi - lo;
            }
        }
    return cnt;
}

void ghip_launch_pairs_probe(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, const uint64_t *d_tables,
                             const uint32_t *d_tags, size_t n, uint32_t s, uint32_t cb, const uint64_t *d_row_start, uint32_t nta, uint64_t n_work,
                             const uint16_t *d_cmin, uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *d_out,
                             unsigned long long *d_count, uint64_t cap, const uint64_t *d_arranged /* nullable: tables and rows in the arranged form */,
                             uint32_t cbits) {
    const uint64_t mine = n_work > rank ? (n_work - rank + world - 1) / world : 0;
    if (mine == 0) return;
    const uint32_t buckets = (uint32_t)(ghip_probe_table_slots(s) / 2);
    const size_t lds = (size_t)PROBE_TA * (s <= 256 ? 256 : 1024) * 8 + PROBE_TA * 64 * 8 + PROBE_TA * 8 + PROBE_TA * 4;   // (pair_probe_tile_kernel: SET_BYTES)
    const unsigned grid = (unsigned)std::min<uint64_t>(mine, GHIP_MAX_GRID);
    ghip_prof_begin(ctx, "pair_intersect_tile");
#define GHIP_PROBE_LAUNCH(KERNEL_, ...)                                                                                                   \
    do {                                                                                                                                 \
        ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(KERNEL_), 160 * 1024);                                                   \
        hipLaunchKernelGGL(KERNEL_, dim3(grid), dim3(PROBE_THREADS), lds, ctx->stream, d_hashes, d_lens, d_tags, __VA_ARGS__ (uint32_t)n, \
                           s, buckets, cb, d_row_start, nta, n_work, rank, world, row_lo, d_cmin, d_out, d_count, cap);                   \
    } while (0)
    if (d_arranged) {
        if (s <= 256) GHIP_PROBE_LAUNCH(pair_probe_arranged_kernel<4>, d_arranged, cbits,); else GHIP_PROBE_LAUNCH(pair_probe_arranged_kernel<16>, d_arranged, cbits,);
    } else {
        if (s <= 256) GHIP_PROBE_LAUNCH(pair_probe_tile_kernel<4>, ); else GHIP_PROBE_LAUNCH(pair_probe_tile_kernel<16>, );
    }
#undef GHIP_PROBE_LAUNCH
    // the listed pairs again, against the full keys (part of the same profiled stage: the list is a few thousand pairs)
    hipLaunchKernelGGL(pair_verify_kernel, dim3(2048), dim3(256), 0, ctx->stream, d_hashes, d_lens, d_tables, s, buckets, d_arranged ? cbits : 0u, d_out, d_count, cap);
    ghip_prof_end(ctx);
}
