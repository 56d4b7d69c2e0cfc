// ANI on candidate pairs, batched on gfx950: replaces the per-pair `skani dist` subprocess of
// SkaniClusterer::calculate_ani (reference src/skani.rs:708-788).  Build-defined estimator in
// skani's style (FracMinHash seeds k=15 c=125, 20 kb chunks, containment^(1/k), aligned-
// fraction gate, two-decimal percent) -- skani parity is UNPINNED, see DESIGN.md "ANI".
// The device does integer work only; the host finishes pow/rounding.
//
//   ani_seeds : pass over the base stream; canonical 2-bit k-mer -> invertible 64-bit mix;
//               seeds with hash < 2^64/c are appended (hash, chunk) and counted per chunk.
//   ani_table : per genome, insert the seed hashes into an open-addressing table (set).
//   ani_pairs : one workgroup per (pair, direction): every query seed probes the reference's
//               table; matches are counted per 20 kb query chunk in LDS; a chunk is aligned
//               iff M_c*10000 >= 510*T_c; emits sum M_c, sum T_c, aligned bases.
#include "ghip_internal.h"

namespace {

__device__ __forceinline__ uint64_t mm_hash64(uint64_t key) {
    key = ~key + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

__device__ __forceinline__ uint32_t base_code(uint32_t c) {
    uint32_t d = c - 0x41u;
    bool ok = d < 20u && ((0x80045u >> d) & 1u);
    return ok ? (((c >> 1) ^ (c >> 2)) & 3u) : 4u;
}

constexpr uint32_t SEED_LDS_CAP = 1024;   // seeds buffered per block (expected 16384/c ~ 131)
constexpr uint32_t SEED_LDS_CHUNKS = 64;  // per-block chunk counters

// Seeds are rare (1/c of the k-mers), so they are collected in LDS and flushed with ONE global
// atomic per block: a per-genome counter bumped once per seed serialises at the L2 (4e7
// same-address atomics at N=1000).
__global__ __launch_bounds__(GHIP_SKETCH_THREADS) void ani_seeds_kernel(
    const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ starts,
    const uint64_t *__restrict__ lens, const ghip_sketch_work *__restrict__ work, uint32_t K,
    uint64_t thr, uint32_t chunk, uint64_t *__restrict__ seed_hash, uint32_t *__restrict__ seed_chunk,
    const uint64_t *__restrict__ seed_start, uint32_t *__restrict__ seed_count,
    uint32_t *__restrict__ chunk_total, const uint64_t *__restrict__ chunk_start) {
    __shared__ uint64_t l_hash[SEED_LDS_CAP];
    __shared__ uint32_t l_chunk[SEED_LDS_CAP];
    __shared__ uint32_t l_ctot[SEED_LDS_CHUNKS];
    __shared__ uint32_t l_n, l_base;

    const ghip_sketch_work wk = work[blockIdx.x];
    const uint32_t g = wk.slot;
    const uint64_t L = lens[g];
    const uint64_t blk0 = (uint64_t)wk.chunk * GHIP_SKETCH_CHUNK;
    const uint64_t p0 = blk0 + (uint64_t)threadIdx.x * GHIP_SKETCH_POS_PER_THREAD;
    const uint32_t ch_first = (uint32_t)(blk0 / chunk);
    const uint64_t sstart = seed_start[g];
    const uint32_t scap = (uint32_t)(seed_start[g + 1] - sstart);
    uint32_t *ctot = chunk_total + chunk_start[g];
    if (threadIdx.x < SEED_LDS_CHUNKS) l_ctot[threadIdx.x] = 0;
    if (threadIdx.x == 0) l_n = 0;
    __syncthreads();

    if (p0 < L) {
        const uint4 *src = reinterpret_cast<const uint4 *>(bytes + starts[g] + p0);
        const uint64_t mask = (K < 32) ? ((1ull << (2 * K)) - 1) : ~0ull;
        uint64_t fwd = 0, rev = 0;
        uint32_t good = 0;
        const int NB = GHIP_SKETCH_POS_PER_THREAD + K - 1;
        const int NV = (NB + 15) / 16;
        for (int v = 0; v < NV; v++) {
            uint4 cur = src[v];
            uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int b = v * 16 + j;
                uint32_t code = base_code((words[j >> 2] >> (8 * (j & 3))) & 0xffu);
                const bool valid = code <= 3u;
                code &= 3u;
                fwd = ((fwd << 2) | code) & mask;
                rev = (rev >> 2) | ((uint64_t)(3u - code) << (2 * (K - 1)));
                good = valid ? good + 1 : 0;
                const uint64_t h = mm_hash64(fwd < rev ? fwd : rev);
                if (b < NB && b >= (int)K - 1 && good >= K && h < thr) {
                    const uint64_t pos = p0 + (uint64_t)(b - ((int)K - 1));
                    const uint32_t ch = (uint32_t)(pos / chunk);
                    const uint32_t li = atomicAdd(&l_n, 1u);
                    if (li < SEED_LDS_CAP) { l_hash[li] = h; l_chunk[li] = ch; }
                    else {  // LDS buffer full (never at c=125): straight to the global list
                        uint32_t idx = atomicAdd(&seed_count[g], 1u);
                        if (idx < scap) { seed_hash[sstart + idx] = h; seed_chunk[sstart + idx] = ch; }
                    }
                    const uint32_t rel = ch - ch_first;
                    if (rel < SEED_LDS_CHUNKS) atomicAdd(&l_ctot[rel], 1u);
                    else atomicAdd(&ctot[ch], 1u);
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nloc = min(l_n, SEED_LDS_CAP);
    if (threadIdx.x == 0) l_base = nloc ? atomicAdd(&seed_count[g], nloc) : 0u;
    __syncthreads();
    const uint32_t base = l_base;
    for (uint32_t i = threadIdx.x; i < nloc; i += blockDim.x) {
        const uint32_t idx = base + i;
        if (idx < scap) { seed_hash[sstart + idx] = l_hash[i]; seed_chunk[sstart + idx] = l_chunk[i]; }
    }
    if (threadIdx.x < SEED_LDS_CHUNKS && l_ctot[threadIdx.x]) atomicAdd(&ctot[ch_first + threadIdx.x], l_ctot[threadIdx.x]);
}

// one block per genome; table size is a power of two >= 2*count; slot = low bits of the hash
__global__ __launch_bounds__(256) void ani_table_kernel(const uint64_t *__restrict__ seed_hash,
                                                        const uint64_t *__restrict__ seed_start,
                                                        const uint32_t *__restrict__ seed_count,
                                                        uint64_t *__restrict__ table,
                                                        const uint64_t *__restrict__ table_start) {
    const uint32_t g = blockIdx.x;
    const uint64_t tstart = table_start[g];
    const uint64_t tsize = table_start[g + 1] - tstart;
    if (tsize == 0) return;
    const uint64_t tmask = tsize - 1;
    unsigned long long *tab = reinterpret_cast<unsigned long long *>(table + tstart);
    const uint64_t *src = seed_hash + seed_start[g];
    const uint32_t n = seed_count[g];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t h = src[i];
        uint64_t slot = h & tmask;
        for (;;) {
            unsigned long long prev = atomicCAS(&tab[slot], (unsigned long long)GHIP_EMPTY_SLOT, (unsigned long long)h);
            if (prev == GHIP_EMPTY_SLOT || prev == h) break;
            slot = (slot + 1) & tmask;
        }
    }
}

__global__ __launch_bounds__(256) void ani_pairs_kernel(
    const uint32_t *__restrict__ pairs, const uint64_t *__restrict__ seed_hash,
    const uint32_t *__restrict__ seed_chunk, const uint64_t *__restrict__ seed_start,
    const uint32_t *__restrict__ seed_count, const uint64_t *__restrict__ table,
    const uint64_t *__restrict__ table_start, const uint32_t *__restrict__ chunk_total,
    const uint64_t *__restrict__ chunk_start, const uint64_t *__restrict__ glen, uint32_t chunk,
    uint64_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *mc = reinterpret_cast<uint32_t *>(smem_raw);
    __shared__ unsigned long long red[3];

    const uint32_t pair = blockIdx.x >> 1, dir = blockIdx.x & 1u;
    const uint32_t q = pairs[2 * pair + dir], r = pairs[2 * pair + (dir ^ 1u)];
    const uint32_t nch = (uint32_t)(chunk_start[q + 1] - chunk_start[q]);
    for (uint32_t i = threadIdx.x; i < nch; i += blockDim.x) mc[i] = 0;
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    __syncthreads();

    const uint64_t tstart = table_start[r];
    const uint64_t tsize = table_start[r + 1] - tstart;
    const uint64_t tmask = tsize - 1;
    const uint64_t *tab = table + tstart;
    const uint64_t qs = seed_start[q];
    const uint32_t nq = seed_count[q];
    if (tsize != 0) {
        for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) {
            const uint64_t h = seed_hash[qs + i];
            uint64_t slot = h & tmask;
            bool hit = false;
            for (;;) {
                const uint64_t v = tab[slot];
                if (v == h) { hit = true; break; }
                if (v == GHIP_EMPTY_SLOT) break;
                slot = (slot + 1) & tmask;
            }
            if (hit) atomicAdd(&mc[seed_chunk[qs + i]], 1u);
        }
    }
    __syncthreads();

    unsigned long long M = 0, T = 0, bases = 0;
    const uint64_t Lq = glen[q];
    const uint32_t *tc = chunk_total + chunk_start[q];
    for (uint32_t c = threadIdx.x; c < nch; c += blockDim.x) {
        const unsigned long long t = tc[c], m = mc[c];
        if (t >= 1 && m * 10000ull >= 510ull * t) {
            M += m; T += t;
            const uint64_t lo = (uint64_t)c * chunk;
            uint64_t hi = lo + chunk;
            if (hi > Lq) hi = Lq;
            bases += hi - lo;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        M += __shfl_xor(M, off, 64);
        T += __shfl_xor(T, off, 64);
        bases += __shfl_xor(bases, off, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&red[0], M); atomicAdd(&red[1], T); atomicAdd(&red[2], bases);
    }
    __syncthreads();
    if (threadIdx.x < 3) out[(uint64_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x];
}

}  // namespace

void ghip_launch_ani_seeds(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t c, uint32_t chunk,
                           uint64_t *d_seed_hash, uint32_t *d_seed_chunk, const uint64_t *d_seed_start,
                           uint32_t *d_seed_count, uint32_t *d_chunk_total, const uint64_t *d_chunk_start,
                           const ghip_sketch_work *d_work, size_t n_work) {
    if (n_work == 0) return;
    const uint64_t thr = ~0ull / c;
    ghip_prof_begin(ctx, "ani_seeds");
    hipLaunchKernelGGL(ani_seeds_kernel, dim3((unsigned)n_work), dim3(GHIP_SKETCH_THREADS), 0, ctx->stream,
                       g->d_bytes, g->d_starts, g->d_lens, d_work, k, thr, chunk, d_seed_hash, d_seed_chunk,
                       d_seed_start, d_seed_count, d_chunk_total, d_chunk_start);
    ghip_prof_end(ctx);
}

void ghip_launch_ani_table(ghip_ctx *ctx, size_t n, const uint64_t *d_seed_hash, const uint64_t *d_seed_start,
                           const uint32_t *d_seed_count, uint64_t *d_table, const uint64_t *d_table_start) {
    if (n == 0) return;
    ghip_prof_begin(ctx, "ani_table");
    hipLaunchKernelGGL(ani_table_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_seed_hash, d_seed_start,
                       d_seed_count, d_table, d_table_start);
    ghip_prof_end(ctx);
}

void ghip_launch_ani_pairs(ghip_ctx *ctx, const ghip_ani_index *idx, const uint32_t *d_pairs, size_t n_pairs,
                           uint32_t max_chunks, uint64_t *d_out) {
    if (n_pairs == 0) return;
    const size_t lds = (size_t)max_chunks * sizeof(uint32_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(ani_pairs_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set = true;
    }
    ghip_prof_begin(ctx, "ani_pairs");
    hipLaunchKernelGGL(ani_pairs_kernel, dim3((unsigned)(2 * n_pairs)), dim3(256), lds, ctx->stream, d_pairs,
                       idx->d_seed_hash, idx->d_seed_chunk, idx->d_seed_start, idx->d_seed_count, idx->d_table,
                       idx->d_table_start, idx->d_chunk_total, idx->d_chunk_start, idx->d_glen, idx->chunk, d_out);
    ghip_prof_end(ctx);
}
