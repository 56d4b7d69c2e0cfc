// All-vs-all precluster pair stage on gfx950: replaces the serial i<j loop of
// finch::distances (reference src/finch.rs:74-96) and finch::distance::raw_distance.
//
// pair_intersect_tile: one workgroup per (A-tile, B-tile) of the upper triangle.  Both tiles
// of sorted u64 sketches are staged in LDS (row-padded by one element per 16 so that lanes
// walking 16-element-strided positions fall on distinct banks); each wavefront then takes
// one sketch pair at a time and intersects it with a 64-way merge path: every lane binary-
// searches its diagonal, merges ~(|A|+|B|)/64 steps and counts equal hashes.  The numbers
// raw_distance would produce follow in closed form:
//     common = |A n B|,  m = min(max A, max B),  i = #{a <= m},  j = #{b <= m},
//     total = i + j - common.
// No floating point on the device: a pair is emitted iff common >= cmin[total], a table the
// host derives from the reference's f64 formula, and the host recomputes the exact f32 ANI.
// Pure integer compare work: no MFMA.  Algorithmic bytes: 2*s*8 per pair.
#include <cstdlib>

#include "ghip_internal.h"

namespace {

constexpr int PAIR_WAVES = 16;
constexpr int PAIR_THREADS = PAIR_WAVES * 64;
constexpr int PAIR_CHAINS = 4;  // sketch pairs interleaved per wavefront

__device__ __forceinline__ uint32_t lds_pos(uint32_t e) { return e + (e >> 4); }

// #elements of sorted X[0..n) that are <= x  (all lanes compute the same value)
__device__ __forceinline__ uint32_t upper_bound_lds(const uint64_t *X, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (X[lds_pos(mid)] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(PAIR_THREADS) void pair_intersect_tile_kernel(
    const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens, uint32_t n, uint32_t s,
    uint32_t s_pad, uint32_t sp /* LDS elements per sketch */, uint32_t pt /* tile edge */,
    uint32_t nt /* tiles per dim */, uint64_t n_tilepairs, uint32_t rank, uint32_t world, uint32_t row_lo,
    const uint16_t *__restrict__ cmin, ghip_pair *__restrict__ out,
    unsigned long long *__restrict__ out_count, uint64_t cap, int dbg_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *lds = reinterpret_cast<uint64_t *>(smem_raw);

    // grid-stride over this rank's tile pairs: an AQL dispatch holds at most 2^32-1 work-items, so
    // the grid is capped on the host and each workgroup walks several tile pairs.
    for (uint64_t blk = blockIdx.x;; blk += gridDim.x) {
    const uint64_t t = blk * world + rank;
    if (t >= n_tilepairs) return;
    __syncthreads();  // every wave is done with the previous tile pair's LDS image
    // t -> (ti, tj), ti <= tj, rows of the upper triangle: row ti holds nt - ti tile pairs
    uint32_t ti;
    {
        double b = 2.0 * nt + 1.0;
        double r = (b - sqrt(b * b - 8.0 * (double)t)) * 0.5;
        ti = (uint32_t)r;
        if (ti >= nt) ti = nt - 1;
        auto row_off = [&](uint64_t x) { return x * nt - x * (x - 1) / 2; };
        while (ti > 0 && row_off(ti) > t) ti--;
        while (ti + 1 < nt && row_off(ti + 1) <= t) ti++;
    }
    const uint32_t tj = ti + (uint32_t)(t - ((uint64_t)ti * nt - (uint64_t)ti * (ti - 1) / 2));
    if ((uint64_t)(tj + 1) * pt <= row_lo) continue;   // incremental run: both tiles hold old genomes only

    // ---- stage 2*pt sketches into LDS (coalesced 8-B loads; pad rows with 2^64-1) ----
    for (uint32_t q = 0; q < 2 * pt; q++) {
        const uint32_t g = (q < pt) ? ti * pt + q : tj * pt + (q - pt);
        const uint64_t *row = hashes + (uint64_t)g * s;
        uint64_t *dst = lds + (uint64_t)q * sp;
        for (uint32_t e = threadIdx.x; e < s_pad; e += PAIR_THREADS) {
            uint64_t v = (g < n && e < s) ? row[e] : ~0ull;
            dst[lds_pos(e)] = v;
        }
    }
    __syncthreads();
    if (dbg_mode == 1) {  // GHIP_PAIR_DEBUG=1: staging only (timing experiments)
        if (threadIdx.x == 0 && lds[lds_pos(5)] == 12345ull) atomicAdd(out_count, 1ull);
        continue;
    }

    // Each wavefront intersects PAIR_CHAINS sketch pairs at once (independent merge chains are
    // interleaved per lane; the kernel is instruction-issue bound, so this mostly amortises the
    // loop and reduction overhead).  One sketch pair is still owned by one wavefront.
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t p0 = wave; p0 < pt * pt; p0 += PAIR_WAVES * PAIR_CHAINS) {
        const uint64_t *A[PAIR_CHAINS], *B[PAIR_CHAINS];
        uint32_t na[PAIR_CHAINS], nb[PAIR_CHAINS], gi[PAIR_CHAINS], gj[PAIR_CHAINS];
        uint32_t d0[PAIR_CHAINS], nstep[PAIR_CHAINS], lo[PAIR_CHAINS], hi[PAIR_CHAINS];
        uint32_t maxstep = 0;
#pragma unroll
        for (int c = 0; c < PAIR_CHAINS; c++) {
            const uint32_t p = p0 + c * PAIR_WAVES;
            const uint32_t qa = p / pt, qb = p % pt;
            gi[c] = ti * pt + qa; gj[c] = tj * pt + qb;
            const bool ok = p < pt * pt && gi[c] < n && gj[c] < n && gi[c] < gj[c];
            A[c] = lds + (uint64_t)(ok ? qa : 0u) * sp;
            B[c] = lds + (uint64_t)(pt + (ok ? qb : 0u)) * sp;
            na[c] = ok ? lens[gi[c]] : 0u;
            nb[c] = ok ? lens[gj[c]] : 0u;
            if (!ok) gi[c] = 0xffffffffu;
            const uint32_t tot = na[c] + nb[c];
            const uint32_t D = (tot + 63u) >> 6;
            d0[c] = min(lane * D, tot);
            nstep[c] = min(d0[c] + D, tot) - d0[c];
            maxstep = max(maxstep, D);
            lo[c] = d0[c] > nb[c] ? d0[c] - nb[c] : 0u;
            hi[c] = min(d0[c], na[c]);
        }
        // merge-path split of diagonal d0 (ties: a before b); <= 11 rounds for 1024-element lists
        for (;;) {
            bool more = false;
#pragma unroll
            for (int c = 0; c < PAIR_CHAINS; c++) more |= lo[c] < hi[c];
            if (!__any(more)) break;
            uint64_t av[PAIR_CHAINS], bv[PAIR_CHAINS];
#pragma unroll
            for (int c = 0; c < PAIR_CHAINS; c++) {
                const uint32_t mid = (lo[c] + hi[c]) >> 1;
                const bool act = lo[c] < hi[c];
                av[c] = A[c][lds_pos(act ? mid : 0u)];
                bv[c] = B[c][lds_pos(act ? d0[c] - 1 - mid : 0u)];
            }
#pragma unroll
            for (int c = 0; c < PAIR_CHAINS; c++) {
                const uint32_t mid = (lo[c] + hi[c]) >> 1;
                if (lo[c] < hi[c]) { if (av[c] <= bv[c]) lo[c] = mid + 1; else hi[c] = mid; }
            }
        }
        uint32_t ai[PAIR_CHAINS], bi[PAIR_CHAINS], common[PAIR_CHAINS];
        uint64_t a[PAIR_CHAINS], b[PAIR_CHAINS];
#pragma unroll
        for (int c = 0; c < PAIR_CHAINS; c++) {
            ai[c] = lo[c]; bi[c] = d0[c] - lo[c]; common[c] = 0;
            a[c] = A[c][lds_pos(ai[c])];   // slot [len] is padding (2^64-1)
            b[c] = B[c][lds_pos(bi[c])];
        }
        for (uint32_t st = 0; st < maxstep; st++) {
#pragma unroll
            for (int c = 0; c < PAIR_CHAINS; c++) {
                const bool run = st < nstep[c];
                const bool take_a = (ai[c] < na[c]) && ((bi[c] >= nb[c]) || (a[c] <= b[c]));
                common[c] += (run && take_a && (bi[c] < nb[c]) && (a[c] == b[c])) ? 1u : 0u;
                const bool adv_a = run && take_a, adv_b = run && !take_a;
                ai[c] += adv_a ? 1u : 0u;
                bi[c] += adv_b ? 1u : 0u;
                const uint64_t v = adv_a ? A[c][lds_pos(ai[c])] : B[c][lds_pos(bi[c])];
                if (adv_a) a[c] = v;
                if (adv_b) b[c] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < PAIR_CHAINS; c++) {
            uint32_t cm = common[c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cm += __shfl_xor(cm, off, 64);
            if (gi[c] == 0xffffffffu) continue;
            uint32_t icnt = 0, jcnt = 0;
            if (na[c] > 0 && nb[c] > 0) {
                const uint64_t maxa = A[c][lds_pos(na[c] - 1)], maxb = B[c][lds_pos(nb[c] - 1)];
                if (maxa <= maxb) { icnt = na[c]; jcnt = upper_bound_lds(B[c], nb[c], maxa); }
                else              { icnt = upper_bound_lds(A[c], na[c], maxb); jcnt = nb[c]; }
            }
            const uint32_t total = icnt + jcnt - cm;
            if (lane == 0 && cm >= (uint32_t)cmin[total] && gj[c] >= row_lo) {   // (row_lo: the rectangle of an incremental run)
                unsigned long long idx = atomicAdd(out_count, 1ull);
                if (idx < cap) {
                    ghip_pair r;
                    r.i = gi[c]; r.j = gj[c]; r.common = cm; r.total = total; r.ani = 0.0f;
                    out[idx] = r;
                }
            }
        }
    }
    }  // grid-stride loop
}


// Sketches too long for two LDS tiles (s > 4096; the reference puts no bound on num_kmers, src/finch.rs:55-61): one
// wavefront per pair straight from the packed matrix in global memory.  common = #{a in A : a in B} by binary search (a
// lane takes every 64th element of A), the ranks by the closed form of the reference's merge loop.  Slow (~na log nb
// dependent loads per pair) and rare: the inverted-index form takes such inputs first and only hands over what it declines.
__device__ __forceinline__ uint32_t upper_bound_global(const uint64_t *__restrict__ row, uint32_t len, uint64_t x) {
    uint32_t lo = 0, hi = len;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (row[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void pair_intersect_global_kernel(
    const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ lens, uint32_t n, uint32_t s, uint64_t n_pairs,
    uint32_t rank, uint32_t world, uint32_t row_lo, const uint16_t *__restrict__ cmin, ghip_pair *__restrict__ out,
    unsigned long long *__restrict__ out_count, uint64_t cap) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);; w += (uint64_t)gridDim.x * 4) {
        const uint64_t p = w * world + rank;
        if (p >= n_pairs) return;
        // p -> (i, j), i < j, rows of the strict upper triangle: row i holds n - 1 - i pairs
        uint32_t gi;
        {
            const double b = 2.0 * n - 1.0;
            double r = (b - sqrt(b * b - 8.0 * (double)p)) * 0.5;
            gi = (uint32_t)r;
            if (gi >= n - 1) gi = n - 2;
            auto row_off = [&](uint64_t x) { return x * (n - 1) - x * (x - 1) / 2; };
            while (gi > 0 && row_off(gi) > p) gi--;
            while (gi + 2 < n && row_off(gi + 1) <= p) gi++;
        }
        const uint32_t gj = gi + 1 + (uint32_t)(p - ((uint64_t)gi * (n - 1) - (uint64_t)gi * (gi - 1) / 2));
        if (gj < row_lo) continue;
        const uint32_t na = lens[gi], nb = lens[gj];
        const uint64_t *A = hashes + (uint64_t)gi * s, *B = hashes + (uint64_t)gj * s;
        uint32_t cm = 0;
        for (uint32_t e = lane; e < na; e += 64) {
            const uint64_t a = A[e];
            const uint32_t ub = upper_bound_global(B, nb, a);
            cm += (ub > 0 && B[ub - 1] == a) ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cm += __shfl_xor(cm, off, 64);
        uint32_t icnt = 0, jcnt = 0;
        if (na > 0 && nb > 0) {
            const uint64_t maxa = A[na - 1], maxb = B[nb - 1];
            if (maxa <= maxb) { icnt = na; jcnt = upper_bound_global(B, nb, maxa); }
            else              { icnt = upper_bound_global(A, na, maxb); jcnt = nb; }
        }
        const uint32_t total = icnt + jcnt - cm;
        if (lane == 0 && cm >= (uint32_t)cmin[total]) {
            unsigned long long idx = atomicAdd(out_count, 1ull);
            if (idx < cap) {
                ghip_pair r;
                r.i = gi; r.j = gj; r.common = cm; r.total = total; r.ani = 0.0f;
                out[idx] = r;
            }
        }
    }
}

}  // namespace

// Tile geometry shared with the host (ghip_precluster): sp elements of LDS per sketch,
// pt sketches per tile edge so that 2*pt*sp*8 bytes fit the 160 KiB LDS.
void ghip_pair_geometry(uint32_t s, uint32_t *s_pad, uint32_t *sp, uint32_t *pt) {
    uint32_t spad = ((s + 1 + 63) / 64) * 64;  // >= s+1 so slot [len] is always padding
    uint32_t per = spad + spad / 16;
    uint32_t fit = (160u * 1024u) / (2u * per * 8u);
    uint32_t t = fit >= 8 ? 8 : (fit >= 4 ? 4 : (fit >= 2 ? 2 : 1));
    *s_pad = spad; *sp = per; *pt = t;
}

void ghip_launch_pairs(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n,
                       uint32_t s, const uint16_t *d_cmin, uint32_t rank, uint32_t world, uint32_t row_lo,
                       ghip_pair *d_out, unsigned long long *d_count, uint64_t cap,
                       uint64_t *pairs_compared) {
    uint32_t s_pad, sp, pt;
    ghip_pair_geometry(s, &s_pad, &sp, &pt);
    const uint32_t nt = (uint32_t)((n + pt - 1) / pt);
    const uint64_t n_tilepairs = (uint64_t)nt * (nt + 1) / 2;
    const uint64_t my_tiles = n_tilepairs > rank ? (n_tilepairs - rank + world - 1) / world : 0;
    // pairs this shard compares (for the throughput metric)
    if (pairs_compared) {
        uint64_t cnt = 0;
        if (world == 1) cnt = (uint64_t)n * (n - 1) / 2;
        else {
            for (uint64_t t = rank; t < n_tilepairs; t += world) {
                // same (ti,tj) mapping as the kernel, on the host
                uint64_t ti = 0, acc = 0;
                {
                    double b = 2.0 * nt + 1.0;
                    double r = (b - __builtin_sqrt(b * b - 8.0 * (double)t)) * 0.5;
                    ti = (uint64_t)r;
                    if (ti >= nt) ti = nt - 1;
                    auto row_off = [&](uint64_t x) { return x * nt - x * (x - 1) / 2; };
                    while (ti > 0 && row_off(ti) > t) ti--;
                    while (ti + 1 < nt && row_off(ti + 1) <= t) ti++;
                    acc = row_off(ti);
                }
                uint64_t tj = ti + (t - acc);
                uint64_t ra = std::min<uint64_t>(pt, n - ti * pt), rb = std::min<uint64_t>(pt, n - tj * pt);
                cnt += (ti == tj) ? ra * (ra - 1) / 2 : ra * rb;
            }
        }
        *pairs_compared = cnt;
    }
    if (my_tiles == 0) return;
    const size_t lds_bytes = (size_t)2 * pt * sp * sizeof(uint64_t);
    ghip_ensure_dyn_lds(ctx, reinterpret_cast<const void *>(pair_intersect_tile_kernel), 160 * 1024);
    ghip_prof_begin(ctx, "pair_intersect_tile");
    const unsigned grid = (unsigned)std::min<uint64_t>(my_tiles, 1u << 21);  // x 1024 threads < 2^32 work-items
    hipLaunchKernelGGL(pair_intersect_tile_kernel, dim3(grid), dim3(PAIR_THREADS), lds_bytes,
                       ctx->stream, d_hashes, d_lens, (uint32_t)n, s, s_pad, sp, pt, nt, n_tilepairs, rank,
                       world, row_lo, d_cmin, d_out, d_count, cap, (int)ctx->opt.pair_debug);
    ghip_prof_end(ctx);
}

// s > 4096: the global-memory form (pair_intersect_global_kernel); *pairs_compared = the pairs of this rank's share
void ghip_launch_pairs_global(ghip_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_lens, size_t n, uint32_t s,
                              const uint16_t *d_cmin, uint32_t rank, uint32_t world, uint32_t row_lo, ghip_pair *d_out,
                              unsigned long long *d_count, uint64_t cap, uint64_t *pairs_compared) {
    const uint64_t P = (uint64_t)n * (n - 1) / 2;
    const uint64_t mine = P > rank ? (P - rank + world - 1) / world : 0;
    if (pairs_compared) *pairs_compared = mine;
    if (mine == 0) return;
    ghip_prof_begin(ctx, "pair_intersect_tile");
    hipLaunchKernelGGL(pair_intersect_global_kernel, dim3((unsigned)std::min<uint64_t>((mine + 3) / 4, GHIP_MAX_GRID)), dim3(256), 0, ctx->stream,
                       d_hashes, d_lens, (uint32_t)n, s, P, rank, world, row_lo, d_cmin, d_out, d_count, cap);
    ghip_prof_end(ctx);
}
