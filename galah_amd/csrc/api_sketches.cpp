// C-ABI of libgalah_hip.so, sketch matrices: handles, the persisted form (GHIPSK02), the MinHash stage (ghip_sketch_genomes / _files).
#include "api_internal.h"

using namespace ghip_api;

// ------------------------------------------------------------------------------------ sketches
void ghip_free_sketches_locked(ghip_sketches *sk) {  // ctx->mu held
    ghip_ctx *ctx = sk->ctx;
    if (sk->owned) { ghip_pool_free(ctx, sk->d_hashes); ghip_pool_free(ctx, sk->d_lens); }
    ghip_pool_free(ctx, sk->d_tables); ghip_pool_free(ctx, sk->d_tags); ghip_pool_free(ctx, sk->d_row_start); ghip_pool_free(ctx, sk->d_arranged);
    ctx->live_handles--;
    delete sk;
}

extern "C" void ghip_sketches_free(ghip_sketches *sk) {
    if (!sk) return;
    ghip_ctx *ctx = sk->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStreamSynchronize(ctx->stream);
        ghip_free_sketches_locked(sk);
    }
    ghip_ctx_release(ctx);
}

extern "C" size_t ghip_sketches_count(const ghip_sketches *sk) { return sk ? sk->n : 0; }
extern "C" uint32_t ghip_sketches_size(const ghip_sketches *sk) { return sk ? sk->s : 0; }
extern "C" uint32_t ghip_sketches_kmer(const ghip_sketches *sk) { return sk ? sk->k : 0; }
extern "C" void *ghip_sketches_device_hashes(const ghip_sketches *sk) { return sk ? sk->d_hashes : nullptr; }
extern "C" void *ghip_sketches_device_lens(const ghip_sketches *sk) { return sk ? sk->d_lens : nullptr; }

extern "C" int ghip_sketches_from_host(ghip_ctx *ctx, const uint64_t *hashes, const uint32_t *lens, size_t n,
                                       uint32_t s, uint32_t k, ghip_sketches **out) {
    if (!ctx || !out || s == 0 || (n && (!hashes || !lens))) return GHIP_EINVAL;
    for (size_t i = 0; i < n; i++) {
        if (lens[i] > s) return ghip_set_error(ctx, GHIP_EINVAL, "sketch length exceeds sketch size");
        for (uint32_t e = 1; e < lens[i]; e++)
            if (hashes[i * s + e] <= hashes[i * s + e - 1]) return ghip_set_error(ctx, GHIP_EINVAL, "sketch hashes must be strictly ascending");
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = n; sk->s = s; sk->k = k;
    std::vector<uint64_t> padded(hashes, hashes + n * (size_t)s);
    for (size_t i = 0; i < n; i++) for (uint32_t e = lens[i]; e < s; e++) padded[i * s + e] = ~0ull;
    int rc = dmalloc(ctx, &sk->d_hashes, n * (size_t)s);
    if (!rc) rc = dmalloc(ctx, &sk->d_lens, n);
    if (!rc) rc = h2d(ctx, sk->d_hashes, padded.data(), n * (size_t)s);
    if (!rc) rc = h2d(ctx, sk->d_lens, lens, n);
    if (rc) { ghip_free_sketches_locked(sk); return rc; }
    *out = sk;
    return GHIP_OK;
}

extern "C" int ghip_sketches_wrap_device(ghip_ctx *ctx, void *d_hashes, void *d_lens, size_t n, uint32_t s,
                                         uint32_t k, ghip_sketches **out) {
    if (!ctx || !out || s == 0 || (n && (!d_hashes || !d_lens))) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = n; sk->s = s; sk->k = k;
    sk->d_hashes = (uint64_t *)d_hashes; sk->d_lens = (uint32_t *)d_lens; sk->owned = false;
    *out = sk;
    return GHIP_OK;
}

extern "C" int ghip_sketches_to_host(ghip_ctx *ctx, const ghip_sketches *sk, uint64_t *hashes, uint32_t *lens) {
    if (!ctx || !sk) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = GHIP_OK;
    if (hashes) rc = d2h(ctx, hashes, sk->d_hashes, sk->n * (size_t)sk->s);
    if (!rc && lens) rc = d2h(ctx, lens, sk->d_lens, sk->n);
    return rc;
}

extern "C" int ghip_sketches_copy_into(ghip_ctx *ctx, const ghip_sketches *sk, void *d_hashes_dst, void *d_lens_dst) {
    if (!ctx || !sk || !d_hashes_dst || !d_lens_dst) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sk->n) {
        GHIP_HIP_CHECK(ctx, hipMemcpyAsync(d_hashes_dst, sk->d_hashes, sk->n * (size_t)sk->s * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        GHIP_HIP_CHECK(ctx, hipMemcpyAsync(d_lens_dst, sk->d_lens, sk->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return GHIP_OK;
}

// On-disk sketch matrix (the reference has no finch counterpart -- skani's --low-memory db dir,
// src/skani.rs:266-304, and the sketched reference set of its --reference-genomes mode, src/skani.rs:502-565, are the
// closest).  "GHIPSK02", little-endian:
//   char[8] magic; u32 k; u32 s; u64 hash seed; u64 n; u64 names_bytes; u32 len[n]; u64 hashes[n][s];
//   char names[names_bytes] (n NUL-terminated genome names, in row order); u64 FNV-1a-64 of every byte before it.
// "GHIPSK01" (round 1-2: no seed, no names, no checksum) still loads.
namespace {
struct Fnv {
    uint64_t h = 0xcbf29ce484222325ull;
    void add(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; } }
};
bool put(FILE *f, Fnv &c, const void *p, size_t n) { c.add(p, n); return n == 0 || fwrite(p, 1, n, f) == n; }
bool get(FILE *f, Fnv &c, void *p, size_t n) { if (n && fread(p, 1, n, f) != n) return false; c.add(p, n); return true; }
}  // namespace

extern "C" int ghip_sketches_save_named(ghip_ctx *ctx, const ghip_sketches *sk, const char *const *names, uint64_t seed, const char *path) {
    if (!ctx || !sk || !path) return GHIP_EINVAL;
    std::vector<uint64_t> h(sk->n * (size_t)sk->s);
    std::vector<uint32_t> l(sk->n);
    int rc = ghip_sketches_to_host(ctx, sk, h.data(), l.data());
    if (rc) return rc;
    std::string blob;
    for (size_t i = 0; i < sk->n; i++) { if (names && names[i]) blob += names[i]; blob.push_back('\0'); }
    FILE *f = fopen(path, "wb");
    if (!f) return ghip_set_error(ctx, GHIP_EIO, std::string("cannot write ") + path);
    const uint64_t n = sk->n, nb = blob.size();
    Fnv c;
    bool ok = put(f, c, "GHIPSK02", 8) && put(f, c, &sk->k, 4) && put(f, c, &sk->s, 4) && put(f, c, &seed, 8) && put(f, c, &n, 8) &&
              put(f, c, &nb, 8) && put(f, c, l.data(), 4 * l.size()) && put(f, c, h.data(), 8 * h.size()) && put(f, c, blob.data(), blob.size());
    ok = ok && fwrite(&c.h, 8, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    return ok ? GHIP_OK : ghip_set_error(ctx, GHIP_EIO, std::string("short write to ") + path);
}

extern "C" int ghip_sketches_save(ghip_ctx *ctx, const ghip_sketches *sk, const char *path) {
    return ghip_sketches_save_named(ctx, sk, nullptr, 0, path);
}

extern "C" int ghip_sketches_load_named(ghip_ctx *ctx, const char *path, ghip_sketches **out, char **out_names, size_t *out_names_bytes, uint64_t *out_seed) {
    if (!ctx || !path || !out) return GHIP_EINVAL;
    if (out_names) *out_names = nullptr;
    if (out_names_bytes) *out_names_bytes = 0;
    if (out_seed) *out_seed = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return ghip_set_error(ctx, GHIP_EIO, std::string("cannot read ") + path);
    char magic[8];
    uint32_t k = 0, s = 0;
    uint64_t n = 0, seed = 0, nb = 0;
    Fnv c;
    bool ok = get(f, c, magic, 8);
    const bool v2 = ok && !memcmp(magic, "GHIPSK02", 8);
    ok = ok && (v2 || !memcmp(magic, "GHIPSK01", 8)) && get(f, c, &k, 4) && get(f, c, &s, 4);
    if (ok && v2) ok = get(f, c, &seed, 8);
    ok = ok && get(f, c, &n, 8);
    if (ok && v2) ok = get(f, c, &nb, 8);
    ok = ok && s >= 1 && s <= GHIP_MAX_SKETCH_SIZE && n < (1ull << 32) && nb < (1ull << 40);
    std::vector<uint32_t> l;
    std::vector<uint64_t> h;
    std::string blob;
    std::string why = "not a sketch matrix file: ";
    if (ok) {
        // the header's counts are believed only when the FILE is exactly as long as they say (a truncated or
        // corrupt header could ask for terabytes before the checksum was ever looked at), and an allocation that still fails
        // is an error code, not an exception through the C boundary
        struct stat st;
        const uint64_t header = 8 + 4 + 4 + (v2 ? 8 : 0) + 8 + (v2 ? 8 : 0);
        const uint64_t want = header + 4 * n + 8 * n * (uint64_t)s + nb + (v2 ? 8 : 0);
        if (fstat(fileno(f), &st) != 0 || (uint64_t)st.st_size != want) { ok = false; why = "sketch matrix file is truncated or its header is damaged (size): "; }
    }
    if (ok) {
        try { l.resize(n); h.resize(n * (size_t)s); blob.resize(nb); }
        catch (const std::exception &) { fclose(f); return ghip_set_error(ctx, GHIP_ENOMEM, std::string("out of host memory loading ") + path); }
        ok = get(f, c, l.data(), 4 * l.size()) && get(f, c, h.data(), 8 * h.size()) && get(f, c, &blob[0], nb);
    }
    if (ok && v2) {
        uint64_t sum = 0;
        ok = fread(&sum, 8, 1, f) == 1 && sum == c.h;
        if (!ok) why = "sketch matrix file is damaged (checksum): ";
        else if ((size_t)std::count(blob.begin(), blob.end(), '\0') != n) { ok = false; why = "sketch matrix file: name table does not match the row count: "; }
    }
    fclose(f);
    if (!ok) return ghip_set_error(ctx, GHIP_EIO, why + path);
    const int rc = ghip_sketches_from_host(ctx, h.data(), l.data(), n, s, k, out);  // validates order and lengths
    if (rc) return rc;
    if (out_names) {
        if (!v2) blob.assign(n, '\0');   // a GHIPSK01 file holds no names
        char *p = (char *)malloc(std::max<size_t>(blob.size(), 1));
        if (!p) { ghip_sketches_free(*out); *out = nullptr; return GHIP_ENOMEM; }
        memcpy(p, blob.data(), blob.size());
        *out_names = p;
        if (out_names_bytes) *out_names_bytes = blob.size();
    }
    if (out_seed) *out_seed = seed;
    return GHIP_OK;
}

extern "C" int ghip_sketches_load(ghip_ctx *ctx, const char *path, ghip_sketches **out) {
    return ghip_sketches_load_named(ctx, path, out, nullptr, nullptr, nullptr);
}

// rows of a followed by the rows of b (same k and s): the matrix of an incremental run = the saved one + the new genomes'
extern "C" int ghip_sketches_concat(ghip_ctx *ctx, const ghip_sketches *a, const ghip_sketches *b, ghip_sketches **out) {
    if (!ctx || !a || !b || !out) return GHIP_EINVAL;
    if (a->s != b->s || a->k != b->k) return ghip_set_error(ctx, GHIP_EINVAL, "sketch matrices differ in sketch size or k-mer length");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = a->n + b->n; sk->s = a->s; sk->k = a->k;
    int rc = dmalloc(ctx, &sk->d_hashes, sk->n * (size_t)sk->s);
    if (!rc) rc = dmalloc(ctx, &sk->d_lens, sk->n);
    const ghip_sketches *parts[2] = {a, b};
    size_t at = 0;
    for (int x = 0; x < 2 && !rc; x++) {
        const ghip_sketches *p = parts[x];
        if (p->n && (hipMemcpyAsync(sk->d_hashes + at * sk->s, p->d_hashes, p->n * (size_t)sk->s * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                     hipMemcpyAsync(sk->d_lens + at, p->d_lens, p->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess))
            rc = ghip_set_error(ctx, GHIP_EHIP, "sketch concatenation failed");
        at += p->n;
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ghip_set_error(ctx, GHIP_EHIP, "sketch concatenation failed");
    if (rc) { ghip_free_sketches_locked(sk); return rc; }
    *out = sk;
    return GHIP_OK;
}

// finch::sketch_files replacement.  Exactness: a genome is accepted only when its candidate
// list did not overflow and held >= s distinct hashes (or the threshold was already 2^64-1).
int ghip_sketch_genomes_locked(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s, uint64_t seed,
                                 const ghip_seed_args *seeds, ghip_sketches **out) {
    if (k < 1 || k > 32) return ghip_set_error(ctx, GHIP_EINVAL, "kmer_length must be in 1..=32");
    if (s < 1 || s > GHIP_MAX_SKETCH_SIZE) return ghip_set_error(ctx, GHIP_EINVAL, "num_kmers must be in 1..=65535");
    if (seed > 0xffffffffull) return ghip_set_error(ctx, GHIP_EINVAL, "hash_seed must fit 32 bits (murmurhash3 seed)");
    const size_t n = g->n;
    ghip_sketches *sk = new ghip_sketches();
    ctx->live_handles++;
    sk->ctx = ctx; sk->n = n; sk->s = s; sk->k = k;
    int rc = dmalloc(ctx, &sk->d_hashes, n * (size_t)s);
    if (!rc) rc = dmalloc(ctx, &sk->d_lens, n);
    if (rc) { ghip_free_sketches_locked(sk); return rc; }

    // pending slots: initially every genome, with a threshold that lets ~1.5*s hashes survive (for s = 1000:
    // 1500 +- 39, so fewer than s survive only on repetitive or tiny genomes, which the retry loop widens)
    std::vector<uint32_t> slot_genome(n);
    std::vector<uint64_t> slot_thr(n);
    std::vector<uint32_t> slot_cap(n);
    const uint32_t cap0 = next_pow2(2ull * s + 64);
    for (size_t i = 0; i < n; i++) {
        slot_genome[i] = (uint32_t)i;
        uint64_t nk = g->lens[i] >= k ? g->lens[i] - k + 1 : 1;
        unsigned __int128 want = ((unsigned __int128)(3ull * s / 2 + 8)) << 64;
        unsigned __int128 t = want / nk;
        slot_thr[i] = (t >> 64) ? ~0ull : (uint64_t)t;
        slot_cap[i] = cap0;
    }
    bool first = true;
    int iter = 0;
    while (!slot_genome.empty()) {
        if (++iter > 80) { ghip_free_sketches_locked(sk); return ghip_set_error(ctx, GHIP_EHIP, "sketch selection did not converge"); }
        const size_t ns = slot_genome.size();
        std::vector<uint64_t> cand_start(ns);
        uint64_t total_cand = 0;
        for (size_t i = 0; i < ns; i++) { cand_start[i] = total_cand; total_cand += slot_cap[i]; }
        DeviceFree tmp(ctx);
        uint32_t *d_slot_genome = nullptr, *d_cap = nullptr, *d_count = nullptr, *d_status = nullptr;
        uint64_t *d_thr = nullptr, *d_cstart = nullptr, *d_cand = nullptr;
        ghip_sketch_work *d_work = nullptr;
        size_t n_work = 0;
        if ((rc = dmalloc(ctx, &d_thr, ns))) break; tmp.add(d_thr);
        if ((rc = dmalloc(ctx, &d_cstart, ns))) break; tmp.add(d_cstart);
        if ((rc = dmalloc(ctx, &d_cap, ns))) break; tmp.add(d_cap);
        if ((rc = dmalloc(ctx, &d_count, ns))) break; tmp.add(d_count);
        if ((rc = dmalloc(ctx, &d_status, ns))) break; tmp.add(d_status);
        if ((rc = dmalloc(ctx, &d_cand, total_cand))) break; tmp.add(d_cand);
        if (first) {
            d_slot_genome = g->d_identity;
            d_work = g->d_work;
            n_work = g->n_work;
        } else {
            std::vector<ghip_sketch_work> work;
            for (size_t i = 0; i < ns; i++) {
                uint64_t chunks = (g->lens[slot_genome[i]] + GHIP_SKETCH_CHUNK - 1) / GHIP_SKETCH_CHUNK;
                for (uint64_t c = 0; c < chunks; c++) work.push_back({(uint32_t)i, (uint32_t)c});
            }
            n_work = work.size();
            if ((rc = dmalloc(ctx, &d_slot_genome, ns))) break; tmp.add(d_slot_genome);
            if ((rc = dmalloc(ctx, &d_work, n_work))) break; tmp.add(d_work);
            if ((rc = h2d(ctx, d_slot_genome, slot_genome.data(), ns))) break;
            if ((rc = h2d(ctx, d_work, work.data(), n_work))) break;
        }
        if ((rc = h2d_nosync(ctx, d_thr, slot_thr.data(), ns))) break;
        if ((rc = h2d_nosync(ctx, d_cstart, cand_start.data(), ns))) break;
        if ((rc = h2d_nosync(ctx, d_cap, slot_cap.data(), ns))) break;
        if (hipMemsetAsync(d_count, 0, ns * sizeof(uint32_t), ctx->stream) != hipSuccess) { rc = ghip_set_error(ctx, GHIP_EHIP, "memset failed"); break; }
        // (no synchronisation here: the three host vectors stay as they are until the one after the kernels)
        ghip_launch_sketch_kmers(ctx, g->d_packed, g->d_valid, g->d_starts, g->d_lens, d_slot_genome, d_thr, d_cstart, d_cap,
                                 d_work, n_work, k, (uint32_t)seed, d_cand, d_count, first ? seeds : nullptr);
        ghip_launch_sketch_select(ctx, d_slot_genome, ns, d_cand, d_count, d_cstart, d_cap, *std::max_element(slot_cap.begin(), slot_cap.end()), s, sk->d_hashes,
                                  sk->d_lens, d_status);
        std::vector<uint32_t> status(ns), count(ns);
        if ((rc = d2h_nosync(ctx, status.data(), d_status, ns))) break;
        if ((rc = d2h_nosync(ctx, count.data(), d_count, ns))) break;
        if ((rc = stream_sync(ctx))) break;
        { hipError_t e = hipGetLastError(); if (e != hipSuccess) { rc = ghip_set_error(ctx, GHIP_EHIP, std::string("sketch kernels: ") + hipGetErrorString(e)); break; } }
        std::vector<uint32_t> ng; std::vector<uint64_t> nthr; std::vector<uint32_t> ncap;
        for (size_t i = 0; i < ns; i++) {
            if (status[i] & 1u) {  // overflow: same threshold, list as large as the survivor count
                ng.push_back(slot_genome[i]); nthr.push_back(slot_thr[i]); ncap.push_back(next_pow2(count[i]));
            } else if ((status[i] & 2u) && slot_thr[i] != ~0ull) {  // too few distinct: widen 8x
                uint64_t t = slot_thr[i];
                ng.push_back(slot_genome[i]);
                nthr.push_back(t > (~0ull >> 3) ? ~0ull : t << 3);
                ncap.push_back(std::max<uint32_t>(slot_cap[i], next_pow2(8ull * count[i] + 64)));
            }
        }
        slot_genome.swap(ng); slot_thr.swap(nthr); slot_cap.swap(ncap);
        first = false;
    }
    if (rc) { ghip_free_sketches_locked(sk); return rc; }
    *out = sk;
    return GHIP_OK;
}

extern "C" int ghip_sketch_genomes(ghip_ctx *ctx, const ghip_genomes *g, uint32_t k, uint32_t s, uint64_t seed,
                                   ghip_sketches **out) {
    if (!ctx || !g || !out) return GHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    GHIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return ghip_sketch_genomes_locked(ctx, g, k, s, seed, nullptr, out);
}

extern "C" int ghip_sketch_and_index_files(ghip_ctx *ctx, const char *const *paths, size_t n, uint32_t k, uint32_t s, uint64_t seed,
                                           uint32_t ani_k, uint32_t ani_c, uint32_t ani_chunk, int io_threads, uint64_t batch_bytes,
                                           ghip_sketches **out_sk, ghip_ani_index **out_idx, uint64_t *out_stats);

// finch::sketch_files for a file list (src/finch.rs:55-69); inputs larger than HBM are sketched in batches
extern "C" int ghip_sketch_files(ghip_ctx *ctx, const char *const *paths, size_t n, uint32_t k, uint32_t s,
                                 uint64_t seed, int io_threads, ghip_sketches **out) {
    return ghip_sketch_and_index_files(ctx, paths, n, k, s, seed, 0, 0, 0, io_threads, 0, out, nullptr, nullptr);
}

