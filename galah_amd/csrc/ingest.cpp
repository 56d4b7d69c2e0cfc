// Host-side FASTA ingest: file -> device-format base stream (see galah_hip.h "genome ingest").
//
// Mirrors what finch::sketch_files does before hashing (reference src/finch.rs:69):
// needletail parse_fastx_file (plain or gzip, '>' records, multi-line sequences) and
// Sequence::normalize(iupac=false).  One stream per file: every record's normalised bytes
// followed by a single 'N' so that no k-mer spans two records.
#include <dlfcn.h>
#include <sys/stat.h>
#include <zlib.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>

#include "ghip_internal.h"

namespace {

// needletail normalize(): 0 = drop (whitespace), otherwise the output byte
struct NormTable {
    uint8_t t[256];
    NormTable() {
        for (int c = 0; c < 256; c++) t[c] = 'N';
        t['A'] = 'A'; t['C'] = 'C'; t['G'] = 'G'; t['T'] = 'T';
        t['a'] = 'A'; t['c'] = 'C'; t['g'] = 'G';
        t['t'] = 'T'; t['u'] = 'T'; t['U'] = 'T';
        t['-'] = '-'; t['.'] = '-'; t['~'] = '-';
        t[' '] = 0; t['\t'] = 0; t['\r'] = 0; t['\n'] = 0;
    }
};
const NormTable kNorm;

// Fast path of the line loop: a line made only of ACGTacgt (nearly every line of an assembly) is upper-cased 32 bytes
// at a time; the last block overlaps the one before it, so an 80-column line is three loads and three stores.
// Returns false (dst possibly half written) as soon as any other byte shows up -- the table loop then redoes the line.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) bool line_acgt_avx2(const uint8_t *src, size_t len, uint8_t *dst) {
    const __m256i up = _mm256_set1_epi8((char)0xDF), A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'),
                  G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T');
    for (size_t q = 0;;) {
        const __m256i u = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(src + q)), up);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, A), _mm256_cmpeq_epi8(u, C)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(u, G), _mm256_cmpeq_epi8(u, T)));
        _mm256_storeu_si256((__m256i *)(dst + q), u);
        if (_mm256_movemask_epi8(ok) != -1) return false;
        if (q + 32 >= len) return true;
        q = q + 64 <= len ? q + 32 : len - 32;
    }
}
const bool kHaveAvx2 = __builtin_cpu_supports("avx2");
#else
bool line_acgt_avx2(const uint8_t *, size_t, uint8_t *) { return false; }
const bool kHaveAvx2 = false;
#endif

// libdeflate, when the host has it (libdeflate.so.0 ships with the ROCm image), inflates a whole gzip member in one call
// about 2-3x faster than zlib's streaming inflate.  It is optional and bound at run time (no header in the image):
// any answer other than success is handed to the zlib loop below, which alone decides what is a read error.
struct LibDeflate {
    void *(*alloc)() = nullptr;
    void (*release)(void *) = nullptr;
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    LibDeflate() {
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        gunzip = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        if (!alloc || !release || !gunzip) alloc = nullptr;
    }
};

// All members of the gzip image `raw` into dst (sized from the last member's ISIZE, doubled while a member lacks room).
bool gunzip_libdeflate(const std::vector<uint8_t> &raw, std::vector<uint8_t> &dst, size_t isize) {
    static const LibDeflate lib;
    if (!lib.alloc || !ghip_process_options().use_libdeflate) return false;   // (process-wide: the parser has no context)
    void *d = lib.alloc();
    if (!d) return false;
    dst.resize(std::max<size_t>(isize, 1 << 16));
    size_t in_pos = 0, out_pos = 0;
    bool ok = true;
    while (in_pos < raw.size()) {
        size_t used = 0, made = 0;
        const int r = lib.gunzip(d, raw.data() + in_pos, raw.size() - in_pos, dst.data() + out_pos, dst.size() - out_pos, &used, &made);
        if (r == 3 /* LIBDEFLATE_INSUFFICIENT_SPACE */ && dst.size() < (1ull << 40)) { dst.resize(dst.size() * 2); continue; }
        if (r != 0 || used == 0) { ok = false; break; }
        in_pos += used;
        out_pos += made;
    }
    lib.release(d);
    if (ok) dst.resize(out_pos);
    return ok;
}

// Whole file into `buf`.  Plain files: one read().  gzip (magic 1f 8b; multi-member too): the compressed image is read
// in one go and inflated straight into a buffer sized from the trailer's ISIZE (grown if a further member follows) --
// about twice as fast as gzread's buffered loop.  Anything zlib cannot inflate is a read error.
bool slurp(const char *path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) { fclose(f); return false; }
    // The compressed image of a gzip file lives in a per-thread buffer, the inflated bytes go straight into `buf`: both
    // keep their pages from file to file (the ingest workers are persistent threads).  Swapping the caller's buffer out
    // for the compressed image, as this did before, made every gzip file page-fault a fresh multi-megabyte buffer -- with
    // 64 threads under one mmap lock.
    static thread_local std::vector<uint8_t> comp;
    unsigned char magic[2] = {0, 0};
    const size_t got = fread(magic, 1, 2, f);
    const bool gz = sz >= 18 && got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    std::vector<uint8_t> &first = gz ? comp : buf;
    first.resize((size_t)sz);
    if (got) memcpy(first.data(), magic, got);
    const bool ok_read = (size_t)sz <= got || fread(first.data() + got, 1, (size_t)sz - got, f) == (size_t)sz - got;
    fclose(f);
    if (!ok_read) return false;
    if (!gz) return true;  // plain
    std::vector<uint8_t> &raw = comp;
    std::vector<uint8_t> &dst = buf;
    size_t isize = (size_t)raw[sz - 4] | ((size_t)raw[sz - 3] << 8) | ((size_t)raw[sz - 2] << 16) | ((size_t)raw[sz - 1] << 24);
    isize = std::min<size_t>(isize, raw.size() * 16 + (1 << 16));  // only a first guess: a damaged trailer must not cost gigabytes
    if (gunzip_libdeflate(raw, dst, isize)) return true;
    dst.resize(std::max<size_t>(isize, 1 << 16));
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    zs.next_in = raw.data();
    zs.avail_in = 0;
    size_t in_pos = 0, out_pos = 0;
    bool ok = true;
    for (;;) {
        if (zs.avail_in == 0) {
            const size_t chunk = std::min<size_t>(raw.size() - in_pos, 1u << 30);
            zs.next_in = raw.data() + in_pos;
            zs.avail_in = (uInt)chunk;
            in_pos += chunk;
        }
        if (out_pos == dst.size()) dst.resize(dst.size() + dst.size() / 2 + (1 << 16));
        const size_t room = std::min<size_t>(dst.size() - out_pos, 1u << 30);
        zs.next_out = dst.data() + out_pos;
        zs.avail_out = (uInt)room;
        const int r = inflate(&zs, Z_NO_FLUSH);
        out_pos += room - zs.avail_out;
        if (r == Z_STREAM_END) {
            const size_t left = zs.avail_in + (raw.size() - in_pos);
            if (left == 0) break;
            if (inflateReset(&zs) != Z_OK) { ok = false; break; }  // next gzip member
            continue;
        }
        if (r != Z_OK && r != Z_BUF_ERROR) { ok = false; break; }
        if (r == Z_BUF_ERROR && zs.avail_in == 0 && in_pos == raw.size()) { ok = false; break; }  // truncated
    }
    inflateEnd(&zs);
    if (!ok) return false;
    dst.resize(out_pos);
    return true;
}

}  // namespace

// Parses a whole FASTA file image into the device-format stream: per record the normalised bytes, then one 'N'.
// Writes at most `cap` bytes to `out`; *out_len is the full stream length even when it exceeds cap (the caller
// then re-parses with room).  The same pass yields the assembly statistics galah computes in a second read of
// every file (reference src/genome_stats.rs:11-51): records, raw 'N'/'n' count, N50 over record lengths (bases,
// line ends excluded).  Line-oriented: memchr finds the line ends, the table maps a line's bytes in a tight loop.
int ghip_parse_fasta(const uint8_t *buf, size_t n, const char *path, uint8_t *out, size_t cap, size_t *out_len,
                     ghip_genome_stats &st, std::string &err) {
    st = ghip_genome_stats();
    *out_len = 0;
    std::vector<uint64_t> contig_lengths;
    size_t p = 0, m = 0;
    while (p < n && (buf[p] == '\n' || buf[p] == '\r')) p++;
    if (p == n) return GHIP_OK;  // empty file: empty stream, empty sketch
    if (buf[p] != '>') { err = std::string("Not a FASTA file (no '>' header): ") + path; return GHIP_EIO; }
    uint64_t amb = 0;
    while (p < n) {
        const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', n - p);  // header line
        p = nl ? (size_t)(nl - buf) + 1 : n;
        uint64_t bases = 0;
        while (p < n && buf[p] != '>') {  // sequence lines up to the next line that starts with '>'
            nl = (const uint8_t *)memchr(buf + p, '\n', n - p);
            const size_t e = nl ? (size_t)(nl - buf) : n;
            size_t len = e - p;
            uint64_t cr = 0;
            const size_t crlf = len && buf[e - 1] == '\r';  // a CRLF line: the fast path takes the part before the '\r'
            if (m + len <= cap && len - crlf >= 32 && kHaveAvx2 && line_acgt_avx2(buf + p, len - crlf, out + m)) {
                m += len - crlf;
                cr = crlf;
            } else if (m + len <= cap) {
                uint8_t *o = out + m;
                size_t w = 0;
                for (size_t q = p; q < e; q++) {
                    const uint8_t c = buf[q];
                    const uint8_t t = kNorm.t[c];
                    o[w] = t;
                    w += (t != 0);
                    amb += (c == 'N') | (c == 'n');
                    cr += (c == '\r');
                }
                m += w;
            } else {  // over capacity: only count
                for (size_t q = p; q < e; q++) {
                    const uint8_t c = buf[q];
                    m += (kNorm.t[c] != 0);
                    amb += (c == 'N') | (c == 'n');
                    cr += (c == '\r');
                }
            }
            bases += len - cr;
            p = nl ? e + 1 : n;
        }
        if (m < cap) out[m] = 'N';
        m++;
        st.num_contigs++;
        contig_lengths.push_back(bases);
    }
    st.num_ambiguous_bases = amb;
    *out_len = m;
    // genome_stats.rs:33-45: ascending lengths, first running sum >= total/2
    std::sort(contig_lengths.begin(), contig_lengths.end());
    uint64_t total = 0, run = 0;
    for (uint64_t l : contig_lengths) total += l;
    for (uint64_t l : contig_lengths) {
        run += l;
        if (run >= total / 2) { st.n50 = l; break; }
    }
    return GHIP_OK;
}

// 2-bit packing of a device-format stream for the trip over PCIe (sketch.hip: unpack_bases_kernel has the layout).  The
// fast path turns 32 A/C/G/T bytes into 8 packed bytes with a dozen AVX2 instructions; a block holding anything else
// (the 'N' after a record, ambiguity codes, gaps) goes byte by byte and extends the run table.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) static size_t pack_acgt_avx2(const uint8_t *src, size_t len, uint8_t *dst) {
    const __m256i A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T');
    const __m256i three = _mm256_set1_epi8(3), m14 = _mm256_set1_epi16(0x0401), m116 = _mm256_set1_epi32(0x00100001);
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    size_t q = 0;
    for (; q + 32 <= len; q += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + q));
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, A), _mm256_cmpeq_epi8(v, C)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(v, G), _mm256_cmpeq_epi8(v, T)));
        if (_mm256_movemask_epi8(ok) != -1) break;
        const __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);
        const __m256i p2 = _mm256_maddubs_epi16(code, m14);          // c0 + 4 c1 per 16-bit lane
        const __m256i p4 = _mm256_madd_epi16(p2, m116);              // + 16 (c2 + 4 c3) per 32-bit lane: one packed byte
        const __m256i b = _mm256_shuffle_epi8(p4, pick);             // the four packed bytes of each half in its low dword
        const uint32_t lo = (uint32_t)_mm256_cvtsi256_si32(b), hi = (uint32_t)_mm256_extract_epi32(b, 4);
        memcpy(dst + q / 4, &lo, 4);
        memcpy(dst + q / 4 + 4, &hi, 4);
    }
    return q;   // bases packed (a multiple of 32)
}
#else
static size_t pack_acgt_avx2(const uint8_t *, size_t, uint8_t *) { return 0; }
#endif

namespace {
struct RunTable {   // (start, length, byte) of the stream bytes that are not A/C/G/T, in stream order
    uint32_t *runs = nullptr;
    size_t n = 0, cap = 0;
    bool add(size_t pos, uint8_t c) {
        if (n && runs[3 * (n - 1) + 2] == c && (size_t)runs[3 * (n - 1)] + runs[3 * (n - 1) + 1] == pos) { runs[3 * (n - 1) + 1]++; return true; }
        if (n == cap) return false;
        runs[3 * n] = (uint32_t)pos; runs[3 * n + 1] = 1; runs[3 * n + 2] = c;
        n++;
        return true;
    }
};

// packs src[0, count) = stream[pos0, pos0 + count) into dst (pos0 a multiple of 4; the bytes of dst it touches are overwritten)
bool pack_range(const uint8_t *src, size_t count, size_t pos0, uint8_t *dst, RunTable &rt) {
    size_t p = 0;
    while (p < count) {
        if (kHaveAvx2 && ((pos0 + p) & 3) == 0) p += pack_acgt_avx2(src + p, count - p, dst + (pos0 + p) / 4);
        const size_t stop = std::min(count, (p & ~(size_t)31) + 32);   // up to the next multiple of 32 (or the end) byte by byte
        for (; p < stop; p++) {
            const uint8_t c = src[p];
            uint32_t code = 0;
            if (c == 'A' || c == 'C' || c == 'G' || c == 'T') code = ((c >> 1) ^ (c >> 2)) & 3u;
            else if (!rt.add(pos0 + p, c)) return false;
            const size_t q = pos0 + p, bit = 2 * (q & 3);
            if (bit == 0) dst[q / 4] = (uint8_t)code;
            else dst[q / 4] |= (uint8_t)(code << bit);
        }
    }
    return true;
}
}  // namespace

bool ghip_pack_stream(const uint8_t *stream, size_t len, uint8_t *dst, size_t dst_bytes, size_t *used, size_t *runs_off, uint32_t *n_runs) {
    if (len >= (1ull << 32)) return false;
    const size_t packed_bytes = ((len + 3) / 4 + 15) / 16 * 16;
    if (packed_bytes + 12 > dst_bytes) return false;
    RunTable rt{reinterpret_cast<uint32_t *>(dst + packed_bytes), 0, (dst_bytes - packed_bytes) / 12};
    memset(dst + (len / 4), 0, packed_bytes - len / 4);   // the partly filled byte and the padding
    if (!pack_range(stream, len, 0, dst, rt)) return false;
    *used = packed_bytes + 12 * rt.n;
    *runs_off = packed_bytes;
    *n_runs = (uint32_t)rt.n;
    return true;
}

// The parser of ghip_parse_fasta and the packer in ONE pass: the normalised bytes only ever live in an 8 KiB buffer (L1)
// before they are packed, instead of being written out whole and read back (a tenth of the ingest's CPU time per file,
// which is what bounds files -> clusters once the bases cross PCIe packed).  The run table starts behind room for
// `cap_hint` bases; *fit = false (nothing usable in dst) when the stream outgrows the hint, the table its room, or 4 GB.
int ghip_parse_fasta_packed(const uint8_t *buf, size_t n, const char *path, uint8_t *dst, size_t dst_bytes, size_t cap_hint,
                            size_t *out_len, ghip_genome_stats &st, std::string &err, size_t *used, size_t *runs_off,
                            uint32_t *n_runs, bool *fit) {
    st = ghip_genome_stats();
    *out_len = 0; *used = 0; *runs_off = 0; *n_runs = 0;
    *fit = false;
    const size_t table_at = ((cap_hint + 3) / 4 + 15) / 16 * 16;
    if (cap_hint >= (1ull << 32) || table_at + 12 > dst_bytes) return GHIP_OK;
    RunTable rt{reinterpret_cast<uint32_t *>(dst + table_at), 0, (dst_bytes - table_at) / 12};
    constexpr size_t BLOCK = 8192;                      // packed whenever the buffer holds this much (a multiple of 32)
    uint8_t tmp[BLOCK + 4096 + 64];
    size_t tc = 0, tpos = 0;                            // bytes waiting in tmp; stream position of tmp[0] (a multiple of BLOCK)
    bool ok = true;
    auto drain = [&](bool all) {
        while (ok && (tc >= BLOCK || (all && tc))) {
            const size_t take = tc >= BLOCK ? BLOCK : tc;
            if (tpos + take > cap_hint) { ok = false; break; }
            ok = pack_range(tmp, take, tpos, dst, rt);
            tc -= take; tpos += take;
            if (tc) memmove(tmp, tmp + take, tc);
        }
    };
    std::vector<uint64_t> contig_lengths;
    size_t p = 0;
    while (p < n && (buf[p] == '\n' || buf[p] == '\r')) p++;
    if (p == n) { *fit = true; *runs_off = table_at; return GHIP_OK; }  // empty file: empty stream
    if (buf[p] != '>') { err = std::string("Not a FASTA file (no '>' header): ") + path; return GHIP_EIO; }
    uint64_t amb = 0;
    while (p < n && ok) {
        const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', n - p);  // header line
        p = nl ? (size_t)(nl - buf) + 1 : n;
        uint64_t bases = 0;
        while (p < n && buf[p] != '>' && ok) {  // sequence lines up to the next line that starts with '>'
            nl = (const uint8_t *)memchr(buf + p, '\n', n - p);
            const size_t e = nl ? (size_t)(nl - buf) : n;
            uint64_t cr = 0;
            for (size_t q = p; q < e && ok;) {       // the line in pieces the buffer has room for
                const size_t len = std::min<size_t>(e - q, 4096);
                const size_t crlf = (q + len == e && buf[e - 1] == '\r') ? 1 : 0;   // a CRLF line: the fast path takes the part before the '\r'
                if (len - crlf >= 32 && kHaveAvx2 && line_acgt_avx2(buf + q, len - crlf, tmp + tc)) {
                    tc += len - crlf;
                    cr += crlf;
                } else {
                    uint8_t *o = tmp + tc;
                    size_t w = 0;
                    for (size_t x = q; x < q + len; x++) {
                        const uint8_t c = buf[x];
                        const uint8_t t = kNorm.t[c];
                        o[w] = t;
                        w += (t != 0);
                        amb += (c == 'N') | (c == 'n');
                        cr += (c == '\r');
                    }
                    tc += w;
                }
                q += len;
                drain(false);
            }
            bases += (e - p) - cr;
            p = nl ? e + 1 : n;
        }
        tmp[tc++] = 'N';
        drain(false);
        st.num_contigs++;
        contig_lengths.push_back(bases);
    }
    drain(true);
    if (!ok) return GHIP_OK;   // (*fit stays false: the caller takes the two-step path)
    const size_t m = tpos;
    if (m % 4) { /* the last, partly filled byte is complete: pack_range wrote it with = for its first base */ }
    memset(dst + (m + 3) / 4, 0, table_at - (m + 3) / 4);   // padding up to the table (shipped with it)
    st.num_ambiguous_bases = amb;
    *out_len = m;
    std::sort(contig_lengths.begin(), contig_lengths.end());   // genome_stats.rs:33-45: ascending lengths, first running sum >= total/2
    uint64_t total = 0, run = 0;
    for (uint64_t l : contig_lengths) total += l;
    for (uint64_t l : contig_lengths) {
        run += l;
        if (run >= total / 2) { st.n50 = l; break; }
    }
    *used = table_at + 12 * rt.n;
    *runs_off = table_at;
    *n_runs = (uint32_t)rt.n;
    *fit = true;
    return GHIP_OK;
}

// Reads a whole file (plain or gzip) into `buf`.
bool ghip_slurp(const char *path, std::vector<uint8_t> &buf) { return slurp(path, buf); }

// Upper bound of the stream length of a file without reading it: every header line ('>' ... '\n', >= 2 bytes)
// is replaced by one 'N', so a plain file's stream is no longer than the file; for a single-member gzip the
// trailer holds the uncompressed size (mod 2^32).  0 = unknown (unreadable; the parse reports the error).
uint64_t ghip_stream_capacity_hint(const char *path) {
    {   // a name that does not end in ".gz": the file size bounds the stream -- one system call instead of five.  But a gzip
        // file WITHOUT the extension must not pass for plain text where it matters: the hint also cuts the batches and pieces
        // of ghip_sketch_and_index_files, which would exceed their byte budget by the compression ratio.  Files
        // above 256 KiB therefore get their first two bytes read (gzip: 1f 8b); small files are grouped by size, cost
        // nothing to under-estimate (the ingest notices and takes the two-phase form) and keep the stat-only path -- the
        // 100 000-contig workload makes 100 000 of these calls.
        const size_t l = strlen(path);
        if (!(l > 3 && !strcmp(path + l - 3, ".gz"))) {
            struct stat st;
            if (stat(path, &st) != 0) return 0;
            bool gzip_magic = false;
            if (st.st_size > (256 << 10)) {
                if (FILE *f0 = fopen(path, "rb")) {
                    unsigned char m[2] = {0, 0};
                    gzip_magic = fread(m, 1, 2, f0) == 2 && m[0] == 0x1f && m[1] == 0x8b;
                    fclose(f0);
                }
            }
            if (!gzip_magic) return (uint64_t)st.st_size + 1;
        }
    }
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    unsigned char head[18] = {0};
    size_t got = fread(head, 1, 18, f);
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    uint64_t cap = size > 0 ? (uint64_t)size : 0;
    if (got >= 2 && head[0] == 0x1f && head[1] == 0x8b && size >= 4) {
        unsigned char t[4];
        fseek(f, -4, SEEK_END);
        if (fread(t, 1, 4, f) == 4) cap = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
        // BGZF (bgzip: what an indexed compressed FASTA is written with) is one gzip member per 64 KiB of text, each naming its own
        // size in an extra field ('B' 'C', SLEN 2, BSIZE = member bytes - 1): the trailer of the LAST member alone says nothing
        // about the file (the end-of-file marker's ISIZE is 0), and a stream that outgrows its hint sends the whole call through
        // the two-phase form.  Walking the members -- an 18-byte read and a 4-byte read each -- gives the exact size.
        if (got == 18 && head[2] == 8 && (head[3] & 4) && head[12] == 'B' && head[13] == 'C' && head[14] == 2 && head[15] == 0 && (head[10] | (head[11] << 8)) == 6) {
            uint64_t total = 0;
            long pos = 0;
            bool ok = true;
            while (ok && pos < size) {
                unsigned char h[18];
                ok = fseek(f, pos, SEEK_SET) == 0 && fread(h, 1, 18, f) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) &&
                     (h[10] | (h[11] << 8)) == 6 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
                if (!ok) break;
                const long member = (long)(h[16] | (h[17] << 8)) + 1;
                ok = member >= 26 && pos + member <= size && fseek(f, pos + member - 4, SEEK_SET) == 0 && fread(t, 1, 4, f) == 4;
                if (!ok) break;
                total += (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
                pos += member;
            }
            if (ok && pos == size) cap = total;   // (anything else: the plain gzip guess above stands)
        }
    }
    fclose(f);
    return cap + 1;  // a file without a final newline still gets its trailing 'N'
}

int ghip_read_fasta_stream(const char *path, std::vector<uint8_t> &out, ghip_genome_stats &st, std::string &err) {
    std::vector<uint8_t> buf;
    if (!slurp(path, buf)) { err = std::string("Failed to open fasta file ") + path; return GHIP_EIO; }
    out.resize(buf.size() + 16);
    size_t len = 0;
    int rc = ghip_parse_fasta(buf.data(), buf.size(), path, out.data(), out.size(), &len, st, err);
    if (rc != GHIP_OK) return rc;
    out.resize(len);  // len <= file size + 1 always fits
    return GHIP_OK;
}

int ghip_read_fasta_streams(const char *const *paths, size_t n, int threads,
                            std::vector<std::vector<uint8_t>> &streams, std::vector<ghip_genome_stats> &stats,
                            std::string &err) {
    streams.assign(n, {});
    stats.assign(n, ghip_genome_stats());
    if (threads < 1) threads = 1;
    threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(n, 1));
    std::atomic<size_t> next{0};
    std::atomic<int> rc{GHIP_OK};
    std::mutex emu;
    auto worker = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n) return;
            std::string e;
            int r = ghip_read_fasta_stream(paths[i], streams[i], stats[i], e);
            if (r != GHIP_OK) {
                std::lock_guard<std::mutex> lk(emu);
                if (rc.load() == GHIP_OK) { rc = r; err = e; }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto &th : pool) th.join();
    return rc.load();
}
