// Host-side FASTA ingest: file -> device-format base stream (see galah_hip.h "genome ingest").
//
// Mirrors what finch::sketch_files does before hashing (reference src/finch.rs:69):
// needletail parse_fastx_file (plain or gzip, '>' records, multi-line sequences) and
// Sequence::normalize(iupac=false).  One stream per file: every record's normalised bytes
// followed by a single 'N' so that no k-mer spans two records.
#include <zlib.h>

#include <atomic>
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

bool slurp(const char *path, std::vector<uint8_t> &buf) {
    gzFile f = gzopen(path, "rb");  // transparently reads plain files too
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    size_t n = 0;
    buf.resize(1 << 22);
    for (;;) {
        if (n == buf.size()) buf.resize(buf.size() * 2);
        size_t want = std::min<size_t>(buf.size() - n, 1u << 30);
        int r = gzread(f, buf.data() + n, (unsigned)want);
        if (r < 0) { gzclose(f); return false; }
        if (r == 0) break;
        n += (size_t)r;
    }
    gzclose(f);
    buf.resize(n);
    return true;
}

}  // namespace

// Returns 0 ok, GHIP_EIO on unreadable / non-FASTA input.
int ghip_read_fasta_stream(const char *path, std::vector<uint8_t> &out, std::string &err) {
    std::vector<uint8_t> buf;
    if (!slurp(path, buf)) { err = std::string("Failed to open fasta file ") + path; return GHIP_EIO; }
    const size_t n = buf.size();
    out.clear();
    out.reserve(n + 16);
    size_t p = 0;
    while (p < n && (buf[p] == '\n' || buf[p] == '\r')) p++;
    if (p == n) return GHIP_OK;  // empty file: empty stream, empty sketch
    if (buf[p] != '>') { err = std::string("Not a FASTA file (no '>' header): ") + path; return GHIP_EIO; }
    while (p < n) {
        while (p < n && buf[p] != '\n') p++;  // header line
        if (p < n) p++;
        bool line_start = true;
        while (p < n) {
            const uint8_t c = buf[p];
            if (line_start && c == '>') break;
            line_start = (c == '\n');
            const uint8_t o = kNorm.t[c];
            if (o) out.push_back(o);
            p++;
        }
        out.push_back('N');
    }
    return GHIP_OK;
}

int ghip_read_fasta_streams(const char *const *paths, size_t n, int threads,
                            std::vector<std::vector<uint8_t>> &streams, std::string &err) {
    streams.assign(n, {});
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
            int r = ghip_read_fasta_stream(paths[i], streams[i], e);
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
