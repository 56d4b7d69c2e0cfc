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

// Returns 0 ok, GHIP_EIO on unreadable / non-FASTA input.  The same pass yields the assembly
// statistics galah computes in a second read of every file (reference src/genome_stats.rs:11-51):
// records, raw 'N'/'n' count, N50 over record lengths (bases, line ends excluded).
int ghip_read_fasta_stream(const char *path, std::vector<uint8_t> &out, ghip_genome_stats &st, std::string &err) {
    st = ghip_genome_stats();
    std::vector<uint64_t> contig_lengths;
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
        uint64_t bases = 0;
        while (p < n) {
            const uint8_t c = buf[p];
            if (line_start && c == '>') break;
            line_start = (c == '\n');
            if (c != '\n' && c != '\r') bases++;
            if (c == 'N' || c == 'n') st.num_ambiguous_bases++;
            const uint8_t o = kNorm.t[c];
            if (o) out.push_back(o);
            p++;
        }
        out.push_back('N');
        st.num_contigs++;
        contig_lengths.push_back(bases);
    }
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
