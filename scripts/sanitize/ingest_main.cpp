#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <cstdint>
struct ghip_genome_stats { uint64_t num_contigs = 0, num_ambiguous_bases = 0, n50 = 0; };
int ghip_read_fasta_stream(const char *path, std::vector<uint8_t> &out, ghip_genome_stats &st, std::string &err);
uint64_t ghip_stream_capacity_hint(const char *path);
bool ghip_slurp(const char *path, std::vector<uint8_t> &buf);
bool ghip_pack_stream(const uint8_t *stream, size_t len, uint8_t *dst, size_t dst_bytes, size_t *used, size_t *runs_off, uint32_t *n_runs);
int ghip_parse_fasta_packed(const uint8_t *buf, size_t n, const char *path, uint8_t *dst, size_t dst_bytes, size_t cap_hint,
                            size_t *out_len, ghip_genome_stats &st, std::string &err, size_t *used, size_t *runs_off,
                            uint32_t *n_runs, bool *fit);

// what sketch.hip's unpack_bases_kernel + patch_runs_kernel do, on the host
static std::vector<uint8_t> unpack(const uint8_t *packed, const uint32_t *runs, uint32_t n_runs, size_t len) {
    std::vector<uint8_t> out(len);
    for (size_t i = 0; i < len; i++) out[i] = "ACGT"[(packed[i / 4] >> (2 * (i & 3))) & 3];
    for (uint32_t r = 0; r < n_runs; r++)
        for (uint32_t k = 0; k < runs[3 * r + 1]; k++) out[(size_t)runs[3 * r] + k] = (uint8_t)runs[3 * r + 2];
    return out;
}

int main(int argc, char **argv) {
    int bad = 0;
    for (int i = 1; i < argc; i++) {
        std::vector<uint8_t> out; ghip_genome_stats st; std::string err;
        int rc = ghip_read_fasta_stream(argv[i], out, st, err);
        const uint64_t cap = ghip_stream_capacity_hint(argv[i]);
        printf("%s rc=%d len=%zu cap=%llu contigs=%llu %s\n", argv[i], rc, out.size(), (unsigned long long)cap, (unsigned long long)st.num_contigs, err.c_str());
        if (rc != 0) continue;
        // the two packed forms against the plain stream: pack the finished stream; parse and pack in one pass
        for (size_t table_room : {(size_t)12, (size_t)65536}) {
            std::vector<uint8_t> dst((out.size() + 3) / 4 + 64 + table_room + cap / 4);
            size_t used = 0, off = 0; uint32_t nr = 0;
            if (ghip_pack_stream(out.data(), out.size(), dst.data(), dst.size(), &used, &off, &nr)) {
                if (unpack(dst.data(), reinterpret_cast<const uint32_t *>(dst.data() + off), nr, out.size()) != out) { printf("BAD pack_stream %s\n", argv[i]); bad++; }
            }
            std::vector<uint8_t> raw;
            if (!ghip_slurp(argv[i], raw)) { printf("BAD slurp\n"); bad++; continue; }
            for (size_t hint : {(size_t)cap, out.size(), out.size() ? out.size() - 1 : (size_t)0}) {
                std::vector<uint8_t> d2((hint + 3) / 4 + 64 + table_room);
                size_t len = 0, used2 = 0, off2 = 0; uint32_t nr2 = 0; bool fit = false; ghip_genome_stats st2; std::string e2;
                int r2 = ghip_parse_fasta_packed(raw.data(), raw.size(), argv[i], d2.data(), d2.size(), hint, &len, st2, e2, &used2, &off2, &nr2, &fit);
                if (r2 != 0) { printf("BAD parse_packed rc %s\n", argv[i]); bad++; continue; }
                if (hint < out.size()) { if (fit) { printf("BAD fit with a short hint %s\n", argv[i]); bad++; } continue; }
                if (!fit) continue;   // the run table had no room: allowed
                if (len != out.size() || used2 > d2.size() || st2.num_contigs != st.num_contigs || st2.num_ambiguous_bases != st.num_ambiguous_bases || st2.n50 != st.n50 ||
                    unpack(d2.data(), reinterpret_cast<const uint32_t *>(d2.data() + off2), nr2, len) != out) { printf("BAD parse_packed %s\n", argv[i]); bad++; }
            }
        }
    }
    if (bad) { printf("%d packed-form mismatches\n", bad); return 1; }
    printf("packed forms ok\n");
}
