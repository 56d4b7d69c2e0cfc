#include <cstdio>
#include <string>
#include <vector>
#include <cstdint>
struct ghip_genome_stats { uint64_t num_contigs = 0, num_ambiguous_bases = 0, n50 = 0; };
int ghip_read_fasta_stream(const char *path, std::vector<uint8_t> &out, ghip_genome_stats &st, std::string &err);
uint64_t ghip_stream_capacity_hint(const char *path);
int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++) {
        std::vector<uint8_t> out; ghip_genome_stats st; std::string err;
        int rc = ghip_read_fasta_stream(argv[i], out, st, err);
        printf("%s rc=%d len=%zu cap=%llu contigs=%llu %s\n", argv[i], rc, out.size(), (unsigned long long)ghip_stream_capacity_hint(argv[i]), (unsigned long long)st.num_contigs, err.c_str());
    }
}
