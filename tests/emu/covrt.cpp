// TEST INFRASTRUCTURE -- which lines of the KERNEL files does the emulated suite execute?  (make SAN=cov)
//
// The image has clang's coverage instrumentation but neither llvm-cov nor llvm-profdata, so this is the small runtime behind
// -fsanitize-coverage=trace-pc-guard,pc-table: the compiler gives every edge of the kernel files a guard and lists every edge's
// code address; a guard's first hit marks its edge; at exit (and on hipemu_cov_dump) the list goes to HIPEMU_COV_OUT.<pid> as
// `library+offset hit` lines.  scripts/emu_coverage.py merges the files of a run's processes, symbolizes the addresses
// (llvm-symbolizer) and reports, per kernel file, the source lines no test reached.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace {
struct Module { uint32_t *g0, *g1; const uintptr_t *p0, *p1; };
// (plain data, no constructors: the compiler's module constructors call in here before this file's own would have run)
Module g_mods[64];
int g_nmods = 0;
uint8_t *g_hit = nullptr;   // by guard number (1-based)
size_t g_hit_size = 0;
std::mutex g_mu;            // (constexpr constructor)
uint32_t g_next = 0;

void dump() {
    const char *path = getenv("HIPEMU_COV_OUT");
    if (!path || !*path) return;
    char name[4096];
    snprintf(name, sizeof name, "%s.%d", path, (int)getpid());
    FILE *f = fopen(name, "w");
    if (!f) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (getenv("HIPEMU_COV_DEBUG")) for (int x = 0; x < g_nmods; x++) { const Module &m = g_mods[x]; fprintf(stderr, "[cov] module guards %zu pcs %zu\n", (size_t)(m.g1 - m.g0), m.p0 ? (size_t)(m.p1 - m.p0) / 2 : 0); }
    for (int x = 0; x < g_nmods; x++) {
        const Module &m = g_mods[x];
        const size_t n = (size_t)(m.g1 - m.g0);
        if (!m.p0 || (size_t)(m.p1 - m.p0) / 2 != n) continue;
        for (size_t i = 0; i < n; i++) {
            const uintptr_t pc = m.p0[2 * i];
            Dl_info di;
            if (!dladdr((const void *)pc, &di) || !di.dli_fname) continue;
            const uint32_t id = m.g0[i];   // the number the guard was given (0: never initialised)
            fprintf(f, "%s+0x%zx %d\n", di.dli_fname, (size_t)(pc - (uintptr_t)di.dli_fbase), id && id < g_hit_size ? (int)g_hit[id] : 0);
        }
    }
    fclose(f);
}
}  // namespace

extern "C" {
void __sanitizer_cov_trace_pc_guard_init(uint32_t *start, uint32_t *stop) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (start == stop || *start) return;
    for (uint32_t *g = start; g < stop; g++) *g = ++g_next;
    {
        const size_t want = (size_t)g_next + 1;
        uint8_t *h = (uint8_t *)calloc(want, 1);
        if (g_hit) { for (size_t i = 0; i < g_hit_size; i++) h[i] = g_hit[i]; }   // (the old table is left to leak: a hit may be landing in it)
        g_hit = h; g_hit_size = want;
    }
    for (int x = 0; x < g_nmods; x++) {
        Module &m = g_mods[x];
        if (!m.g0 && m.p0 && (size_t)(m.p1 - m.p0) / 2 == (size_t)(stop - start)) { m.g0 = start; m.g1 = stop; return; }
    }
    if (g_nmods < 64) g_mods[g_nmods++] = Module{start, stop, nullptr, nullptr};
    static bool once = (atexit(dump), true);
    (void)once;
}
void __sanitizer_cov_pcs_init(const uintptr_t *beg, const uintptr_t *end) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int x = 0; x < g_nmods; x++) {
        Module &m = g_mods[x];
        if (m.p0 == beg) return;
        if (!m.p0 && (size_t)(end - beg) / 2 == (size_t)(m.g1 - m.g0)) { m.p0 = beg; m.p1 = end; return; }
    }
    if (g_nmods < 64) g_mods[g_nmods++] = Module{nullptr, nullptr, beg, end};
}
void __sanitizer_cov_trace_pc_guard(uint32_t *guard) {
    const uint32_t id = *guard;
    if (id && id < g_hit_size) g_hit[id] = 1;   // (a racy byte store of 1: fine)
}
void hipemu_cov_dump() { dump(); }
}
