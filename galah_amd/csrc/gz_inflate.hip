// gzip FASTA files inflated, checked, parsed and packed ON THE DEVICE (ghip_options.gz_device; host driver: ingest_gz.cpp).
//
// What it replaces: the host's libdeflate / zlib inflate + ghip_parse_fasta_packed of ingest.cpp for files that arrive as
// .gz -- the form real collections ship in.  Host inflate is what bounds files -> clusters for gzip input (0.50 s per 1 000
// 5 Mb genomes on the 16-CPU quota of the GPU boxes against 0.076 s for plain files, profiles/r04a_bench.json).  Reference
// behaviour: needletail auto-detects gzip behind finch::sketch_files (reference src/finch.rs:69); the reference's gz test is
// tests/test_cmdline.rs:612-629.
//
// DEFLATE is serial within a member, so the parallelism is ACROSS files: one wavefront per file, thousands of files per
// launch.  Inside a wavefront:
//   * the Huffman decoding is one dependent chain and runs on the SCALAR unit: every value of the chain is wave-uniform
//     (v_readfirstlane behind each LDS look-up), so the compiler keeps bit buffer, bit count and table entries in SGPRs;
//     the input reaches it through a 2 x 64-dword window the 64 lanes hold in two VGPRs (one coalesced load per 256 bytes,
//     v_readlane per dword).  Every scalar instruction is a whole issue slot of the wavefront, so the chain only looks the
//     codes up and passes over them: the table entries and the stream's bits behind each code go to the lanes;
//   * the decode tables (10-bit litlen + 8-bit distance primaries with sub-tables, 7 KiB) live in LDS and are built by the
//     64 lanes together; with the copy stage 9.9 KiB of LDS per wavefront = 16 wavefronts per CU;
//   * the LZ77 copies are the data-parallel half: 64 tokens at a time, one per lane, which finishes the decoding (base +
//     extra bits), finds its place by a wave prefix sum and writes its literal or copies its match.  A match can name bytes
//     of its own batch: it waits for exactly the tokens that write its source (a mask of lanes), in rounds -- two on average
//     for a gzip -6 genome, five at gzip -1 -- and the rounds do not go through memory: the batch's bytes are also kept in a
//     2 KiB stage in LDS, so only sources older than the batch are read from the text (HBM / L2: a 32 KiB window per
//     wavefront in LDS would leave 4 wavefronts per CU), once, and those bytes were stored a whole decode phase ago.
// Anything this path does not take -- further members, FHCRC, an incomplete code, a text that does not start with '>' ... --
// sets a status and the host path (ingest.cpp) ingests that file instead and alone decides what is an error.
//
// Behind the inflate: gz_crc_kernel (CRC-32 of the text by spans, combined in GF(2): gz_common.h), the FASTA pass
// (fasta_chunk / fasta_scan / fasta_emit: the parse of ghip_parse_fasta as two sweeps around a scan over 16 KiB chunks,
// yielding the genome in its resident 2-bit + validity form, the record table and the assembly statistics of reference
// src/genome_stats.rs:11-51).  No host round trip between the five launches.
#include <hip/hip_runtime.h>

#include "ghip_internal.h"
#include "gz_common.h"

namespace {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
__device__ __forceinline__ void wave_sync() {   // LDS written by some lanes is read by others of the same wavefront
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Text written by some lanes is read back by others of the same wavefront (the history of a match).  The L1 is shared by the
// work-items of a workgroup and a store waits in vmcnt: workgroup scope is what orders a store in front of a later load
// here (s_waitcnt vmcnt(0); no cache invalidate).
__device__ __forceinline__ void text_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// ------------------------------------------------------------------------------------------------ decode tables
constexpr uint32_t LL_P = 10, D_P = 8, PRE_P = 7;          // primary index bits
constexpr uint32_t LL_SYMS = 288, D_SYMS = 32, PRE_SYMS = 19;
constexpr uint32_t LL_ROOM = 1334, D_ROOM = 402;           // primary + every sub-table a complete code can need (15-bit codes)
constexpr uint32_t LENS_LL = 0, LENS_D = LL_SYMS, LENS_PRE = LL_SYMS + D_SYMS;

// entry: bits 0-3 code bits to consume, 4-7 what it is (one flag; none = not a code), 8-11 extra bits (F_SUB: index bits of the
// sub-table), 16-31 value (literal / base length / base distance / first entry of the sub-table)
constexpr uint32_t K_BAD = 0, K_LIT = 1u << 4, K_BASE = 1u << 5, K_SUB = 1u << 6, K_EOB = 1u << 7;
__device__ __forceinline__ uint32_t entry(uint32_t kind, uint32_t nbits, uint32_t value, uint32_t extra) { return nbits | kind | (extra << 8) | (value << 16); }
enum { T_PRE = 0, T_LL = 1, T_D = 2 };
template <int T> __device__ __forceinline__ uint32_t symbol_entry(uint32_t sym, uint32_t nbits) {
    if (T == T_PRE) return entry(K_LIT, nbits, sym, 0);
    if (T == T_LL) {
        if (sym < 256) return entry(K_LIT, nbits, sym, 0);
        if (sym == 256) return entry(K_EOB, nbits, 0, 0);
        if (sym > 285) return entry(K_BAD, nbits, 0, 0);
        const uint32_t i = sym - 257;   // RFC 1951 3.2.5: 257-264 lengths 3-10, then four codes per extra bit, 285 = 258
        if (i < 8) return entry(K_BASE, nbits, 3 + i, 0);
        if (i == 28) return entry(K_BASE, nbits, 258, 0);
        const uint32_t e = (i >> 2) - 1;
        return entry(K_BASE, nbits, 3 + ((4 + (i & 3)) << e), e);
    }
    if (sym > 29) return entry(K_BAD, nbits, 0, 0);
    if (sym < 4) return entry(K_BASE, nbits, 1 + sym, 0);
    const uint32_t e = (sym >> 1) - 1;   // distances: two codes per extra bit
    return entry(K_BASE, nbits, 1 + ((2 + (sym & 1)) << e), e);
}

constexpr uint32_t STAGE_BYTES = 2048;   // (a token writes at most 258 bytes; a batch of a genome ~300)
struct InflateLds {
    uint32_t ll[LL_ROOM];
    uint32_t dt[D_ROOM];                 // (the precode's 128 entries live here while the code lengths are read)
    uint32_t count[16], first[16], offs[16];
    uint32_t sub_next, err;
    uint8_t stage[STAGE_BYTES];          // the bytes of the token group being written (emit_tokens): a source inside the group is read here
    uint16_t sorted[LL_SYMS];            // symbols by (code length, symbol)
    uint8_t lens[LL_SYMS + D_SYMS + 32]; // code lengths: litlen, distance, precode
};

// Canonical Huffman code of lens[at .. at + nsyms) -> decode table with a 2^P primary.  Built by index: entry x of the
// primary is the symbol whose bit-reversed code is a prefix of x (canonical codes of one length are consecutive, so
// "is x's L-bit prefix a code" is one subtraction), or a pointer to a sub-table as wide as the longest code below that
// prefix.  Returns GHIP_GZ_OK / EDATA (over-subscribed) / EUNUSUAL (incomplete, or out of room).
template <int T> __device__ uint32_t build_table(InflateLds &L, uint32_t at, uint32_t nsyms, uint32_t P, uint32_t *tab, uint32_t room, uint32_t lane) {
    if (lane < 16) L.count[lane] = 0;
    if (lane == 0) { L.sub_next = 1u << P; L.err = 0; }
    wave_sync();
    for (uint32_t s = lane; s < nsyms; s += 64) {
        const uint32_t l = L.lens[at + s];
        if (l) atomicAdd(&L.count[l], 1u);
    }
    wave_sync();
    uint32_t code = 0, off = 0, kraft = 0, used = 0;
    for (uint32_t l = 1; l < 16; l++) {
        const uint32_t c = uni(L.count[l]);
        if (lane == 0) { L.first[l] = code; L.offs[l] = off; }
        code = (code + c) << 1;
        off += c;
        used += c;
        kraft += c << (15 - l);
    }
    if (kraft > (1u << 15)) return GHIP_GZ_EDATA;
    if (kraft < (1u << 15)) {
        // an incomplete code: legal for distances when no or one code is in use (RFC 1951 3.2.7), otherwise the host's call
        const bool fine = T == T_D && (used == 0 || (used == 1 && uni(L.count[1]) == 1));
        if (!fine) return GHIP_GZ_EUNUSUAL;
    }
    wave_sync();
    // symbols in code order: within a length by symbol, found with one ballot per (64 symbols, length in use)
    for (uint32_t l = 1; l < 16; l++) {
        if (uni(L.count[l]) == 0) continue;
        uint32_t run = uni(L.offs[l]);
        for (uint32_t s0 = 0; s0 < nsyms; s0 += 64) {
            const uint32_t s = s0 + lane;
            const bool mine = s < nsyms && L.lens[at + s] == l;
            const uint64_t m = __ballot(mine);
            if (mine) L.sorted[run + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)s;
            run += (uint32_t)__popcll(m);
        }
    }
    wave_sync();
    for (uint32_t x = lane; x < (1u << P); x += 64) {
        const uint32_t rev = __brev(x);       // bit 31 = the first bit of the stream
        uint32_t e = entry(K_BAD, 0, 0, 0);
        bool found = false;
        for (uint32_t l = 1; l <= P && !found; l++) {
            const uint32_t rel = (rev >> (32 - l)) - L.first[l];
            if (rel < L.count[l]) { e = symbol_entry<T>(L.sorted[L.offs[l] + rel], l); found = true; }
        }
        if (!found) {
            const uint32_t p = rev >> (32 - P);   // the P-bit prefix in code order
            uint32_t longest = 0;
            for (uint32_t l = P + 1; l < 16; l++) {
                const uint32_t lo = p << (l - P), hi = lo + (1u << (l - P)), f = L.first[l], c = L.count[l];
                if (c && lo < f + c && hi > f) longest = l;
            }
            if (longest) {
                const uint32_t sb = longest - P, base = atomicAdd(&L.sub_next, 1u << sb);
                if (base + (1u << sb) > room) L.err = 1;
                else {
                    e = entry(K_SUB, P, base, sb);
                    for (uint32_t j = 0; j < (1u << sb); j++) {
                        const uint32_t rj = __brev(j);
                        uint32_t se = entry(K_BAD, 0, 0, 0);
                        for (uint32_t l = P + 1; l <= longest; l++) {
                            const uint32_t rel = ((p << (l - P)) | (rj >> (32 - (l - P)))) - L.first[l];
                            if (rel < L.count[l]) { se = symbol_entry<T>(L.sorted[L.offs[l] + rel], l - P); break; }
                        }
                        tab[base + j] = se;
                    }
                }
            }
        }
        tab[x] = e;
    }
    wave_sync();
    return uni(L.err) ? GHIP_GZ_EUNUSUAL : GHIP_GZ_OK;
}

// ------------------------------------------------------------------------------------------------ bit input
struct Reader {
    const uint32_t *w;     // the image as dwords (its area is 16-byte aligned and padded)
    uint32_t n_words;      // dwords that may be loaded; reads behind them give 0
    uint32_t cur, nxt;     // LANE-PRIVATE: dword (wpos & ~63) + lane of the image, and the same of the next 64
    uint32_t wpos;         // next dword to take (uniform, like everything below)
    uint64_t bb;           // bit buffer, the stream's next bit in bit 0
    uint32_t bl;           // bits in bb
};
__device__ __forceinline__ uint32_t rd_load(const Reader &r, uint32_t idx) { return idx < r.n_words ? r.w[idx] : 0u; }
__device__ __forceinline__ uint32_t rd_take(Reader &r, uint32_t lane) {
    const uint32_t w = __builtin_amdgcn_readlane(r.cur, r.wpos & 63u);
    r.wpos++;
    if ((r.wpos & 63u) == 0) { r.cur = r.nxt; r.nxt = rd_load(r, r.wpos + 64 + lane); }
    return w;
}
__device__ __forceinline__ void rd_seek(Reader &r, uint32_t byte_pos, uint32_t lane) {
    r.wpos = byte_pos >> 2;
    const uint32_t base = r.wpos & ~63u;
    r.cur = rd_load(r, base + lane);
    r.nxt = rd_load(r, base + 64 + lane);
    const uint32_t skip = 8 * (byte_pos & 3u);
    r.bb = rd_take(r, lane) >> skip;
    r.bl = 32 - skip;
}
// after this at least 33 bits are in the buffer: a litlen code with its extra bits (20), a distance code with its (28)
__device__ __forceinline__ void rd_fill(Reader &r, uint32_t lane) {
    if (r.bl <= 32) { r.bb |= (uint64_t)rd_take(r, lane) << r.bl; r.bl += 32; }
}
__device__ __forceinline__ uint32_t rd_bits(Reader &r, uint32_t n) {   // n <= 16
    const uint32_t v = (uint32_t)r.bb & ((1u << n) - 1u);
    r.bb >>= n;
    r.bl -= n;
    return v;
}
__device__ __forceinline__ uint32_t rd_byte_pos(const Reader &r) { return r.wpos * 4 - (r.bl >> 3); }   // of the next whole byte (bl a multiple of 8)

// ------------------------------------------------------------------------------------------------ the data-parallel half
typedef uint64_t __attribute__((aligned(1))) u64_any;   // global loads / stores at any byte address (one instruction on gfx9+)
typedef uint32_t __attribute__((aligned(1))) u32_any;

// text[p, p + len) = the len bytes that start dist in front of p, LZ77's way (a distance shorter than the length repeats) --
// and the same bytes into the group's stage (stage[0] = text[group0]), where the later tokens of the group find them.  A
// source older than the group comes from the text in whole words; one that reaches into the group (or repeats) byte by byte,
// each byte from where it lives: the text in front of group0, the stage from there on.
__device__ __forceinline__ void stage_word(uint8_t *at, uint64_t v, uint32_t n) {
    for (uint32_t k = 0; k < n; k++) at[k] = (uint8_t)(v >> (8 * k));
}
__device__ __forceinline__ void copy_match(uint8_t *text, uint8_t *stage, uint32_t group0, uint32_t p, uint32_t len, uint32_t dist) {
    uint8_t *d = text + p, *g = stage + (p - group0);
    const uint8_t *s = d - dist;
    if (dist >= len && p - dist + len <= group0) {   // apart, and older than the group: whole words, the last one overlapping the one before
        if (len >= 8) {
            for (uint32_t i = 0; i + 8 <= len; i += 8) { const uint64_t v = *(const u64_any *)(s + i); *(u64_any *)(d + i) = v; stage_word(g + i, v, 8); }
            if (len & 7u) { const uint64_t v = *(const u64_any *)(s + len - 8); *(u64_any *)(d + len - 8) = v; stage_word(g + len - 8, v, 8); }
        } else if (len >= 4) {
            const uint32_t a = *(const u32_any *)s, b = *(const u32_any *)(s + len - 4);
            *(u32_any *)d = a;
            *(u32_any *)(d + len - 4) = b;
            stage_word(g, a, 4);
            stage_word(g + len - 4, b, 4);
        } else {
            const uint8_t a = s[0], b = s[1], c = s[2];   // len == 3
            d[0] = a; d[1] = b; d[2] = c;
            g[0] = a; g[1] = b; g[2] = c;
        }
    } else {
        const uint32_t src = p - dist;
        for (uint32_t i = 0, j = 0; i < len; i++) {
            const uint32_t from = src + j;
            const uint8_t v = from < group0 ? text[from] : stage[from - group0];
            d[i] = v;
            g[i] = v;
            if (++j == dist) j = 0;
        }
    }
}

// The batch of ntok tokens written to text[pos0 ..]: lane t holds token t (tk: decode_batch).  The lanes finish what the chain
// left undone -- lengths and distances from their base values and extra bits, the tokens' places, the checks that need them --
// and copy.  Returns GHIP_GZ_OK and the bytes written, or what is wrong (nothing written then).
__device__ uint32_t emit_tokens(uint8_t *text, uint8_t *stage, const uint4 tk, uint32_t ntok, uint32_t pos0, uint32_t room, uint32_t lane, uint32_t &written,
                                uint32_t &n_matches, uint32_t &n_rounds) {
    const bool active = lane < ntok;
    const bool is_match = active && (tk.x & K_BASE);
    const uint32_t len = !active ? 0u : is_match ? (tk.x >> 16) + (tk.y & ((1u << ((tk.x >> 8) & 15u)) - 1u)) : 1u;
    const uint32_t dist = (tk.z >> 16) + (tk.w & ((1u << ((tk.z >> 8) & 15u)) - 1u));
    uint32_t x = len;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    const uint32_t total = __builtin_amdgcn_readlane(x, 63);
    const uint32_t p = pos0 + x - len;
    written = 0;
    if (total > room) return GHIP_GZ_EOVERFLOW;                        // more text than the trailer promised
    if (__ballot(is_match && dist > p)) return GHIP_GZ_EDATA;          // a distance that reaches in front of the member's first byte
    // Which tokens of the batch a match waits for: those that write a byte of its source [src, src_end) -- positions and ends
    // grow with the lane number, so both edges of that run of tokens come from a binary search over the lanes (6 shuffles
    // each).  A source that ends in front of the batch waits for nothing.  (The first version let a match go only when its
    // source ended in front of the first unfinished match: 2.1 rounds per batch on a gzip -6 genome, but 8.8 on gzip -1,
    // whose matches all name the most recent occurrence -- a chain through the batch; this rule: 1.9 and 5.4.)
    const uint32_t src = p - dist, src_end = src + (len < dist ? len : dist), end = p + len;
    uint32_t lo = 0, hi1 = 0;   // tokens [0, lo) end at or in front of src; tokens [0, hi1) start in front of src_end
#pragma unroll
    for (uint32_t s = 32; s >= 1; s >>= 1) {
        const uint32_t e = __shfl(end, (int)(lo + s - 1)), q = __shfl(p, (int)(hi1 + s - 1));
        if (e <= src) lo += s;
        if (q < src_end) hi1 += s;
    }
    uint64_t wait_for = 0;
    if (is_match && src_end > pos0) {
        if (hi1 > lane) hi1 = lane;   // (only tokens in front of this one: its own bytes are not its source)
        wait_for = hi1 > lo ? ((1ull << hi1) - 1ull) & ~((1ull << lo) - 1ull) : 0ull;
    }
    uint64_t done = ~__ballot(active);   // lanes without a token have nothing to do
    n_matches += (uint32_t)__popcll(__ballot(is_match));
    // The tokens go in GROUPS of at most STAGE_BYTES of text (a genome's batch is one group), each written to the text AND to
    // the stage in LDS: the dependency depth of a batch is what it is (5 rounds at gzip -1), but a round whose sources sit in
    // LDS costs a wave barrier, not a drained store queue and a trip to L2.  Only the group's first step waits for the text
    // (the stores of the group before: issued a whole decode phase ago).
    uint32_t emitted = 0;
    for (uint32_t group = 0; emitted < ntok; group++) {
        if (group == 64) return GHIP_GZ_EDATA;   // (cannot happen: a group takes at least one token -- but no loop of this kernel is left unbounded)
        const uint32_t group0 = __builtin_amdgcn_readlane(p, emitted);
        const bool mine = active && lane >= emitted && end - group0 <= STAGE_BYTES;
        const uint64_t members = __ballot(mine);
        emitted += (uint32_t)__popcll(members);
        text_sync();       // what the group before stored (a whole decode phase ago, as a rule) is in the text
        if (mine && !is_match) { text[p] = (uint8_t)(tk.x >> 16); stage[p - group0] = (uint8_t)(tk.x >> 16); }
        done |= members & ~__ballot(is_match);   // (the group's literals)
        for (uint32_t round = 0; members & ~done; round++) {
            if (round == 64) return GHIP_GZ_EDATA;   // (cannot happen: the first unfinished match waits for finished tokens only)
            n_rounds++;
            wave_sync();   // the stage as the round before left it
            const bool ready = mine && is_match && !((done >> lane) & 1ull) && !(wait_for & ~done);
            if (ready) copy_match(text, stage, group0, p, len, dist);
            done |= __ballot(ready);
        }
        wave_sync();       // (the next group writes the stage afresh)
    }
    written = total;       // (no wait for these stores here: the next group's first step, or the kernel's end, sees to them)
    return GHIP_GZ_OK;
}

__device__ __forceinline__ uint32_t ld_byte(const uint8_t *in, uint32_t i) { return uni((uint32_t)in[i]); }

// The serial half: up to 64 symbols of the current block off the bit stream, symbol t into LANE t's registers (a compare of
// the lane number with the token count and one v_cndmask per word: vector instructions, whose issue slots are idle here -- no
// LDS, nothing for the scalar unit).  The chain does the least it can per symbol -- look the code up, pass over it and its
// extra bits, hand the entry and the stream's bits behind the code to the lane -- since every scalar instruction is a whole
// issue slot of the wavefront: lengths, distances, positions and the checks on them are the lanes' work (emit_tokens).
// tk = (a literal's entry) or (length entry, bits behind the length code, distance entry, bits behind the distance code).
enum : uint32_t { BATCH_FULL = 0, BATCH_END_OF_BLOCK = 1, BATCH_DAMAGED = 2 };
__device__ __forceinline__ uint32_t decode_batch(InflateLds &L, Reader &r, uint32_t lane, uint32_t &ntok_out, uint4 &tk) {
    uint32_t ntok = 0;
    do {
        rd_fill(r, lane);
        uint32_t e = uni(L.ll[(uint32_t)r.bb & ((1u << LL_P) - 1u)]);
        if (e & K_SUB) {
            r.bb >>= LL_P;
            r.bl -= LL_P;
            e = uni(L.ll[(e >> 16) + ((uint32_t)r.bb & ((1u << ((e >> 8) & 15u)) - 1u))]);
        }
        r.bb >>= e & 15u;
        const bool mine = lane == ntok;
        tk.x = mine ? e : tk.x;
        if (e & K_LIT) {
            r.bl -= e & 15u;
            ntok++;
            continue;
        }
        const uint32_t xl = (e >> 8) & 15u;
        tk.y = mine ? (uint32_t)r.bb : tk.y;
        r.bb >>= xl;
        r.bl -= (e & 15u) + xl;
        if (!(e & K_BASE)) { ntok_out = ntok; return (e & K_EOB) ? BATCH_END_OF_BLOCK : BATCH_DAMAGED; }
        rd_fill(r, lane);
        uint32_t d = uni(L.dt[(uint32_t)r.bb & ((1u << D_P) - 1u)]);
        if (d & K_SUB) {
            r.bb >>= D_P;
            r.bl -= D_P;
            d = uni(L.dt[(d >> 16) + ((uint32_t)r.bb & ((1u << ((d >> 8) & 15u)) - 1u))]);
        }
        if (!(d & K_BASE)) { ntok_out = ntok; return BATCH_DAMAGED; }
        r.bb >>= d & 15u;
        const uint32_t xd = (d >> 8) & 15u;
        tk.z = mine ? d : tk.z;
        tk.w = mine ? (uint32_t)r.bb : tk.w;
        r.bb >>= xd;
        r.bl -= (d & 15u) + xd;
        ntok++;
    } while (ntok < 64);
    ntok_out = ntok;
    return BATCH_FULL;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ inflate
// One wavefront per job.  in_area / text_area: the batch's compressed images and texts.
__global__ __launch_bounds__(64) void gz_inflate_kernel(const uint8_t *__restrict__ in_area, uint8_t *__restrict__ text_area, ghip_gz_job *__restrict__ jobs,
                                                        uint32_t n_jobs) {
    __shared__ InflateLds L;
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= n_jobs) return;
    ghip_gz_job *job = jobs + blockIdx.x;
    const uint32_t in_len = uni(job->in_len), text_cap = uni(job->text_cap);
    const uint8_t *in = in_area + uni64(job->in_off);
    uint8_t *text = text_area + uni64(job->text_off);
    uint32_t status = GHIP_GZ_OK, pos = 0, blocks = 0, crc_want = 0, n_tokens = 0, n_matches = 0, n_batches = 0, n_rounds = 0;

    // ---- the gzip header (RFC 1952 2.3): magic, deflate, flags; FEXTRA / FNAME / FCOMMENT skipped; FHCRC left to the host
    uint32_t at = 10;
    if (in_len < 18 || ld_byte(in, 0) != 0x1f || ld_byte(in, 1) != 0x8b || ld_byte(in, 2) != 8) status = GHIP_GZ_EFORMAT;
    else {
        const uint32_t flg = ld_byte(in, 3);
        if (flg & 0xe2u) status = GHIP_GZ_EFORMAT;   // reserved bits, or a header CRC
        if (status == GHIP_GZ_OK && (flg & 4u)) at += 2 + (ld_byte(in, 10) | (ld_byte(in, 11) << 8));
        for (uint32_t f = 8; f <= 16 && status == GHIP_GZ_OK; f <<= 1) {   // FNAME, FCOMMENT: zero-terminated
            if (!(flg & f)) continue;
            while (at < in_len && ld_byte(in, at) != 0) at++;
            at++;
        }
        if (status == GHIP_GZ_OK && (at > in_len || in_len - at < 8)) status = GHIP_GZ_EFORMAT;
    }

    Reader r;
    r.w = reinterpret_cast<const uint32_t *>(in);
    r.n_words = (in_len + 3) / 4;
    if (status == GHIP_GZ_OK) rd_seek(r, at, lane);
    bool last = false;
    while (status == GHIP_GZ_OK && !last) {
        if (rd_byte_pos(r) > in_len + 8) { status = GHIP_GZ_EDATA; break; }   // ran off the image (empty blocks made of the zeros behind it)
        rd_fill(r, lane);
        last = rd_bits(r, 1) != 0;
        const uint32_t type = rd_bits(r, 2);
        blocks++;
        if (type == 3) { status = GHIP_GZ_EDATA; break; }
        if (type == 0) {   // stored: LEN, ~LEN, bytes
            rd_bits(r, r.bl & 7u);
            rd_fill(r, lane);
            const uint32_t len = rd_bits(r, 16);
            rd_fill(r, lane);
            const uint32_t nlen = rd_bits(r, 16);
            const uint32_t from = rd_byte_pos(r);
            if (len != (~nlen & 0xffffu) || from > in_len || in_len - from < len) { status = GHIP_GZ_EDATA; break; }
            if (len > text_cap - pos) { status = GHIP_GZ_EOVERFLOW; break; }
            for (uint32_t i = lane; i < len; i += 64) text[pos + i] = in[from + i];
            pos += len;
            text_sync();
            rd_seek(r, from + len, lane);
            continue;
        }
        // ---- the block's two codes
        uint32_t hlit = LL_SYMS, hdist = D_SYMS;
        if (type == 1) {   // fixed (RFC 1951 3.2.6)
            for (uint32_t s = lane; s < LL_SYMS; s += 64) L.lens[LENS_LL + s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < D_SYMS) L.lens[LENS_D + lane] = 5;
        } else {           // dynamic (3.2.7): code lengths of the code-length code, then the two codes' lengths run-length coded
            rd_fill(r, lane);
            hlit = rd_bits(r, 5) + 257;
            hdist = rd_bits(r, 5) + 1;
            const uint32_t hclen = rd_bits(r, 4) + 4;
            if (hlit > 286 || hdist > 30) { status = GHIP_GZ_EDATA; break; }
            if (lane < 32) L.lens[LENS_PRE + lane] = 0;
            wave_sync();
            static constexpr uint8_t kOrder[PRE_SYMS] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (uint32_t i = 0; i < hclen; i++) {
                rd_fill(r, lane);
                const uint32_t v = rd_bits(r, 3);
                if (lane == 0) L.lens[LENS_PRE + kOrder[i]] = (uint8_t)v;
            }
            wave_sync();
            status = build_table<T_PRE>(L, LENS_PRE, PRE_SYMS, PRE_P, L.dt, 1u << PRE_P, lane);
            if (status != GHIP_GZ_OK) break;
            for (uint32_t s = lane; s < LL_SYMS + D_SYMS; s += 64) L.lens[s] = 0;
            wave_sync();
            uint32_t i = 0, prev = 0;
            const uint32_t total = hlit + hdist;
            while (i < total && status == GHIP_GZ_OK) {
                rd_fill(r, lane);
                const uint32_t e = uni(L.dt[(uint32_t)r.bb & ((1u << PRE_P) - 1u)]);
                if (!(e & K_LIT)) { status = GHIP_GZ_EDATA; break; }
                rd_bits(r, e & 15u);
                const uint32_t sym = e >> 16;
                uint32_t rep = 1, val = sym;
                if (sym == 16) { if (i == 0) { status = GHIP_GZ_EDATA; break; } rep = 3 + rd_bits(r, 2); val = prev; }
                else if (sym == 17) { rep = 3 + rd_bits(r, 3); val = 0; }
                else if (sym == 18) { rep = 11 + rd_bits(r, 7); val = 0; }
                if (rep > total - i) { status = GHIP_GZ_EDATA; break; }
                // litlen lengths at [0, hlit), distance lengths at LENS_D: index i of the joint sequence -> its place
                if (lane < rep && val) {   // (rep <= 138: up to three per lane)
                    for (uint32_t q = lane; q < rep; q += 64) { const uint32_t k = i + q; L.lens[k < hlit ? LENS_LL + k : LENS_D + (k - hlit)] = (uint8_t)val; }
                }
                i += rep;
                prev = val;
            }
            if (status != GHIP_GZ_OK) break;
            wave_sync();
            if (uni((uint32_t)L.lens[LENS_LL + 256]) == 0) { status = GHIP_GZ_EDATA; break; }   // no end-of-block code
        }
        wave_sync();
        status = build_table<T_LL>(L, LENS_LL, LL_SYMS, LL_P, L.ll, LL_ROOM, lane);
        if (status == GHIP_GZ_OK) status = build_table<T_D>(L, LENS_D, D_SYMS, D_P, L.dt, D_ROOM, lane);
        if (status != GHIP_GZ_OK) break;

        // ---- the block's symbols, 64 tokens at a time
        bool eob = false;
        while (!eob && status == GHIP_GZ_OK) {
            uint32_t ntok = 0;
            uint4 tk = make_uint4(0, 0, 0, 0);
            const uint32_t how = decode_batch(L, r, lane, ntok, tk);
            eob = how == BATCH_END_OF_BLOCK;
            if (how == BATCH_DAMAGED || rd_byte_pos(r) > in_len + 8) { status = GHIP_GZ_EDATA; break; }   // (the latter: ran off the image, zeros behind it)
            uint32_t written = 0;
            status = emit_tokens(text, L.stage, tk, ntok, pos, text_cap - pos, lane, written, n_matches, n_rounds);
            n_tokens += ntok;
            n_batches++;
            pos += written;
        }
    }
    if (status == GHIP_GZ_OK) {   // the trailer: CRC-32 and ISIZE, then nothing
        rd_bits(r, r.bl & 7u);
        rd_fill(r, lane);
        crc_want = rd_bits(r, 16);
        rd_fill(r, lane);
        crc_want |= rd_bits(r, 16) << 16;
        rd_fill(r, lane);
        uint32_t isize = rd_bits(r, 16);
        rd_fill(r, lane);
        isize |= rd_bits(r, 16) << 16;
        const uint32_t used = rd_byte_pos(r);
        if (used > in_len) status = GHIP_GZ_EDATA;
        else if (isize != pos) status = GHIP_GZ_ECRC;
        else if (used < in_len) status = GHIP_GZ_EMULTI;
    }
    if (lane == 0) {
        job->status = status;
        job->text_len = pos;
        job->crc_want = crc_want;
        job->blocks = blocks;
        job->crc_acc = 0;
        job->first_byte = 0xffffffffu;
        job->stream_len = 0; job->records = 0; job->ambiguous = 0; job->seq_bytes = 0; job->rec_off = 0;
        job->tokens = n_tokens; job->matches = n_matches; job->batches = n_batches; job->rounds = n_rounds;
    }
}

// ------------------------------------------------------------------------------------------------ CRC-32 of the texts
// Every work-item takes the remainder of its own 2 KiB span (register 0, table-driven from LDS), multiplies it by x^(8 *
// bytes behind the span) and the workgroup adds its sum into the job's accumulator (gz_common.h: crc_finish).
constexpr uint32_t CRC_SPAN = 2048, CRC_THREADS = 256;
__global__ __launch_bounds__(CRC_THREADS) void gz_crc_kernel(const uint8_t *__restrict__ text_area, ghip_gz_job *__restrict__ jobs) {
    __shared__ uint32_t table[256];
    __shared__ uint32_t acc;
    ghip_gz_job *job = jobs + blockIdx.y;
    const uint32_t n = job->text_len;
    if (job->status != GHIP_GZ_OK || (uint64_t)blockIdx.x * CRC_THREADS * CRC_SPAN >= n) return;
    table[threadIdx.x] = ghip_gz::crc_byte(0, threadIdx.x);
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const uint64_t from = ((uint64_t)blockIdx.x * CRC_THREADS + threadIdx.x) * CRC_SPAN;
    if (from < n) {
        const uint32_t m = (uint32_t)(n - from < CRC_SPAN ? n - from : CRC_SPAN);
        const uint8_t *p = text_area + job->text_off + from;   // (text_off and from are multiples of 16)
        uint32_t c = 0, i = 0;
        for (; i + 16 <= m; i += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int b = 0; b < 4; b++) c = table[(c ^ (w[q] >> (8 * b))) & 0xffu] ^ (c >> 8);
            }
        }
        for (; i < m; i++) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
        atomicXor(&acc, ghip_gz::gf_mul(ghip_gz::gf_x_pow_bytes(n - from - m), c));
    }
    __syncthreads();
    if (threadIdx.x == 0 && acc) atomicXor(&job->crc_acc, acc);
}

// ------------------------------------------------------------------------------------------------ the FASTA pass
// ghip_parse_fasta (ingest.cpp) as a parallel computation over the text T[0, n):
//   a byte is a LINE START when it follows a '\n' (or is T[0]); a line that starts with '>' is a header line and is dropped
//   whole; every other line is a sequence line: its bytes other than space, tab, '\r', '\n' are KEPT (upper-cased; anything
//   but A, C, G, T/U an invalid position), 'N'/'n' count as ambiguous, bytes other than '\r', '\n' count towards the record's
//   length.  Each header but the first puts one 'N' in front of it, the end of the text one more.
// Whether a byte is dropped depends on the start of its line, which may lie any distance in front: a 64-byte span (one
// work-item) or a 16 KiB chunk (one workgroup) is summarised as (counts in front of its first line start, taken as a
// sequence line; counts behind; whether it has a line start; the kind of its last line), fasta_scan runs over a file's
// chunk summaries, fasta_emit repeats the sweep with every span's incoming kind and offsets known.
// What ghip_parse_fasta does that this does not is left to it: a text whose first byte that is neither '\n' nor '\r' is
// not '>' (its error), or has a '\r' right in front of it (there the host parser sees a header where the rule above does not).
constexpr uint32_t FA_SPAN = 64, FA_THREADS = 256, FA_CHUNK = FA_SPAN * FA_THREADS;

struct ghip_fa_chunk {      // summary of one chunk, then (fasta_scan) what lies in front of it
    uint32_t pre_kept, pre_amb, pre_seq;               // in front of the chunk's first line start, were the line a sequence line
    uint32_t kept, amb, seq, headers;                  // from the first line start on
    uint32_t flags;                                    // bit 0: has a line start, bit 1: its last line is a header line
    uint32_t in_header, kept_before, headers_before, seq_before;   // fasta_scan: kind of the line running into the chunk; totals in front of it
};

namespace {

struct SpanSum { uint32_t pre_kept, pre_amb, pre_seq, kept, amb, seq, headers, has_start, last_header, first_other; };

__device__ __forceinline__ bool fa_space(uint32_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

// the 64 bytes of a span into registers (bytes behind the text read as '\n', which neither counts nor is kept)
__device__ __forceinline__ void fa_load(const uint8_t *text, uint32_t n, uint32_t from, uint32_t (&w)[16]) {
    const uint4 *p = reinterpret_cast<const uint4 *>(text + from);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = make_uint4(0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au);
        if (from + 16 * q < n) v = p[q];   // (the text area is padded: a partly valid 16 bytes may be read whole)
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
}

__device__ __forceinline__ SpanSum fa_span(const uint32_t (&w)[16], uint32_t valid, bool starts_line, uint32_t from) {
    SpanSum s{0, 0, 0, 0, 0, 0, 0, starts_line ? 1u : 0u, 0, 0xffffffffu};
    bool at_start = starts_line, header = false;
#pragma unroll
    for (uint32_t i = 0; i < FA_SPAN; i++) {
        const uint32_t c = (w[i / 4] >> (8 * (i % 4))) & 0xffu;
        const bool in = i < valid;
        if (in && at_start) { header = c == '>'; s.has_start = 1; s.headers += header ? 1u : 0u; }
        if (in && c != '\n' && c != '\r' && s.first_other == 0xffffffffu) s.first_other = from + i;
        const uint32_t count = (in && !header) ? 1u : 0u;
        const uint32_t k = count & (fa_space(c) ? 0u : 1u), a = count & ((c == 'N' || c == 'n') ? 1u : 0u), q = count & ((c != '\n' && c != '\r') ? 1u : 0u);
        if (s.has_start) { s.kept += k; s.amb += a; s.seq += q; } else { s.pre_kept += k; s.pre_amb += a; s.pre_seq += q; }
        at_start = in && c == '\n';
    }
    s.last_header = header ? 1u : 0u;
    return s;
}

// Kind of the line that runs into this work-item's span, from the spans in front of it in the workgroup: 0 sequence, 1
// header, 2 none of them has a line start (the chunk's own incoming kind decides).  lds: 2 * (FA_THREADS / 64) u64.
__device__ __forceinline__ uint32_t fa_incoming(bool has_start, bool last_header, unsigned long long *lds) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mh = __ballot(has_start), mt = __ballot(last_header);
    if (lane == 0) { lds[2 * wave] = mh; lds[2 * wave + 1] = mt; }
    __syncthreads();
    unsigned long long below = mh & ((1ull << lane) - 1ull), kinds = mt;
    for (int wv = (int)wave - 1; wv >= 0 && !below; wv--) { below = lds[2 * wv]; kinds = lds[2 * wv + 1]; }
    if (!below) return 2;
    return (uint32_t)(kinds >> (63 - __clzll((long long)below))) & 1u;
}

}  // namespace

// sweep 1: chunk summaries, and the first byte that is neither '\n' nor '\r'
__global__ __launch_bounds__(FA_THREADS) void fasta_chunk_kernel(const uint8_t *__restrict__ text_area, ghip_gz_job *__restrict__ jobs,
                                                                 const uint32_t *__restrict__ chunk_start, ghip_fa_chunk *__restrict__ chunks) {
    __shared__ unsigned long long kinds[2 * (FA_THREADS / 64)];
    __shared__ uint32_t sum[8];
    __shared__ uint32_t first_other;
    ghip_gz_job *job = jobs + blockIdx.y;
    const uint32_t n = job->text_len;
    if (job->status != GHIP_GZ_OK || (uint64_t)blockIdx.x * FA_CHUNK >= n) return;
    const uint8_t *text = text_area + job->text_off;
    const uint32_t from = blockIdx.x * FA_CHUNK + threadIdx.x * FA_SPAN;
    if (threadIdx.x < 8) sum[threadIdx.x] = 0;
    if (threadIdx.x == 8) first_other = 0xffffffffu;
    SpanSum s{};
    s.first_other = 0xffffffffu;
    if (from < n) {
        uint32_t w[16];
        fa_load(text, n, from, w);
        s = fa_span(w, n - from < FA_SPAN ? n - from : FA_SPAN, from == 0 || text[from - 1] == '\n', from);
    }
    const uint32_t in = fa_incoming(s.has_start != 0, s.last_header != 0, kinds);   // (has a __syncthreads: sum[] is zero behind it)
    // a span with no line start continues the incoming line with all of its bytes (they were counted as "pre")
    if (in == 2) { atomicAdd(&sum[0], s.pre_kept); atomicAdd(&sum[1], s.pre_amb); atomicAdd(&sum[2], s.pre_seq); }
    const uint32_t take = in == 0 ? 1u : 0u;
    atomicAdd(&sum[3], s.kept + take * s.pre_kept);
    atomicAdd(&sum[4], s.amb + take * s.pre_amb);
    atomicAdd(&sum[5], s.seq + take * s.pre_seq);
    atomicAdd(&sum[6], s.headers);
    if (s.first_other != 0xffffffffu) atomicMin(&first_other, s.first_other);
    __syncthreads();
    if (threadIdx.x == 0) {
        ghip_fa_chunk c{};
        c.pre_kept = sum[0]; c.pre_amb = sum[1]; c.pre_seq = sum[2];
        c.kept = sum[3]; c.amb = sum[4]; c.seq = sum[5]; c.headers = sum[6];
        unsigned long long has = 0, kind = 0;
        for (int wv = FA_THREADS / 64 - 1; wv >= 0 && !has; wv--) { has = kinds[2 * wv]; kind = kinds[2 * wv + 1]; }
        c.flags = has ? 1u | ((uint32_t)((kind >> (63 - __clzll((long long)has))) & 1ull) << 1) : 0u;
        chunks[chunk_start[blockIdx.y] + blockIdx.x] = c;
        if (first_other != 0xffffffffu) atomicMin(&job->first_byte, first_other);
    }
}

// the scan: one wavefront per file over its chunk summaries; the file's totals, its verdicts, its place in the record pool
__global__ __launch_bounds__(64) void fasta_scan_kernel(const uint8_t *__restrict__ text_area, ghip_gz_job *__restrict__ jobs, const uint32_t *__restrict__ chunk_start, ghip_fa_chunk *__restrict__ chunks, uint32_t *__restrict__ rec_next,
                                                        uint32_t rec_room, uint32_t n_jobs) {
    if (blockIdx.x >= n_jobs) return;
    ghip_gz_job *job = jobs + blockIdx.x;
    if (job->status != GHIP_GZ_OK) return;
    const uint32_t lane = threadIdx.x, n = job->text_len;
    if (ghip_gz::crc_finish(job->crc_acc, n) != job->crc_want) { if (lane == 0) job->status = GHIP_GZ_ECRC; return; }
    const uint8_t *text = text_area + job->text_off;
    const uint32_t first = job->first_byte < n ? job->first_byte : n;
    if (first < n && (text[first] != '>' || (first > 0 && text[first - 1] == '\r'))) { if (lane == 0) job->status = GHIP_GZ_EFASTA; return; }
    ghip_fa_chunk *mine = chunks + chunk_start[blockIdx.x];
    const uint32_t n_chunks = (n + FA_CHUNK - 1) / FA_CHUNK;
    uint32_t kept = 0, amb = 0, seq = 0, headers = 0, carry_kind = 0;   // (chunk 0 starts a line: the carry is never used for it)
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64) {
        const uint32_t c = c0 + lane;
        ghip_fa_chunk v{};
        if (c < n_chunks) v = mine[c];
        const unsigned long long mh = __ballot(v.flags & 1u), mt = __ballot(v.flags & 2u);
        const unsigned long long below = mh & ((1ull << lane) - 1ull);
        const uint32_t in = below ? (uint32_t)(mt >> (63 - __clzll((long long)below))) & 1u : carry_kind;
        const uint32_t take = in == 0 ? 1u : 0u;
        uint32_t x[4] = {v.kept + take * v.pre_kept, v.amb + take * v.pre_amb, v.seq + take * v.pre_seq, v.headers};
        uint32_t tot[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t y = x[q];
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t z = __shfl_up(y, d);
                if (lane >= d) y += z;
            }
            tot[q] = __shfl(y, 63);
            x[q] = y - x[q];   // exclusive
        }
        if (c < n_chunks) {
            mine[c].in_header = in;
            mine[c].kept_before = kept + x[0];
            mine[c].seq_before = seq + x[2];
            mine[c].headers_before = headers + x[3];
        }
        kept += tot[0]; amb += tot[1]; seq += tot[2]; headers += tot[3];
        if (mh) carry_kind = (uint32_t)(mt >> (63 - __clzll((long long)mh))) & 1u;
    }
    if (lane == 0) {
        uint32_t status = GHIP_GZ_OK, rec_off = 0;
        const uint64_t stream_len = (uint64_t)kept + headers;
        if (stream_len > job->stream_cap) status = GHIP_GZ_EOVERFLOW;
        else {   // a slice of the record pool -- taken only if it fits, so that a file of very many records costs the others nothing
            uint32_t seen = __hip_atomic_load(rec_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (other workgroups CAS it: not a plain read)
            for (;;) {
                if ((uint64_t)seen + headers > rec_room) { status = GHIP_GZ_EFASTA; break; }   // the host parser has no such limit
                const uint32_t was = atomicCAS(rec_next, seen, seen + headers);
                if (was == seen) { rec_off = seen; break; }
                seen = was;
            }
        }
        job->status = status;
        job->stream_len = (uint32_t)stream_len;
        job->records = headers;
        job->ambiguous = amb;
        job->seq_bytes = seq;
        job->rec_off = rec_off;
    }
}

// sweep 2: the genome in its resident form, and the record table.  A work-item's kept bytes are consecutive stream
// positions (a header start takes one too: the 'N' behind the record in front of it -- an invalid position like every byte
// other than A, C, G, T, i.e. no bit to set in arrays that start out zero).  The 2-bit codes and validity bits are gathered
// word by word in registers and OR-ed into place: neighbours share their first and last words.
__global__ __launch_bounds__(FA_THREADS) void fasta_emit_kernel(const uint8_t *__restrict__ text_area, const ghip_gz_job *__restrict__ jobs,
                                                                const uint32_t *__restrict__ chunk_start, const ghip_fa_chunk *__restrict__ chunks,
                                                                uint32_t *__restrict__ rec_pool, uint32_t *__restrict__ packed, uint32_t *__restrict__ valid_bits) {
    __shared__ unsigned long long kinds[2 * (FA_THREADS / 64)];
    __shared__ unsigned long long wave_sum[FA_THREADS / 64];
    const ghip_gz_job *job = jobs + blockIdx.y;
    const uint32_t n = job->text_len;
    if (job->status != GHIP_GZ_OK || (uint64_t)blockIdx.x * FA_CHUNK >= n) return;
    const uint8_t *text = text_area + job->text_off;
    uint32_t *rec = rec_pool + job->rec_off;
    uint32_t *pk = packed + job->gbase / 16, *vd = valid_bits + job->gbase / 32;   // (gbase is a multiple of 64)
    const ghip_fa_chunk ch = chunks[chunk_start[blockIdx.y] + blockIdx.x];
    const uint32_t from = blockIdx.x * FA_CHUNK + threadIdx.x * FA_SPAN, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t w[16] = {};
    uint32_t valid = 0;
    bool starts_line = false;
    SpanSum s{};
    if (from < n) {
        fa_load(text, n, from, w);
        valid = n - from < FA_SPAN ? n - from : FA_SPAN;
        starts_line = from == 0 || text[from - 1] == '\n';
        s = fa_span(w, valid, starts_line, from);
    }
    uint32_t in = fa_incoming(s.has_start != 0, s.last_header != 0, kinds);
    if (in == 2) in = ch.in_header;
    const uint32_t take = in == 0 ? 1u : 0u;
    // exclusive prefix over the workgroup of (kept, headers, seq): 20 bits each in one word
    const unsigned long long my = (unsigned long long)(s.kept + take * s.pre_kept) | ((unsigned long long)s.headers << 20) |
                                  ((unsigned long long)(s.seq + take * s.pre_seq) << 40);
    unsigned long long y = my;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const unsigned long long z = __shfl_up(y, d);
        if (lane >= d) y += z;
    }
    if (lane == 63) wave_sum[wave] = y;
    __syncthreads();
    unsigned long long before = y - my;
    for (uint32_t wv = 0; wv < wave; wv++) before += wave_sum[wv];
    uint32_t kept = ch.kept_before + (uint32_t)(before & 0xfffffu), headers = ch.headers_before + (uint32_t)((before >> 20) & 0xfffffu),
             seq = ch.seq_before + (uint32_t)(before >> 40);
    bool at_start = starts_line, header = in == 1;
    uint32_t word_p = 0xffffffffu, word_v = 0xffffffffu, acc_p = 0, acc_v = 0;   // the packed / validity word being gathered
#pragma unroll
    for (uint32_t i = 0; i < FA_SPAN; i++) {   // (unrolled and predicated: w[] stays in registers)
        const uint32_t c = (w[i / 4] >> (8 * (i % 4))) & 0xffu;
        const bool in = i < valid;
        if (in && at_start) {
            header = c == '>';
            if (header) { rec[headers] = seq; headers++; }   // (from the second header on this also passes over the 'N' behind the record in front)
        }
        if (in && !header) {
            if (!fa_space(c)) {
                const uint32_t u = c & 0xdfu, at = kept + headers - 1;   // stream position (a kept byte has a header in front of it)
                const bool t = u == 'T' || u == 'U';
                if (u == 'A' || u == 'C' || u == 'G' || t) {
                    if ((at >> 4) != word_p) { if (acc_p) atomicOr(&pk[word_p], acc_p); acc_p = 0; word_p = at >> 4; }
                    if ((at >> 5) != word_v) { if (acc_v) atomicOr(&vd[word_v], acc_v); acc_v = 0; word_v = at >> 5; }
                    acc_p |= (t ? 3u : ((u >> 1) ^ (u >> 2)) & 3u) << (2 * (at & 15u));   // A0 C1 G2 T3
                    acc_v |= 1u << (at & 31u);
                }
                kept++;
            }
            seq += (c != '\n' && c != '\r') ? 1u : 0u;
        }
        at_start = in && c == '\n';
    }
    if (acc_p) atomicOr(&pk[word_p], acc_p);
    if (acc_v) atomicOr(&vd[word_v], acc_v);
}

// ------------------------------------------------------------------------------------------------ launchers
size_t ghip_gz_chunks_of(uint64_t text_cap) { return (size_t)((text_cap + FA_CHUNK - 1) / FA_CHUNK); }
size_t ghip_gz_chunk_bytes() { return sizeof(ghip_fa_chunk); }

// the whole device path of one batch on `stream`: inflate, CRC, FASTA pass straight into the resident arrays (which are
// zero where these genomes go).  max_text_cap = the largest text_cap of the batch; d_chunk_start[j] = first chunk summary of
// job j; d_rec_next = one zeroed word; rec_room = entries of d_rec_pool.
void ghip_launch_gz_batch(hipStream_t stream, const uint8_t *d_in, uint8_t *d_text, ghip_gz_job *d_jobs, uint32_t n_jobs, uint64_t max_text_cap,
                          const uint32_t *d_chunk_start, void *d_chunks, uint32_t *d_rec_next, uint32_t *d_rec_pool, uint32_t rec_room, uint32_t *d_packed,
                          uint32_t *d_valid) {
    if (n_jobs == 0) return;
    ghip_fa_chunk *chunks = reinterpret_cast<ghip_fa_chunk *>(d_chunks);
    hipLaunchKernelGGL(gz_inflate_kernel, dim3(n_jobs), dim3(64), 0, stream, d_in, d_text, d_jobs, n_jobs);
    const unsigned n_chunks = (unsigned)ghip_gz_chunks_of(max_text_cap);
    if (max_text_cap) {
        const unsigned crc_blocks = (unsigned)((max_text_cap + (uint64_t)CRC_THREADS * CRC_SPAN - 1) / ((uint64_t)CRC_THREADS * CRC_SPAN));
        hipLaunchKernelGGL(gz_crc_kernel, dim3(crc_blocks, n_jobs), dim3(CRC_THREADS), 0, stream, d_text, d_jobs);
        hipLaunchKernelGGL(fasta_chunk_kernel, dim3(n_chunks, n_jobs), dim3(FA_THREADS), 0, stream, d_text, d_jobs, d_chunk_start, chunks);
    }
    hipLaunchKernelGGL(fasta_scan_kernel, dim3(n_jobs), dim3(64), 0, stream, d_text, d_jobs, d_chunk_start, chunks, d_rec_next, rec_room, n_jobs);
    if (max_text_cap)
        hipLaunchKernelGGL(fasta_emit_kernel, dim3(n_chunks, n_jobs), dim3(FA_THREADS), 0, stream, d_text, d_jobs, d_chunk_start, chunks, d_rec_pool, d_packed, d_valid);
}
