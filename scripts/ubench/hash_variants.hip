// What the ORDER of the hash's instructions costs, and what each class of instruction costs inside it: the filter block
// half after half, with one class removed at a time (wrong results, timing only), with the halves interleaved (the order
// murmur21_asm.h uses), and two evaluations interleaved; 8 waves per SIMD, one dependent evaluation after the other per wave.
// Kernels this short are timed at whatever clock the box has reached: the first launches of a process run slower, which
// once made the old order look 12 % worse than it is -- hence the warm-up and the repeats at the end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../galah_amd/csrc/murmur21_asm.h"

// The hash written half after half (the product's order until round 3; murmur21_asm.h now interleaves the halves):
#define GHIP_MULC(R0, R1, E0, E1, X0, X1, CLO, CHI)                                     \
    "v_mad_u64_u32 v[" #E0 ":" #E1 "], vcc, " X1 ", " CLO ", 0\n"                      \
    "v_mad_u64_u32 v[" #E0 ":" #E1 "], vcc, " X0 ", " CHI ", v[" #E0 ":" #E1 "]\n"     \
    "v_mad_u64_u32 v[" #R0 ":" #R1 "], vcc, " X0 ", " CLO ", 0\n"                      \
    "v_add_u32 v" #R1 ", v" #R1 ", v" #E0 "\n"
#define GHIP_XORSHIFT33(LO, HI) "v_lshrrev_b32 v52, 1, " HI "\n v_xor_b32 " LO ", " LO ", v52\n"
#define BODY_SERIAL                                                                      \
    "v_alignbit_b32 v48, %[a0], %[a1], 1\n"                                             \
    "v_alignbit_b32 v49, %[a1], %[a0], 1\n"                                             \
    GHIP_MULC(40, 41, 42, 43, "v48", "v49", "%[c2lo]", "%[c2hi]")                       \
    "v_alignbit_b32 v48, v40, v41, 5\n"                                                 \
    "v_alignbit_b32 v49, v41, v40, 5\n"                                                 \
    "v_lshl_add_u64 v[48:49], v[48:49], 2, v[48:49]\n"                                  \
    "v_lshl_add_u64 v[48:49], v[48:49], 0, %[k52]\n"                                    \
    "v_alignbit_b32 v50, %[b1], %[b0], 31\n"                                            \
    "v_alignbit_b32 v51, %[b0], %[b1], 31\n"                                            \
    GHIP_MULC(44, 45, 46, 47, "v50", "v51", "%[c1lo]", "%[c1hi]")                       \
    "v_alignbit_b32 v50, v44, v45, 1\n"                                                 \
    "v_alignbit_b32 v51, v45, v44, 1\n"                                                 \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, v[48:49]\n"                                  \
    "v_lshl_add_u64 v[50:51], v[50:51], 2, v[50:51]\n"                                  \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, %[k38]\n"                                    \
    "v_xor_b32 v48, v48, %[t0]\n"                                                       \
    "v_xor_b32 v49, v49, %[t1]\n"                                                       \
    "v_xor_b32 v50, 21, v50\n"                                                          \
    "v_lshl_add_u64 v[48:49], v[48:49], 0, v[50:51]\n"                                  \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, v[48:49]\n"                                  \
    GHIP_XORSHIFT33("v48", "v49")                                                       \
    GHIP_MULC(40, 41, 42, 43, "v48", "v49", "%[f1lo]", "%[f1hi]")                       \
    GHIP_XORSHIFT33("v40", "v41")                                                       \
    GHIP_XORSHIFT33("v50", "v51")                                                       \
    GHIP_MULC(44, 45, 46, 47, "v50", "v51", "%[f1lo]", "%[f1hi]")                       \
    GHIP_XORSHIFT33("v44", "v45")                                                       \
    "v_lshl_add_u64 v[48:49], v[40:41], 0, v[44:45]\n"                                  \
    "v_mad_u64_u32 v[42:43], vcc, v49, %[f2lo], 0\n"                                    \
    "v_mad_u64_u32 v[42:43], vcc, v48, %[f2hi], v[42:43]\n"                             \
    "v_mad_u64_u32 v[50:51], vcc, v48, %[f2lo], 0\n"                                    \
    "v_add3_u32 %[s1], v51, v42, 1\n"

#define OPERANDS                                                                                                     \
    [s1] "=&v"(s1), "=&{v[40:41]}"(A), "=&{v[44:45]}"(B)                                                              \
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [t0] "v"(t0), [t1] "v"(t1),                        \
        [c1lo] "s"(0x114253d5u), [c1hi] "s"(0x87c37b91u), [c2lo] "s"(0x2745937fu), [c2hi] "s"(0x4cf5ad43u),          \
        [f1lo] "s"(0xed558ccdu), [f1hi] "s"(0xff51afd7u), [f2lo] "s"(0x1a85ec53u), [f2hi] "s"(0xc4ceb9feu),          \
        [k52] "s"(k52), [k38] "s"(k38)                                                                               \
        : "v42", "v43", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "vcc"

#define VARIANT(NAME, BODY)                                                                                          \
    __global__ void NAME(uint32_t *out, uint32_t seed, uint32_t iters) {                                             \
        uint32_t a0 = seed + threadIdx.x, a1 = seed * 3 + threadIdx.x, b0 = a0 ^ 0x1234567, b1 = a1 + 99, t0 = a0 * 7, t1 = a1 * 11; \
        const uint64_t k52 = 0x52dce729ull, k38 = 0x38495ab5ull;                                                     \
        uint32_t acc = 0;                                                                                            \
        for (uint32_t i = 0; i < iters; i++) {                                                                       \
            uint64_t A, B; uint32_t s1;                                                                              \
            asm volatile(BODY : OPERANDS);                                                                           \
            acc += s1; a0 = (uint32_t)A; a1 ^= (uint32_t)B;                                                          \
        }                                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                            \
    }

VARIANT(k_full, BODY_SERIAL)

// -- no xor-shifts (8 cheap)
#undef GHIP_XORSHIFT33
#define GHIP_XORSHIFT33(LO, HI) ""
VARIANT(k_no_xorshift, BODY_SERIAL)
#undef GHIP_XORSHIFT33
#define GHIP_XORSHIFT33(LO, HI) "v_lshrrev_b32 v52, 1, " HI "\n v_xor_b32 " LO ", " LO ", v52\n"

// -- multiplies without their cross terms (2 of 3 mads and the add gone: 10 mads, 4 adds... of the 5 MULC-shaped ones 4 are macros)
#undef GHIP_MULC
#define GHIP_MULC(R0, R1, E0, E1, X0, X1, CLO, CHI) "v_mad_u64_u32 v[" #R0 ":" #R1 "], vcc, " X0 ", " CLO ", 0\n"
VARIANT(k_one_mad, BODY_SERIAL)
// -- multiplies without the add only
#undef GHIP_MULC
#define GHIP_MULC(R0, R1, E0, E1, X0, X1, CLO, CHI)                                     \
    "v_mad_u64_u32 v[" #E0 ":" #E1 "], vcc, " X1 ", " CLO ", 0\n"                      \
    "v_mad_u64_u32 v[" #E0 ":" #E1 "], vcc, " X0 ", " CHI ", v[" #E0 ":" #E1 "]\n"     \
    "v_mad_u64_u32 v[" #R0 ":" #R1 "], vcc, " X0 ", " CLO ", 0\n"
VARIANT(k_no_add, BODY_SERIAL)
// -- no multiplies at all (4 macros x 4 instructions)
#undef GHIP_MULC
#define GHIP_MULC(R0, R1, E0, E1, X0, X1, CLO, CHI) "v_mov_b32 v" #R0 ", " X0 "\n v_mov_b32 v" #R1 ", " X1 "\n"
VARIANT(k_no_mulc, BODY_SERIAL)


// -- V2: the same 47 instructions, the two independent halves interleaved so that the multiplies stand together: the product's order
VARIANT(k_v2, GHIP_MURMUR21_BODY("", "", ""))

// -- V3: two evaluations at a time, instruction by instruction (what interleaving two positions of the k-mer loop would do)
#define BODY_V3 \
    "v_alignbit_b32 v48, %[a0], %[a1], 1\n" \
    "v_alignbit_b32 v68, %[a0x], %[a1x], 1\n" \
    "v_alignbit_b32 v49, %[a1], %[a0], 1\n" \
    "v_alignbit_b32 v69, %[a1x], %[a0x], 1\n" \
    "v_alignbit_b32 v50, %[b1], %[b0], 31\n" \
    "v_alignbit_b32 v70, %[b1x], %[b0x], 31\n" \
    "v_alignbit_b32 v51, %[b0], %[b1], 31\n" \
    "v_alignbit_b32 v71, %[b0x], %[b1x], 31\n" \
    "v_mad_u64_u32 v[42:43], vcc, v49, %[c2lo], 0\n" \
    "v_mad_u64_u32 v[62:63], vcc, v69, %[c2lo], 0\n" \
    "v_mad_u64_u32 v[46:47], vcc, v51, %[c1lo], 0\n" \
    "v_mad_u64_u32 v[66:67], vcc, v71, %[c1lo], 0\n" \
    "v_mad_u64_u32 v[40:41], vcc, v48, %[c2lo], 0\n" \
    "v_mad_u64_u32 v[60:61], vcc, v68, %[c2lo], 0\n" \
    "v_mad_u64_u32 v[44:45], vcc, v50, %[c1lo], 0\n" \
    "v_mad_u64_u32 v[64:65], vcc, v70, %[c1lo], 0\n" \
    "v_mad_u64_u32 v[42:43], vcc, v48, %[c2hi], v[42:43]\n" \
    "v_mad_u64_u32 v[62:63], vcc, v68, %[c2hi], v[62:63]\n" \
    "v_mad_u64_u32 v[46:47], vcc, v50, %[c1hi], v[46:47]\n" \
    "v_mad_u64_u32 v[66:67], vcc, v70, %[c1hi], v[66:67]\n" \
    "v_add_u32 v41, v41, v42\n" \
    "v_add_u32 v61, v61, v62\n" \
    "v_add_u32 v45, v45, v46\n" \
    "v_add_u32 v65, v65, v66\n" \
    "v_alignbit_b32 v48, v40, v41, 5\n" \
    "v_alignbit_b32 v68, v60, v61, 5\n" \
    "v_alignbit_b32 v49, v41, v40, 5\n" \
    "v_alignbit_b32 v69, v61, v60, 5\n" \
    "v_alignbit_b32 v50, v44, v45, 1\n" \
    "v_alignbit_b32 v70, v64, v65, 1\n" \
    "v_alignbit_b32 v51, v45, v44, 1\n" \
    "v_alignbit_b32 v71, v65, v64, 1\n" \
    "v_lshl_add_u64 v[48:49], v[48:49], 2, v[48:49]\n" \
    "v_lshl_add_u64 v[68:69], v[68:69], 2, v[68:69]\n" \
    "v_lshl_add_u64 v[48:49], v[48:49], 0, %[k52]\n" \
    "v_lshl_add_u64 v[68:69], v[68:69], 0, %[k52]\n" \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, v[48:49]\n" \
    "v_lshl_add_u64 v[70:71], v[70:71], 0, v[68:69]\n" \
    "v_lshl_add_u64 v[50:51], v[50:51], 2, v[50:51]\n" \
    "v_lshl_add_u64 v[70:71], v[70:71], 2, v[70:71]\n" \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, %[k38]\n" \
    "v_lshl_add_u64 v[70:71], v[70:71], 0, %[k38]\n" \
    "v_xor_b32 v48, v48, %[t0]\n" \
    "v_xor_b32 v68, v68, %[t0x]\n" \
    "v_xor_b32 v49, v49, %[t1]\n" \
    "v_xor_b32 v69, v69, %[t1x]\n" \
    "v_xor_b32 v50, 21, v50\n" \
    "v_xor_b32 v70, 21, v70\n" \
    "v_lshl_add_u64 v[48:49], v[48:49], 0, v[50:51]\n" \
    "v_lshl_add_u64 v[68:69], v[68:69], 0, v[70:71]\n" \
    "v_lshl_add_u64 v[50:51], v[50:51], 0, v[48:49]\n" \
    "v_lshl_add_u64 v[70:71], v[70:71], 0, v[68:69]\n" \
    "v_lshrrev_b32 v52, 1, v49\n" \
    "v_lshrrev_b32 v72, 1, v69\n" \
    "v_lshrrev_b32 v42, 1, v51\n" \
    "v_lshrrev_b32 v62, 1, v71\n" \
    "v_xor_b32 v48, v48, v52\n" \
    "v_xor_b32 v68, v68, v72\n" \
    "v_xor_b32 v50, v50, v42\n" \
    "v_xor_b32 v70, v70, v62\n" \
    "v_mad_u64_u32 v[42:43], vcc, v49, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[62:63], vcc, v69, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[46:47], vcc, v51, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[66:67], vcc, v71, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[40:41], vcc, v48, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[60:61], vcc, v68, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[44:45], vcc, v50, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[64:65], vcc, v70, %[f1lo], 0\n" \
    "v_mad_u64_u32 v[42:43], vcc, v48, %[f1hi], v[42:43]\n" \
    "v_mad_u64_u32 v[62:63], vcc, v68, %[f1hi], v[62:63]\n" \
    "v_mad_u64_u32 v[46:47], vcc, v50, %[f1hi], v[46:47]\n" \
    "v_mad_u64_u32 v[66:67], vcc, v70, %[f1hi], v[66:67]\n" \
    "v_add_u32 v41, v41, v42\n" \
    "v_add_u32 v61, v61, v62\n" \
    "v_add_u32 v45, v45, v46\n" \
    "v_add_u32 v65, v65, v66\n" \
    "v_lshrrev_b32 v52, 1, v41\n" \
    "v_lshrrev_b32 v72, 1, v61\n" \
    "v_lshrrev_b32 v42, 1, v45\n" \
    "v_lshrrev_b32 v62, 1, v65\n" \
    "v_xor_b32 v40, v40, v52\n" \
    "v_xor_b32 v60, v60, v72\n" \
    "v_xor_b32 v44, v44, v42\n" \
    "v_xor_b32 v64, v64, v62\n" \
    "v_lshl_add_u64 v[48:49], v[40:41], 0, v[44:45]\n" \
    "v_lshl_add_u64 v[68:69], v[60:61], 0, v[64:65]\n" \
    "v_mad_u64_u32 v[42:43], vcc, v49, %[f2lo], 0\n" \
    "v_mad_u64_u32 v[62:63], vcc, v69, %[f2lo], 0\n" \
    "v_mad_u64_u32 v[50:51], vcc, v48, %[f2lo], 0\n" \
    "v_mad_u64_u32 v[70:71], vcc, v68, %[f2lo], 0\n" \
    "v_mad_u64_u32 v[42:43], vcc, v48, %[f2hi], v[42:43]\n" \
    "v_mad_u64_u32 v[62:63], vcc, v68, %[f2hi], v[62:63]\n" \
    "v_add3_u32 %[s1], v51, v42, 1\n" \
    "v_add3_u32 %[s1x], v71, v62, 1\n"
__global__ void k_v3(uint32_t *out, uint32_t seed, uint32_t iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = seed * 3 + threadIdx.x, b0 = a0 ^ 0x1234567, b1 = a1 + 99, t0 = a0 * 7, t1 = a1 * 11;
    uint32_t a0x = a0 * 13, a1x = a1 ^ 0x777, b0x = b0 + 5, b1x = b1 * 3, t0x = t0 ^ 1, t1x = t1 + 9;
    const uint64_t k52 = 0x52dce729ull, k38 = 0x38495ab5ull;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i += 2) {
        uint64_t A, B, Ax, Bx; uint32_t s1, s1x;
        asm volatile(BODY_V3 : [s1] "=&v"(s1), "=&{v[40:41]}"(A), "=&{v[44:45]}"(B), [s1x] "=&v"(s1x), "=&{v[60:61]}"(Ax), "=&{v[64:65]}"(Bx)
                     : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [t0] "v"(t0), [t1] "v"(t1),
                       [a0x] "v"(a0x), [a1x] "v"(a1x), [b0x] "v"(b0x), [b1x] "v"(b1x), [t0x] "v"(t0x), [t1x] "v"(t1x),
                       [c1lo] "s"(0x114253d5u), [c1hi] "s"(0x87c37b91u), [c2lo] "s"(0x2745937fu), [c2hi] "s"(0x4cf5ad43u),
                       [f1lo] "s"(0xed558ccdu), [f1hi] "s"(0xff51afd7u), [f2lo] "s"(0x1a85ec53u), [f2hi] "s"(0xc4ceb9feu),
                       [k52] "s"(k52), [k38] "s"(k38)
                     : "v42", "v43", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v62", "v63", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "vcc");
        acc += s1 + s1x; a0 = (uint32_t)A; a1 ^= (uint32_t)B; a0x = (uint32_t)Ax; a1x ^= (uint32_t)Bx;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename K>
static double run(const char *name, K kern, uint32_t *d_out, int n_inst) {
    const int blocks = 256 * 8, threads = 256;
    const uint32_t iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, iters);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double cyc = best * 1e-3 * 2.4e9 * 1024 / ((double)blocks * (threads / 64) * iters);
    printf("%-28s %3d instructions  %7.1f SIMD-cycles per evaluation  (%.2f per instruction)\n", name, n_inst, cyc, cyc / n_inst);
    return cyc;
}

int main() {
    uint32_t *d_out;
    (void)hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4);
    // (the clocks settle during the first launches: the two orders are timed again at the end)
    for (int warm = 0; warm < 20; warm++) hipLaunchKernelGGL(k_v2, dim3(256 * 8), dim3(256), 0, 0, d_out, 1u, 2048u);
    (void)hipDeviceSynchronize();
    run("halves interleaved (murmur21_asm.h)", k_v2, d_out, 47);
    const double full = run("half after half (old order)", k_full, d_out, 47);
    const double a = run("without the xor-shifts", k_no_xorshift, d_out, 39);
    const double b = run("one mad per multiply", k_one_mad, d_out, 35);
    const double c = run("multiplies without the add", k_no_add, d_out, 43);
    const double d = run("multiplies as two moves", k_no_mulc, d_out, 39);
    run("halves interleaved (murmur21_asm.h)", k_v2, d_out, 47);
    run("V3: two evaluations interleaved", k_v3, d_out, 47);
    run("half after half (old order), again", k_full, d_out, 47);
    run("halves interleaved, again", k_v2, d_out, 47);
    printf("marginal: xor-shift pair %.2f per instruction, cross-term mad+mad+add %.2f per instruction, the add %.2f, a whole multiply %.1f\n",
           (full - a) / 8, (full - b) / 12, (full - c) / 4, (full - d) / 4 + 0);
    // V2 computes the same values: the two kernels' outputs agree
    std::vector<uint32_t> x(256 * 8 * 256), y(x.size());
    hipLaunchKernelGGL(k_full, dim3(256 * 8), dim3(256), 0, 0, d_out, 777u, 64u);
    (void)hipMemcpy(x.data(), d_out, x.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_v2, dim3(256 * 8), dim3(256), 0, 0, d_out, 777u, 64u);
    (void)hipMemcpy(y.data(), d_out, y.size() * 4, hipMemcpyDeviceToHost);
    printf("V2 == full: %s\n", x == y ? "yes" : "NO");
    return 0;
}
